/* mscl_hip.h -- C ABI of libmscl_hip.so: the MI355X (gfx950) kernels of the MSCL training hot path.
 *
 * The reference has NO native layer on this path: every entry point below replaces
 * a PyTorch op chain of the reference, cited per function as reference file:line (relative to the
 * reference repo root).  Conventions:
 *   - activations are NDHWC bf16 (uint16 storage), channel count a multiple of 8;
 *   - parameters are fp32 masters with a bf16 shadow; conv kernels are laid out [Cout][kT][kH][kW][Cin]
 *     (the memory order of a channels_last_3d torch tensor of logical shape (Cout,Cin,kT,kH,kW));
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     immediately: 0 = ok, <0 = argument error (MSCL_E_*), >0 = hipError_t of the launch;
 *   - pointers are device pointers borrowed for the duration of the call; nothing is allocated.
 */
#ifndef MSCL_HIP_H
#define MSCL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MSCL_E_ARG      (-1)   /* null pointer / non-positive size */
#define MSCL_E_SHAPE    (-2)   /* shape unsupported by the kernel family (e.g. C % 8 != 0) */
#define MSCL_E_STRIDE   (-3)   /* stride other than 1 or 2 where the kernel needs a power of two */

typedef struct {
  int N, T, H, W, C;      /* conv input  x  (N,T,H,W,C)  */
  int To, Ho, Wo, K;      /* conv output y  (N,To,Ho,Wo,K) */
  int kT, kH, kW;
  int sT, sH, sW;
  int pT, pH, pW;
} mscl_conv_desc;

/* BatchNorm statistics buffers (the `ssum` / `ssq` outputs of the convolution entry points and the `sum` / `sumsq` inputs
 * of mscl_bn_params) are MSCL_STAT_SLOTS copies of the [2][C] sums, laid out [slot][2][C] fp32 and zeroed by the caller:
 * the pointers address slot 0, producers add into a slot chosen by their block index, consumers add the slots up.  (Thousands
 * of blocks adding into one 512-byte row run an order of magnitude below the float-atomic rate.)  Outside deterministic mode
 * only the first MSCL_STAT_ACTIVE (4) slots are used -- every block of a consuming pass re-reads the slots of every channel, and
 * that cost more than the contention more slots avoid; in deterministic mode each of up to MSCL_STAT_SLOTS blocks owns a slot.
 * A caller that fills a statistics buffer itself puts its sums into the first MSCL_STAT_ACTIVE slots. */
#define MSCL_STAT_ACTIVE 4
#ifndef MSCL_STAT_SLOTS
#define MSCL_STAT_SLOTS 16
#endif

int mscl_abi_version(void);

/* Deterministic mode (the reference's documented launch is `tools/train.py --deterministic`, README.md:19, tools/train.py:55-57,
 * 149): while on, every sum that is otherwise taken by float atomics in hardware order -- BatchNorm statistics in the conv
 * epilogues, the BatchNorm-backward sums, weight / bias gradients over position splits, the small linear layers' input
 * gradient, the InfoNCE gradient, the LMCL loss -- is taken in a fixed order (per-block partials in slots or slabs, added in
 * index order), so two runs on the same inputs give bit-identical results.  Costs one extra read of each conv output (the
 * statistics pass, mscl_bn_stats), a fold launch per BatchNorm sum and a slab pass per weight gradient.  Process-wide; set it before the first step.
 * Not covered: mscl_conv_halo64 called directly with statistics pointers. */
/* test aid: number of launches the ping-pong shared-tap conv kernel (conv_pp.hip) has taken in this process, so that a parity
 * test can assert which kernel family produced the result it checked */
int64_t mscl_debug_pp_launches(void);
/* the same for the window-resident 1x3x3 kernel of the 16- / 32-channel maps (conv_thin.hip) */
/* the same for the window-resident layer-1 kernels: forward / input gradient (conv_halo.hip), weight gradient (conv_wgrad_halo.hip) */
int64_t mscl_debug_halo_launches(void);
int64_t mscl_debug_stem_launches(void);   /* conv_stem.hip: the window-resident RGB-stem forward */
int64_t mscl_debug_wgrad_halo_launches(void);
int64_t mscl_debug_wgrad_stem_launches(void);   /* window-resident weight gradient of the W-paired RGB stem (conv_wgrad_stem.hip) */
int64_t mscl_debug_thin_launches(void);
int64_t mscl_debug_k1_launches(void);         /* thin-K 1x1x1 streaming kernel (csrc/conv_k1.hip) */
int64_t mscl_debug_dgrad_s2_launches(void);   /* window-resident stride-2 input gradient (csrc/conv_dgrad_s2.hip) */
int64_t mscl_debug_thin_wgrad_launches(void);
int mscl_set_deterministic(int on);
int mscl_get_deterministic(void);
/* The library's tuning switches (MSCL_* environment variables) are read once and cached; after this call every
 * switch re-reads its variable at its next use (tests and A/B sweeps that flip a switch inside one process). */
int mscl_tuning_reload(void);
/* BatchNorm batch statistics of a stored bf16 map (rows, C) in `groups` statistics groups, summed in a fixed order and stored
 * into slot 0 of ssum / ssq ([group][slot][2][C], ssq = ssum + C; the other slots must hold zeros and do so afterwards).
 * Two levels: per-block partials over contiguous row shares, then one add per channel in partial order; the number of partials
 * depends on (rows, C, groups) alone.  `parts`: scratch of >= mscl_det_parts_floats(rows, C, groups, 2) floats for the partials
 * (up to MSCL_DET_PARTS = 128 blocks per group); NULL or smaller: the MSCL_STAT_SLOTS slots hold them (16 blocks, several
 * times slower on the large maps). */
int64_t mscl_det_parts_floats(int64_t rows, int C, int groups, int vecs);
int mscl_bn_stats(const uint16_t* y, float* ssum, float* ssq, int64_t rows, int C, int groups, float* parts, int64_t parts_floats,
                  void* stream);
/* floats of workspace mscl_conv3d_wgrad wants for this layer: the slabs of the window-resident kernels (conv_wgrad_halo.hip,
 * conv_thin.hip) where they apply, the per-split slabs of deterministic mode otherwise (0: none needed) */
int64_t mscl_conv3d_wgrad_ws(const mscl_conv_desc* d, int with_bias);
/* floats of workspace the window-resident weight-gradient kernel wants (conv_wgrad_halo.hip: one 9 x 64 x 64 partial per block,
 * added in slot order); 0 = the layer is not one of its shapes (3x3x3 / 1 / 1, channels multiples of 64, planes that fill
 * 256-position tiles).  Included in mscl_conv3d_wgrad_ws. */
int64_t mscl_wgrad_halo_ws(const mscl_conv_desc* d);
/* floats of workspace the window-resident weight-gradient kernel of the 1x3x3 16- / 32-channel layers wants (conv_thin.hip: one
 * 9 x K x C partial per block, added in block order); 0 = the layer is not one of them.  Included in mscl_conv3d_wgrad_ws. */
int64_t mscl_wgrad_thin_ws(const mscl_conv_desc* d);

/* ---- Conv3d as implicit GEMM on MFMA (bf16 in, fp32 accumulate) --------------------------------
 * replaces nn.Conv3d forward in r3d.py:16-34,176-184,285-288 / fastonly.py:61-80,185-193 /
 * necks/fpn.py:131-149 / necks/sepc.py:74-104.
 * y = conv(x, w) [+ bias] [+ addend] [relu];  optional per-channel sum / sum-of-squares of the
 * fp32 result accumulated (atomically) into stat_sum/stat_sq (K floats each, caller zeroes them):
 * the BatchNorm batch statistics of r3d.py:103-127 fused into the producer. */
int mscl_conv3d_fwd(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* w_bf16, uint16_t* y,
                    const float* bias, const uint16_t* addend, int relu,
                    float* stat_sum, float* stat_sq, float* splitk_ws, int64_t splitk_ws_floats, void* stream);
/* The same with `stat_groups` BatchNorm statistics groups: the batch holds stat_groups calls of the reference's module
 * (the base and the rotated flow clips of recognizers/mscl.py:239-240, which the reference feeds through the flow encoder
 * one after the other, each with its own batch statistics); samples [k*N/G, (k+1)*N/G) feed the k-th [slot][2][C] block of
 * stat_sum / stat_sq (G * MSCL_STAT_SLOTS * 2 * K floats).  One launch instead of G for the convolution itself. */
int mscl_conv3d_fwd_groups(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* w_bf16, uint16_t* y,
                           const float* bias, const uint16_t* addend, int relu, float* stat_sum, float* stat_sq,
                           int stat_groups, float* splitk_ws, int64_t splitk_ws_floats, void* stream);
/* splitk_ws: optional fp32 scratch of splitk_ws_floats floats; when it holds >= 2 copies of the output and
 * the layer has too few position tiles to fill 256 CUs, the K loop is split over the grid (one fp32 slab
 * per split, then a summing finalize pass that also applies the epilogue and the BN statistics). */

/* Specialised 3x3x3 / stride 1 / pad 1 / 64->64 path (halo-resident window in LDS, conv_halo.hip); mode 0 =
 * forward (w laid out [Cout][tap][Cin]), 1 = input gradient (wT laid out [Cin][tap][Cout]).  Returns 1 when it
 * handled the shape, 0 when the shape is not covered (mscl_conv3d_fwd / _dgrad call it first and fall back). */
int mscl_conv_halo64(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                     const uint16_t* addend, float* stat_sum, float* stat_sq, void* stream);

/* dx = conv_transpose(dy, w) [+ addend]; wT_bf16 is the kernel re-laid out [Cin][kT][kH][kW][Cout]
 * (mscl_weight_transpose).  Replaces autograd's conv3d input gradient.  For a STRIDED conv addend == dx is allowed (dx += ...: an
 * element is read and written by the same wave, the read first) and the positions no tap reaches are then left alone -- a
 * 1x1x1 / stride-2 shortcut touches one position in eight.  Stride-1 convs: addend and dx must not overlap. */
int mscl_conv3d_dgrad(const mscl_conv_desc* d, const uint16_t* dy, const uint16_t* wT_bf16, uint16_t* dx,
                      const uint16_t* addend, float* splitk_ws, int64_t splitk_ws_floats, void* stream);

/* dw[Cout][taps][Cin] (fp32) += sum over positions of dy (x) x ; atomically accumulated, so the
 * caller zeroes dw once per step and repeated traversals of a shared trunk simply add up
 * (the flow encoder is traversed twice, recognizers/mscl.py:239-240).  dbias (K floats, optional)
 * += sum over positions of dy.  Replaces autograd's conv3d weight/bias gradient.  ws (optional, ws_floats fp32):
 * scratch for the window-resident layer-1 kernel (3x3x3 s1 p1, 64 -> 64), which stores per-block partial slabs and
 * reduces them in a second pass; 256 * 36864 floats cover every shape; NULL selects the general kernel.
 * INVARIANT (one stream per parameter): where a layer's positions form a single split, and in the slab-reducing kernels, dw is
 * updated by plain read-modify-write adds, not atomics -- two mscl_conv3d_wgrad calls for the SAME dw must not overlap, i.e. every
 * launch that writes one parameter's gradient is issued on one stream (or ordered by events).  The MSCL step keeps it: both
 * traversals of the shared flow trunk run on the flow stream.  The transposed kernels of mscl_weight_transpose[_batched] obey the
 * same rule with respect to the input-gradient launches that read them. */
int mscl_conv3d_wgrad(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw,
                      float* dbias, float* ws, int64_t ws_floats, void* stream);

/* Grouped form for the small layers (round 5).  On the small maps -- layers 3-4, their entries and 1x1x1 shortcuts, the pyramid
 * levels of necks/fpn.py:188-203 and necks/sepc.py:118-135 -- the weight gradient is a launch-latency-bound leaf of the backward
 * chain (8-41 us at 6 % MFMA busy, ~30 launches per step).  mscl_conv3d_wgrad_group computes the weight gradients of n <=
 * MSCL_WGRAD_GROUP_MAX layers in ONE launch (and the bias gradients of those with dbias[i] != NULL in one more): the caller defers
 * the layers' launches, keeps x[i] / dy[i] alive, and hands them over together.  descs: n descriptors; x, dy, dw, dbias: host arrays
 * of n device pointers (dbias may be NULL, or hold NULLs).  Every layer must be groupable: mscl_conv3d_wgrad_groupable(d) == 1, i.e.
 * mscl_conv3d_wgrad would take it on its general 64 x 64-tile kernel with no workspace (not in deterministic mode, whose per-split
 * slab sums run per layer).  Layers of one group that share a dw (a module applied to several pyramid levels) add with float
 * atomics; the one-stream-per-parameter INVARIANT above holds for the group as for a single call. */
#define MSCL_WGRAD_GROUP_MAX 16
int mscl_conv3d_wgrad_groupable(const mscl_conv_desc* d);
int mscl_conv3d_wgrad_group(int n, const mscl_conv_desc* descs, const uint16_t* const* x, const uint16_t* const* dy,
                            float* const* dw, float* const* dbias, void* stream);
int64_t mscl_debug_wgrad_group_launches(void);   /* test aid, as the other launch counters */

/* [Cout][taps][Cin] bf16 -> [Cin][taps][Cout] bf16 */
int mscl_weight_transpose(const uint16_t* w, uint16_t* wT, int Cout, int taps, int Cin, void* stream);
/* the same for every conv kernel of a model in ONE launch: `table` is a device array of n entries
 * {const uint16_t* w; uint16_t* wT; int32 Cout, taps, Cin, first_block;} (32 bytes each), entry i owning
 * blocks [first_block_i, first_block_{i+1}) of 256 elements; total_blocks = sum of ceil(elems_i / 256). */
int mscl_weight_transpose_batched(const void* table, int n, int total_blocks, void* stream);

/* ---- BatchNorm3d (training mode) + ReLU + residual -------------------------------------------
 * replaces nn.BatchNorm3d/ReLU/`out += residual` in r3d.py:95-127, fastonly.py:104-136.
 * stats = {sum[C], sumsq[C]} from mscl_conv3d_fwd.  Computes mean/var over `rows` positions,
 * out = relu?( (y-mean)*invstd*gamma+beta + residual ), writes mean/invstd (saved for backward)
 * and updates running_mean/var (momentum, unbiased var) and num_batches_tracked (int64) in place.
 * If res_sum != NULL the residual is itself a raw conv output normalised with its own statistics
 * (the downsample branch r3d.py:285-288): its saved mean/invstd go to res_mean/res_invstd and its
 * running buffers are updated too.
 * Evaluation mode (module.eval(), used by Recognizer3D._do_test, recognizers/recognizer3d.py:33-96): pass
 * sum = sumsq = NULL; the running statistics normalise, nothing is written (save_* may be NULL). */
typedef struct {
  const float* sum; const float* sumsq; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* save_mean; float* save_invstd;
} mscl_bn_params;

int mscl_bn_act_fwd(const uint16_t* y, const mscl_bn_params* bn,
                    const uint16_t* residual, const mscl_bn_params* res_bn,
                    uint16_t* out, int64_t rows, int C, float eps, float momentum, int relu, void* stream);

/* The same over `groups` (1 or 2) BatchNorm statistics groups (see mscl_conv3d_fwd_groups): rows [k*rows/G, (k+1)*rows/G)
 * are normalised with the k-th [slot][2][C] block of bn->sum / bn->sumsq; save_mean / save_invstd hold [G][C] floats; the
 * running statistics take the groups' momentum updates one after the other in group order and num_batches_tracked grows
 * by G -- exactly what the reference's G consecutive calls of the module do (recognizers/mscl.py:239-240). */
int mscl_bn_act_fwd_groups(const uint16_t* y, const mscl_bn_params* bn,
                           const uint16_t* residual, const mscl_bn_params* res_bn,
                           uint16_t* out, int64_t rows, int C, float eps, float momentum, int relu, int groups, void* stream);

/* backward of the above.  dz = dout * (out > 0 if relu).  Pass 1 reduces dgamma/dbeta (accumulated
 * into the fp32 gradient buffers, caller-zeroed) and keeps the sums in `scratch` (MSCL_STAT_SLOTS * 4*C
 * floats, caller-zeroed: blocks spread their partial sums over the slots); pass 2 writes dy (and dres: dz itself for an identity residual, the BN input
 * gradient for a normalised residual).
 * relu: 0 none, 1 mask by out > 0.
 * beta (optional; only with relu and no residual of either kind): the ReLU mask is recomputed as
 * gamma*invstd*(y - mean) + beta > 0, the forward's own arithmetic, and `out` is not read (may be NULL):
 * one map less per pass for the conv1 / stem BatchNorms (r3d.py:116-118, :176-184). */
int mscl_bn_act_bwd(const uint16_t* dout, const uint16_t* out, const uint16_t* y,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                    float* dgamma, float* dbeta,
                    const uint16_t* res_y, const float* res_gamma, const float* res_mean, const float* res_invstd,
                    float* res_dgamma, float* res_dbeta,
                    uint16_t* dy, uint16_t* dres, int want_identity_dres,
                    float* scratch, int64_t rows, int C, int relu, void* stream);

/* with statistics groups: save_mean / save_invstd (and the residual's) are [G][C], scratch is [G][MSCL_STAT_SLOTS][4*C];
 * dgamma / dbeta receive the sum over the groups.
 * det_parts: deterministic mode's scratch for the per-block partial sums, >= mscl_det_parts_floats(rows, C, groups, 4) floats
 * (NULL or smaller: the slots of `scratch` hold them, 16 blocks); ignored outside deterministic mode. */
int mscl_bn_act_bwd_groups(const uint16_t* dout, const uint16_t* out, const uint16_t* y,
                           const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                           float* dgamma, float* dbeta,
                           const uint16_t* res_y, const float* res_gamma, const float* res_mean, const float* res_invstd,
                           float* res_dgamma, float* res_dbeta,
                           uint16_t* dy, uint16_t* dres, int want_identity_dres,
                           float* scratch, int64_t rows, int C, int relu, int groups, float* det_parts, int64_t det_parts_floats,
                           void* stream);

/* ---- layout / elementwise ---------------------------------------------------------------------
 * frames [t_off, t_off+T) of (B,Cin<=3,T_total,H,W) fp32 NCTHW -> (B,T,H,W,8) bf16 NDHWC, channels Cin..7
 * zero (the chunk(2, dim=2) of recognizers/mscl.py:230-235 without a copy); optional per-channel
 * (x-mean)/std = the deterministic Normalize of common/ssl_aug_v2.py:66-68.  mean3/std3 are HOST
 * arrays of 3 floats (or NULL).  flip_mask (device, B bytes, or NULL): samples with a non-zero byte are mirrored
 * along W = the horizontal flip of common/ssl_aug_v2.py:107-118 given its Bernoulli draw. */
int mscl_pack_input(const float* x, uint16_t* out, int B, int Cin, int T, int H, int W, int T_total, int t_off,
                    const float* mean3, const float* std3, const uint8_t* flip_mask, void* stream);
/* the same with the clip's device address read from device memory at kernel time (*xpp): a launch captured into a HIP graph whose
 * input batch changes per replay -- the caller rewrites the pointer word, in stream order, before each replay */
int mscl_pack_input_ind(const float* const* xpp, uint16_t* out, int B, int Cin, int T, int H, int W, int T_total, int t_off,
                        const float* mean3, const float* std3, const uint8_t* flip_mask, void* stream);
/* (rows, W, 8) bf16 packed 3-channel clip -> (rows, (W+1)/2 + 1, 8): pair j = pixels 2j-1 and 2j of the row as channels
 * [3p + c], zeros outside.  Turns the RGB stem (torchvision BasicStem == backbones/r3d.py:176-184: Conv3d(3, 64, (3,7,7),
 * stride (1,2,2), padding (1,3,3))) into a (3,7,4) / stride (1,2,1) / padding (1,3,1) convolution over the pairs with
 * weights w2[co][kt][kh][j][3p + c] = w[co][c][kt][kh][2j + p] (zero for kw = 7) and identical outputs. */
int mscl_pair_w(const uint16_t* x, uint16_t* out, int64_t rows, int W, void* stream);
/* optical flow (B,2,T_total,H,W) fp32 uv -> colour-wheel image, frames [t_off, t_off+T), as (B,T,H,W,8) bf16 NDHWC
 * (channels 3..7 zero): FlowVisualizer / flow_uv_to_colors of common/ssl_aug.py:87-136 with the Middlebury wheel of
 * tools/RAFT/core/utils/flow_viz.py:19-68, including the uint8 floor; `levels` (optional, (B,T,H,W,3) bytes) receives
 * the quantised levels themselves.  flip_mask as above (the reference flips the visualised image, not the vectors). */
int mscl_flow_visualize(const float* uv, uint16_t* out, uint8_t* levels, int B, int T, int H, int W, int T_total, int t_off,
                        const uint8_t* flip_mask, void* stream);
/* Flow Rotation Augmentation + visualiser in one pass over raw flow: NormFlowWithStidedAug of
 * datasets/pipelines/transforms_motion.py:7-29,103-142 (per-frame division by max radius + 1e-5; a second copy rotated
 * by (ratio_lo + (ratio_hi - ratio_lo) / num_chunks * cid[b]) * pi first) followed by FlowVisualizer.  uv (B,2,T,H,W)
 * fp32, cid (B) int32 on the device -> out (B,2T,H,W,8) bf16: frames [0,T) base, [T,2T) rotated (merge_aug=True).
 * Optional outputs: levels (B,2T,H,W,3) bytes, normed (B,2T,H,W,2) fp32 = the normalised vectors.  scratch: 2*B*T
 * doubles. */
int mscl_flow_fra_visualize(const float* uv, const int32_t* cid, float ratio_lo, float ratio_hi, int num_chunks,
                            uint16_t* out, uint8_t* levels, float* normed, double* scratch, int B, int T, int H, int W,
                            const uint8_t* flip_mask, void* stream);
/* Colour augmentation of an RGB clip GIVEN its sampled parameters (the arithmetic of the kornia ops chained in
 * common/ssl_aug_v2.py:31-43: ColorJitter(0.4,0.4,0.4,0.1) then RandomGrayscale; parameter sampling is the host's job).
 * x, out (B,3,T,H,W) fp32 in [0,1]; params (B,16) fp32 on the device, per sample: [0] jitter on/off, [1..4] order of the
 * four ops (0 brightness, 1 contrast, 2 saturation, 3 hue), [5] brightness factor f (x + f - 1, clamped), [6] contrast
 * factor (x * f, clamped), [7] saturation factor (HSV s * f, clamped), [8] hue shift in radians, [9] grayscale on/off
 * (0.299 R + 0.587 G + 0.114 B in all three channels), [10] Gaussian sigma for mscl_gauss_blur (0 = none). */
int mscl_color_aug(const float* x, float* out, const float* params, int B, int T, int H, int W, void* stream);
/* Separable ksize x ksize Gaussian blur with reflect border of every (H,W) frame of the samples whose params[b][10]
 * (sigma) is > 0, a copy for the others: GaussianBlur of common/ssl_aug.py:163-171 (kornia GaussianBlur2d, taps
 * exp(-d^2 / 2 sigma^2) normalised to sum 1).  x, tmp, out (B,frames,H,W) fp32; out may alias x, tmp may not.
 * ksize odd, <= 33, H and W > ksize / 2. */
int mscl_gauss_blur(const float* x, float* tmp, float* out, const float* params, int ksize, int B, int frames, int H, int W,
                    void* stream);
/* ---- input data path: paired crop + resize + normalise of decoded clips, one pass -----------------
 * replaces datasets/pipelines/moco_augmentations.py:110-163 (the crop of MoCoRandomResizedCrop; the box is drawn on the
 * host), :236-321 (MoCoResize = mmcv.imresize = cv2.resize INTER_LINEAR) and :324-354 (MoCoNormalize: / 255, HWC -> CTHW).
 * src: raw frames (B,T,Hs,Ws,3) uint8; boxes: (B,4) int32 {x1, y1, x2, y2} (exclusive ends, inside the frame);
 * out: (B,3,T,Ho,Wo) fp32 in [0,1].  cv2's fixed-point arithmetic (11-bit coefficients, its 8-bit vertical pass, the 2 x 2
 * area special case) restated from OpenCV's published resize.cpp; OpenCV is not vendored in the reference: unpinned. */
int mscl_crop_resize_u8(const uint8_t* src, const int32_t* boxes, float* out, int B, int T, int Hs, int Ws, int Ho, int Wo,
                        int64_t src_batch_stride, void* stream);
/* the same for float maps with C <= 16 channels (the (u, v) flow after NormFlowWithStidedAug, ori_flow=True: no / 255):
 * src (B,T,Hs,Ws,C) fp32 -> out (B,C,T,Ho,Wo) fp32, float bilinear taps in cv2's order (horizontal, then vertical).
 * src_batch_stride (both): elements between consecutive samples of src, 0 = dense; lets one upload that holds the frames
 * of both views (or base and rotated flow) be read as two T-frame sources. */
int mscl_crop_resize_f32(const float* src, const int32_t* boxes, float* out, int B, int T, int Hs, int Ws, int C, int Ho, int Wo,
                         int64_t src_batch_stride, void* stream);
/* out = relu?(a + b + c) elementwise bf16 (b, c optional) */
int mscl_add_relu(const uint16_t* a, const uint16_t* b, const uint16_t* c, uint16_t* out, int64_t n, int relu, void* stream);
/* din = dout * (out > 0) */
int mscl_relu_bwd(const uint16_t* dout, const uint16_t* out, uint16_t* din, int64_t n, void* stream);
/* nearest / trilinear (align_corners=False) resize-and-add of NDHWC maps:
 * dst[n,t,h,w,:] (+)= interp(src); replaces F.interpolate in necks/fpn.py:188-203 (nearest) and
 * necks/sepc.py:125-129 (trilinear).  accumulate=0 overwrites.  The *_bwd forms scatter-add the
 * gradient back to the coarse map (gather form, deterministic). */
int mscl_upsample_add(const uint16_t* src, uint16_t* dst, int N, int Ts, int Hs, int Ws, int Td, int Hd, int Wd,
                      int C, int trilinear, int accumulate, void* stream);
int mscl_upsample_bwd(const uint16_t* ddst, uint16_t* dsrc, int N, int Ts, int Hs, int Ws, int Td, int Hd, int Wd,
                      int C, int trilinear, void* stream);

/* mean over the middle axis: x (outer, inner, C) bf16 -> out (outer, C) fp32.
 * AdaptiveAvgPool3d((1,1,1)) of necks/base.py:17-21 and ((None,1,1)) of heads/local_cl_head.py:23-24 */
int mscl_pool_fwd(const uint16_t* x, float* out, int outer, int inner, int C, void* stream);
/* dx (outer, inner, C) bf16 = dout (outer, C) / inner, broadcast; accumulate into dx if accumulate */
int mscl_pool_bwd(const float* dout, uint16_t* dx, int outer, int inner, int C, int accumulate, void* stream);

/* max-pool (1,3,3) / stride (1,2,2) / pad (0,1,1) on an NDHWC bf16 map with NT = N*T planes of H x W x C:
 * nn.MaxPool3d of backbones/resnet3d.py:461-467 (ResNet3dSlowOnly pool1, pool1_stride_t = 1) and of
 * backbones/fastonly.py:222-235 (r2d_50 BottleneckStem).  out is (NT, Ho, Wo, C) with Ho = (H-1)/2+1; `win` holds one
 * uint32 per 8 output channels: the winning tap (0..8, row-major over the window; ties to the first, as torch) of each
 * channel in 4 bits.  The backward gathers (no atomics, bit-reproducible): dx = sum over the windows a pixel won. */
int mscl_maxpool_hw_fwd(const uint16_t* x, uint16_t* out, uint32_t* win, int NT, int H, int W, int C, void* stream);
int mscl_maxpool_hw_bwd(const uint16_t* dout, const uint32_t* win, uint16_t* dx, int NT, int H, int W, int C, void* stream);

/* ---- projection MLP: Linear(+ReLU) on a handful of rows (recognizers/moco.py:367-372) ---------- */
int mscl_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in_f, int out_f, int relu, void* stream);
/* dx = (dy*mask) W ; dw += dy^T x ; db += colsum(dy); mask = (y>0) if relu */
int mscl_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                    int rows, int in_f, int out_f, int relu, void* stream);
/* F.normalize(dim=1, eps=1e-12) and its backward (recognizers/moco.py:528-529) */
int mscl_l2norm_fwd(const float* x, float* y, float* norms, int rows, int dim, void* stream);
int mscl_l2norm_bwd(const float* y, const float* norms, const float* dy, float* dx, int rows, int dim, void* stream);

/* ---- MoCo contrastive pass over the negative queue ---------------------------------------------
 * replaces recognizers/moco.py:481-498 + heads/moco_head.py:38-77 + losses/cross_entropy_loss.py:134-138
 * + core/evaluation/accuracy.py:130-149 (and heads/moco_head_v2.py:38-53 for the cross-modal rows).
 * queue is the reference buffer (dim, K) fp32, count (K) int64.  For each of R query rows
 * (q: R x dim fp32, pos_logit: R floats = q.k_pos, *not yet* divided by T) computes over the aged
 * snapshot W = queue * 0.99999^count without materialising it:
 *   part[blk][r] = {max, sum exp(l-max), #negatives with logit > pos logit}   (pass 1, one chunk of 128 queue columns per block:
 *   nblk = ceil(K / 128); K must be even -- a lane reads two columns of a queue row as one 8-byte piece)
 *   then mscl_nce_finish: lse, loss_r = lse - pos/T, rank_r, and probabilities' normaliser.
 * Pass 2 (mscl_nce_bwd) re-streams the queue and accumulates dq[r] = (1/T) sum_k softmax_k * W[:,k]
 * (the positive-key term is added by the caller's tiny kernel mscl_nce_pos_bwd); ws = scratch for the per-block
 * partial sums, at least ceil(K / 128) * roundup(min(R, 32), 8) * dim floats.
 * Any R: rows beyond 32 run as further 32-row tiles over the same snapshot (3 row groups of a per-GPU batch of 32 = 96
 * rows); part then holds [tile][blk][rows of the tile], R * nblk * 3 floats in all. */
/* The *_virt forms read the snapshot the reference takes AFTER `_dequeue_and_enqueue(new_keys)` (moco.py:423-440: every age +1,
 * the n_new columns from *queue_ptr on replaced by new_keys (n_new x dim fp32) at age 1) from the buffers as they stand BEFORE
 * that write -- the same arithmetic on the same values, so the pass on the later snapshot (the rotated-flow and rf terms of
 * mscl.py:255-261 read the flow queue after the base pass's enqueue) can run beside the pass on the earlier one.  new_keys = NULL:
 * the plain forms. */
int mscl_nce_fwd_virt(const float* queue, const int64_t* count, const float* q, const float* pos_logit, float* part,
                      int R, int dim, int K, float inv_T, const float* new_keys, int n_new, const int64_t* queue_ptr, void* stream);
/* mscl_nce_bwd_virt, kpos / pos_logit (both or neither; NULL in the plain form): the positive pair's term
 * dq[r] += row_scale[r] * inv_T * (softmax_pos[r] - 1) * kpos[r] is added in the same launch that sums the per-block partial
 * sums (the arithmetic of mscl_nce_pos_bwd; one launch less per pass on the step's serial loss phase). */
int mscl_nce_bwd_virt(const float* queue, const int64_t* count, const float* q, const float* lse, const float* row_scale,
                      float* dq, float* ws, int64_t ws_floats, int R, int dim, int K, float inv_T,
                      const float* new_keys, int n_new, const int64_t* queue_ptr, const float* kpos, const float* pos_logit,
                      void* stream);
int mscl_nce_fwd(const float* queue, const int64_t* count, const float* q, const float* pos_logit,
                 float* part, int R, int dim, int K, float inv_T, void* stream);
int mscl_nce_finish(const float* part, const float* pos_logit, float* lse, float* loss_rows, int32_t* rank,
                    int R, int nblk, float inv_T, void* stream);
int mscl_nce_bwd(const float* queue, const int64_t* count, const float* q, const float* lse,
                 const float* row_scale, float* dq, float* ws, int64_t ws_floats, int R, int dim, int K, float inv_T,
                 void* stream);

/* The step's log vector (17 or 23 floats, MSCLWithAug's key order, last entry = sum of the loss entries) from the per-row
 * results of the three queue passes and the LMCL kernel: mean loss / top-1 / top-5 per InfoNCE group
 * (heads/moco_head.py:60-77, core/evaluation/accuracy.py:130-149 as a rank count) and recognizers/base.py:287-298.
 * rank/loss arrays hold nA (A), 1 (B), nC (C) groups of B rows; nA == nC in {2, 3}. */
int mscl_step_logs(const int32_t* rankA, const float* lossA, const int32_t* rankB, const float* lossB,
                   const int32_t* rankC, const float* lossC, const float* lmcl_sum, const int32_t* lmcl_hits, int B,
                   int nA, int nC, float w_intra, float n_rows, float* logs, void* stream);

/* pos[r] = <a[r], b[r]> (l_pos, recognizers/moco.py:481) and the positive-key term of the query gradient:
 * dq[r] += row_scale[r] * inv_T * (softmax_pos[r] - 1) * kpos[r] */
int mscl_rowdot(const float* a, const float* b, float* out, int rows, int dim, void* stream);
int mscl_nce_pos_bwd(const float* kpos, const float* pos, const float* lse, const float* row_scale, float* dq,
                     int R, int dim, float inv_T, void* stream);

/* Loss-phase layout in two launches (recognizers/mscl.py:239-261, heads/moco_head_v2.py:38-53, local_cl_head.py:59): pack the
 * query / key rows of the RGB-queue pass (A) and the post-enqueue flow-queue pass (C), their row scales and the LMCL flow frames
 * into one workspace  QA[n B D] KA[n B D] QC[n B D] KC[n B D] sA[n B] sC[n B] ones[B] flow[B 2t Cf]  (n = 3 with use_aug, else 2);
 * unpack: out = dq_rgb[B D] dq_fb[B D] dq_fa[B D] dp_rgb[B t C] dp_fb[B t Cf] dp_fa[B t Cf] from the passes' query gradients. */
/* pos (NULL: not wanted): the positive logits <query row, key row> of every row of the three passes, [n B] of pass A, [B] of the
 * pre-enqueue flow-queue pass B (q_fb . k_fb), [n B] of pass C -- the arithmetic of mscl_rowdot, in the pack launch. */
int mscl_loss_pack(const float* q_rgb, const float* q_fb, const float* q_fa, const float* k_rgb, const float* k_fb,
                   const float* k_fa, const float* p_fb, const float* p_fa, float* ws, float* pos, int B, int D, int t, int Cf,
                   int use_aug, float w_intra, void* stream);
int mscl_loss_unpack(const float* dA, const float* dB, const float* dC, const float* dpr, const float* dpf, float* out,
                     int B, int D, int t, int C, int Cf, int use_aug, void* stream);

/* queue bookkeeping, bit-exact int64: count += 1; queue[:, ptr:ptr+n] = keys^T; count[ptr:ptr+n] = 1;
 * ptr = (ptr+n) % K.   recognizers/moco.py:423-440.  keys: (n, dim) fp32, ptr: int64[1] on device.
 * One launch (the block that finishes last moves the pointer, through a library-owned ticket): calls on DIFFERENT streams must
 * not overlap in time -- the step orders its enqueues on one stream, as the reference orders them. */
int mscl_queue_enqueue(float* queue, int64_t* count, int64_t* ptr, const float* keys, int n, int dim, int K, void* stream);

/* ---- LMCL (heads/local_cl_head.py:57-73,41-55) ------------------------------------------------
 * rgb (B, t, C) and flow (B, 2t, C) are spatially pooled features (fp32).  Per clip: L2-normalise
 * over C, sim = rgb . flow^T / T, CE against label j for row j, top-1/top-5 hit counts.
 * Outputs: loss_sum (1 float, += sum of row losses), hits (2 int32: top1, top5 += hits),
 * drgb / dflow = gradient of mean-over-(B*t) loss wrt the pooled features. */
int mscl_lmcl(const float* rgb, const float* flow, float* loss_sum, int32_t* hits, float* drgb, float* dflow,
              int B, int t, int C, float inv_T, void* stream);

/* ---- parameter-sized elementwise passes --------------------------------------------------------
 * key-encoder EMA (recognizers/moco.py:408-421): pk = m*pk + (1-m)*pq, and refresh pk's bf16 shadow */
int mscl_ema_update(float* pk, const float* pq, uint16_t* pk_bf16, int64_t n, float m, void* stream);
/* same with m read from device memory, so a captured HIP graph can follow the cosine momentum schedule */
int mscl_ema_update_dev(float* pk, const float* pq, uint16_t* pk_bf16, int64_t n, const float* m_dev, void* stream);
/* *out = sum of squares of g (fp32, overwritten): first half of clip_grad_norm_.  Bit-reproducible (no atomics): every
 * data-parallel replica derives the same clip coefficient from the same all-reduced gradient.  `partials` is caller
 * scratch of n_partials floats (one per block of the first phase; 1024 is the most that is used). */
int mscl_sumsq(const float* g, float* out, int64_t n, float* partials, int n_partials, void* stream);
/* second half + torch.optim.SGD (momentum, dampening 0, no nesterov), mmcv OptimizerHook wiring at
 * mmaction/apis/train.py:111-119: coef = min(1, max_norm/(sqrt(*sumsq)+1e-6)) (coef = 1 when max_norm <= 0);
 * g = g*coef + wd*p; buf = first ? g : mom*buf + g; p -= lr*buf; p_bf16 = bf16(p).
 * `first` selects the buffer initialisation of the very first step (buf = g). */
int mscl_sgd_step(float* p, const float* g, float* buf, uint16_t* p_bf16, int64_t n, const float* sumsq,
                  float max_norm, float lr, float momentum, float wd, int first, void* stream);
/* lr read from device memory (graph replay); momentum buffers must be zero-initialised */
int mscl_sgd_step_dev(float* p, const float* g, float* buf, uint16_t* p_bf16, int64_t n, const float* sumsq,
                      float max_norm, const float* lr_dev, float momentum, float wd, void* stream);
int mscl_cast_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
