// Window-resident 3x3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (R3D-18 layer 1: 46 % of the
// step's FLOPs; forward AND stride-1 input gradient) on bf16 MFMA, gfx950.
// Reference op: the 3x3x3 convolutions of layer1's BasicBlocks (mmaction/models/backbones/r3d.py:16-34,95-127) and their
// autograd input gradient.
//
// Why a second conv kernel: the implicit-GEMM kernels (conv_igemm.hip) re-stage every input row once per tap, 27x.
// With only 64 output channels a staged byte feeds 64 FLOP; a CU moves global memory into LDS at <= 64 B/clk, the
// MFMA pipe wants 4096 FLOP/clk, so re-staging caps those kernels near 40 % of MFMA peak (measured 23 %).
//
// Here a block owns BM = 256 consecutive positions of ONE (n, t) plane in PADDED-LINEAR order: the plane is walked
// as H rows of W + 2 columns (one zero column each side), q = hp * (W + 2) + wp.  In that order every tap is a
// constant shift of q -- (kh - 1) * (W + 2) + (kw - 1) -- and the zero padding is part of the data, so there is no
// per-tap masking at all.  Rows that fall on padding or outside the clip are zero-filled by the buffer unit's
// out-of-range rule.  The 2 pad columns cost 3.4 % extra MFMA work at W = 56.
//
// TWO blocks per CU (round 4; the one-block-per-CU form of rounds 1-3 -- a 128-KB block holding two window slots and a 4-stage
// ring -- measured 108-111 / 95-96 us against 82 / 77-80 us and was deleted in round 5 with its switches MSCL_HALO_BLOCKS /
// MSCL_HALO_RING / MSCL_HALO_TPS; what a one-block form does besides its 27 taps -- dispatch gap, the exposed 48-KB window
// prologue, the statistics epilogue, the output stores -- was 43 % of a block slot).  A block is small enough for a second one
// beside it (80 KB of LDS, <= 128 registers per lane), so one block's prologue / epilogue / plane switches run under the
// other's taps:
//  * ONE window slot: the three source planes are staged one after the other into the same 48 KB (384 rows of 128 B); a plane
//    switch (taps 8 -> 9, 17 -> 18) is [all waves done with the old plane | 6 DMA pieces per thread | second k step of the old
//    tap | landed | barrier];
//  * half-tap operand pipeline: the fragments of (tap, k step 1) are read while (tap, k step 0) multiplies and those of
//    (tap + 1, k step 0) while (tap, k step 1) multiplies -- two sets of 6 fragments: 48 operand registers;
//  * weights (8 KB per tap) stream through an LDS ring of RING = 2 stages of TPS = 2 taps by LDS-DMA; ONE barrier per STAGE
//    publishes the next stage and frees the one just used: 17 barriers per block (one per tap, 31, measured 2 % slower; ring
//    depths 2 / 3 / 4 of one-tap stages measured alike);
//  * 8 waves = two per SIMD, 32 positions x 64 channels each, 12 fragment reads per 16 MFMAs (4 waves of 64 x 64 measured a tie);
//  * epilogue: BatchNorm sum / sum-of-squares as a reduce-scatter over the lane rows (common.h), optional addend, bf16,
//    16-byte stores.
#include "common.h"
#include <cstdlib>

struct HaloGeom {
  int N, T, H, W, HW, Wp;          // Wp = W + 2 padded row length
  int tiles, mode;                 // tiles per plane; mode 0 forward, 1 input gradient (taps mirrored)
  FastDiv dWp;                     // division by Wp
  FastDiv dT, dTiles;              // persistent kernel: item -> (chain, t), chain -> (n, tile)
  int stat_stride;                 // floats between two statistics slots (2 * C: forward [slot][2][C]; 4 * C: BatchNorm-backward scratch)
};

__device__ __forceinline__ auto halo_rsrc(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}


constexpr int HBM = 256;           // padded-linear positions per block
constexpr int HC = 64;             // channels (in = out)
constexpr int NH = 384;            // window rows per source plane (HBM + 2 * Wp + 2 <= NH, i.e. W <= 61)
constexpr unsigned HOOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NW = 8;              // waves per block (two per SIMD)
constexpr int RING = 2, TPS = 2;   // weight ring: RING stages of TPS taps
// KT = temporal taps: 3 (pad 1: R3D-18 layer 1) or 1 (pad 0: the 1x3x3 conv2 of the SlowOnly-50 / r2d Bottlenecks at 64 channels -- one
// source plane, nine taps, no plane switch; round 6)
template <int KT>
__global__ __launch_bounds__(64 * NW, NW / 2) void conv_halo64b_kernel(const HaloGeom g, const bf16_t* __restrict__ src,
                                                              const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                                              float* __restrict__ stat_sq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, NT = 9 * KT;
  constexpr int RPP = 8 * NW;                              // window rows per DMA pass
  constexpr int NHK = (BM + 128 + RPP - 1) / RPP * RPP;    // 384 window rows (BM + 2 * (W + 2) + 2, W <= 61)
  constexpr int PLANE = NHK * 128;
  constexpr int NPS = NHK / RPP;                           // 6 DMA pieces per thread per plane
  constexpr int WPOS = BM / NW, IM = WPOS / 16;            // 32 positions per wave, 2 position tiles
  constexpr int WP = 8 / NW;                               // DMA instructions per weight tile per wave
  static_assert(RING * TPS <= 4, "ring fits the 80-KB block");
  unsigned char* const Hs = smem;                          // [NHK][128 B] the one window slot, row j <-> q0 - Wp - 1 + j
  unsigned char* const Ws = smem + PLANE;                  // [RING][TPS taps][64][128 B] weight ring (a stage = TPS taps)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = bid % g.tiles, plane = bid / g.tiles;   // plane = n*T + t
  const int t = plane % g.T;
  const int q0 = g.Wp + tile * BM;
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);
  const auto rs_src = halo_rsrc(src, 0x7FFFFFFFu);
  const auto rs_wgt = halo_rsrc(wgt, 0x7FFFFFFFu);

  unsigned win_voff[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int j = ps * RPP + (tid >> 3), pg = tid & 7;
    const int lg = pg ^ (j & 7);                           // source-side swizzle keyed on the window row
    const int q = q0 - g.Wp - 1 + j;
    const int hp = fdiv(q < 0 ? 0 : q, g.dWp), wp = q - hp * g.Wp;
    const bool ok = q >= 0 && hp >= 1 && hp <= g.H && wp >= 1 && wp <= g.W;
    win_voff[ps] = ok ? (unsigned)((((hp - 1) * g.W + (wp - 1)) * HC + lg * 8) * 2) : HOOB;
  }
  auto issue_plane = [&](int hp) {                         // source plane t + hp - KT / 2 into the window slot (rows outside the clip: zeros)
    const bool okp = (unsigned)(t + hp - KT / 2) < (unsigned)g.T;
    const unsigned so = __builtin_amdgcn_readfirstlane(okp ? (unsigned)((plane + hp - KT / 2) * g.HW) * (HC * 2) : 0u);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const unsigned vo = okp ? win_voff[ps] : HOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(Hs + (ps * 64 * NW + wave * 64) * 16), 16, vo, so, 0, 0);
    }
  };
  const int w_row = tid >> 3, w_lg = tid & 7;
  const unsigned w_voff0 = (unsigned)((w_row * NT * HC + (w_lg ^ (w_row & 7)) * 8) * 2);     // source-side swizzle
  auto issue_weights = [&](int tap) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(tap * HC * 2));
    unsigned char* dst = Ws + (((tap / TPS) % RING) * TPS + tap % TPS) * (64 * 128) + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(dst), 16, w_voff0, so, 0, 0);
  };
  auto tap_plane = [&](int kt) { return mode ? KT - 1 - kt : kt; };
  auto tap_shift = [&](int kh, int kw) { return mode ? (2 - kh) * g.Wp + (2 - kw) : kh * g.Wp + kw; };

  // ---- prologue: first plane, the first RING weight stages ----
  issue_plane(tap_plane(0));
#pragma unroll
  for (int w0 = 0; w0 < RING * TPS; ++w0) issue_weights(w0);

  const int fr = lane & 15, fq = lane >> 4;
  const int arow0 = wave * WPOS + fr;
  const int b_addr[2] = {fr * 128 + ((0 + fq) ^ (fr & 7)) * 16, fr * 128 + ((4 + fq) ^ (fr & 7)) * 16};

  bf16x8_t fa[2][IM], fb[2][4];                            // [set][fragment]: set s holds the operands of one half tap
  auto read_half = [&](int tap, int ks, int set) {
    const int kh = (tap % 9) / 3, kw = tap % 3;
    const unsigned char* wb = Ws + (((tap / TPS) % RING) * TPS + tap % TPS) * (64 * 128) + b_addr[ks];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8_t*>(wb + j * 2048);
    const int row = arow0 + tap_shift(kh, kw);
    const unsigned char* hb = Hs + row * 128 + (((ks * 4 + fq) ^ (row & 7)) * 16);
#pragma unroll
    for (int i = 0; i < IM; ++i) fa[set][i] = *reinterpret_cast<const bf16x8_t*>(hb + i * 2048);
  };
  f32x4_t acc[4][IM];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], acc[j][i], 0, 0, 0);
  };
#define HB_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory")
#define HB_WAIT_N(n) do { if ((n) >= 8) HB_WAIT(8); else if ((n) == 7) HB_WAIT(7); else if ((n) == 6) HB_WAIT(6); else if ((n) == 5) HB_WAIT(5); \
    else if ((n) == 4) HB_WAIT(4); else if ((n) == 3) HB_WAIT(3); else if ((n) == 2) HB_WAIT(2); else if ((n) == 1) HB_WAIT(1); else HB_WAIT(0); } while (0)
  // ---- a ring stage = TPS taps: one barrier per STAGE (it publishes the next stage and frees the one just used) instead of one per
  // tap -- 17 barriers per block instead of 31; the DMA of a stage has a whole stage of MFMAs to land under ----
  constexpr int NST = (NT + TPS - 1) / TPS;
  auto taps_in = [](int st) { return st < 0 || st >= NST ? 0 : (NT - st * TPS < TPS ? NT - st * TPS : TPS); };
  auto issue_stage = [&](int st) {
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) if (st < NST && st * TPS + tt < NT) issue_weights(st * TPS + tt);
  };
  {   // plane + stage 0 landed; stages 1 .. RING - 1 stay in flight
    int infl = 0;
#pragma unroll
    for (int st = 1; st < RING; ++st) infl += WP * taps_in(st);
    HB_WAIT_N(infl);
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_half(0, 0, 0);
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    read_half(tap, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < NT) {
      const bool edge = (tap + 1) % TPS == 0;              // tap + 1 opens a new stage
      const int st_e = tap / TPS;                          // the stage tap belongs to
      if ((tap + 1) % 9 == 0) {
        // plane switch: every wave holds the last fragments of the old plane in registers
        HB_WAIT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_plane(tap_plane((tap + 1) / 9));
        if (edge) issue_stage(st_e + RING);                // (the stage just used is free behind the barrier too)
        __builtin_amdgcn_sched_barrier(0);
        mma(1);                                            // the old tap's second k step runs under the DMA
        __builtin_amdgcn_sched_barrier(0);
        HB_WAIT_N(edge ? WP * taps_in(st_e + RING) : 0);   // the plane has landed; the stage issued behind it may stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_half(tap + 1, 0, 0);
        continue;
      }
      if (edge) {
        // stage st_e + 1 landed (the younger stages st_e + 2 .. st_e + RING - 1 may stay in flight); this wave's reads of stage st_e retired
        int infl = 0;
#pragma unroll
        for (int st = st_e + 2; st <= st_e + RING - 1; ++st) infl += WP * taps_in(st);
        HB_WAIT_N(infl);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_stage(st_e + RING);
      }
      read_half(tap + 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HB_WAIT_N
#undef HB_WAIT
  __syncthreads();                  // the epilogue reuses the window memory

  long orow[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int q = q0 + wave * WPOS + i * 16 + fr;
    const int hp = fdiv(q, g.dWp), wp = q - hp * g.Wp;
    const bool ok = hp <= g.H && wp >= 1 && wp <= g.W;
    orow[i] = ok ? ((long)plane * g.HW + (hp - 1) * g.W + (wp - 1)) * HC : -1;
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- epilogue: BatchNorm statistics (as conv_halo64_kernel) ----
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [NW][2][64]
    const int wv = tid >> 6;
    // value 8 j + 4 sq + r = (channel tile j, sum / sum of squares, channel r of this lane row's quad); after the reduce-scatter
    // (common.h) quad q of a lane row holds the 8 totals of channel tile q: 64 DPP moves per wave instead of 128, two LDS stores
    // by four lanes of a row instead of eight by one
    float sv[32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IM; ++i) { const float v = acc[j][i][r]; s1 += v; s2 += v * v; }
        sv[j * 8 + r] = s1; sv[j * 8 + 4 + r] = s2;
      }
    row16_reduce_scatter<32>(sv);
    if ((fr & 3) == 0) {
      const int jq = fr >> 2;
      *reinterpret_cast<float4*>(&red[(wv * 2 + 0) * HC + jq * 16 + fq * 4]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      *reinterpret_cast<float4*>(&red[(wv * 2 + 1) * HC + jq * 16 + fq * 4]) = make_float4(sv[4], sv[5], sv[6], sv[7]);
    }
    __syncthreads();
    if (tid < 2 * HC) {
      float tsum = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) tsum += red[w8 * 2 * HC + tid];
      const int so = (int)(blockIdx.x % MSCL_STAT_ACTIVE) * g.stat_stride;
      atomicAdd(tid < HC ? &stat_sum[so + tid] : &stat_sq[so + tid - HC], tsum);
    }
  }
  uint2 add4[4][IM];
  if (addend != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
        add4[j][i] = *reinterpret_cast<const uint2*>(addend + (orow[i] < 0 ? 0 : orow[i]) + j * 16 + fq * 4);
  }
  // 16-byte stores: two channel tiles paired through v_permlane16_swap (igemm.h: an even lane row ends up with 8 consecutive
  // channels of tile j, an odd one with 8 of tile j + 1; the partner row holds the same position)
  auto quad = [&](int i, int j) -> uint2 {
    float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
    if (addend != nullptr) {
      const uint2 av = add4[j][i];
      v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
      v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
    }
    uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
    return pv;
  };
#pragma unroll
  for (int i = 0; i < IM; ++i) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const uint2 q0 = quad(i, j), q1 = quad(i, j + 1);
      const auto sx = __builtin_amdgcn_permlane16_swap(q0.x, q1.x, false, false);      // every lane takes part (no divergent branch around)
      const auto sy = __builtin_amdgcn_permlane16_swap(q0.y, q1.y, false, false);
      const int n = (j + (fq & 1)) * 16 + (fq & 2) * 4;
      if (orow[i] >= 0) *reinterpret_cast<uint4*>(out + orow[i] + n) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
  }
}

// (A persistent plane-walking variant of this kernel -- one block per CU walking 192-position items, two of three planes kept
// resident, statistics once per block -- was measured in rounds 1 and 2: faster alone on the forward conv, 112 vs 125-134 us at
// the time, slower inside the three-stream step, 785-793 vs 804-808 clip-pairs/s, and was removed in round 3.  Its successor, a
// window-resident ping-pong kernel with the K split between SIMD partners (round 3), showed the same pattern -- 101 vs 107 us alone,
// 970 vs 1007 clip-pairs/s in the step -- and was removed in round 4.)

static long g_halo_launches = 0;
extern "C" int64_t mscl_debug_halo_launches(void) { return g_halo_launches; }      // tests: which kernel family took a launch

// returns 1 if launched, 0 if the shape is not covered (caller falls back to the implicit-GEMM kernel), <0 / >0 on error
static int halo_launch(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                       const uint16_t* addend, float* ssum, float* ssq, int stat_stride, void* stream);

extern "C" int mscl_conv_halo64(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                                const uint16_t* addend, float* ssum, float* ssq, void* stream) {
  return halo_launch(d, mode, src, w, out, addend, ssum, ssq, 2 * HC, stream);
}

// (Round 1-3 kept a fused form of the input gradient, mscl_conv_halo64_dgrad_bn: the backward reduce of the BatchNorm + ReLU that
// consumes the gradient computed in this kernel's epilogue.  It broke even at best -- the epilogue's two extra map reads cost 38 us at
// one block per CU against 36-42 us saved in the BatchNorm pass -- and was removed in round 4 with its plumbing.)
static int halo_launch(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                       const uint16_t* addend, float* ssum, float* ssq, int stat_stride, void* stream) {
  if (!d || !src || !w || !out) return MSCL_E_ARG;
  if (d->C != HC || d->K != HC || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pH != 1 || d->pW != 1) return 0;
  if (!((d->kT == 3 && d->pT == 1) || (d->kT == 1 && d->pT == 0))) return 0;
  HaloGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.HW = d->H * d->W; g.Wp = d->W + 2;
  if (HBM + 2 * g.Wp + 2 > NH || (long)d->N * d->T * g.HW * HC * 2 >= (1L << 31)) return 0;
  g.tiles = (d->H * g.Wp + HBM - 1) / HBM; g.mode = mode;
  g.dWp = make_fastdiv(g.Wp);
  g.dT = make_fastdiv(d->T); g.dTiles = make_fastdiv(g.tiles);
  g.stat_stride = stat_stride;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const unsigned nblk = (unsigned)(d->N * d->T * g.tiles);
  const size_t lds = (size_t)NH * 128 + (size_t)RING * TPS * 64 * 128;
  if (d->kT == 3) hipLaunchKernelGGL(conv_halo64b_kernel<3>, dim3(nblk), dim3(64 * NW), lds, (hipStream_t)stream, g, src, w, out, addend, ssum, ssq);
  else hipLaunchKernelGGL(conv_halo64b_kernel<1>, dim3(nblk), dim3(64 * NW), lds, (hipStream_t)stream, g, src, w, out, addend, ssum, ssq);
  MSCL_LAUNCH_CHECK();
  ++g_halo_launches;
  return 1;
}
