// Weight gradient of the RGB stems with the input window resident in LDS -- the twin of conv_stem.hip (round 6): nn.Conv3d(3, 64,
// (kT, 7, 7), stride (sT, 2, 2), padding (pT, 3, 3)) of r3d.py:176-184 (kT = 3) and resnet3d.py conv1 (kT = 1; mscl_r50_cosm_lr3e-2.py:18:
// kT = 5, sT = 2), executed on the W-PAIRED clip as (kT, 7, 4) / (sT, 2, 1) / (pT, 3, 1) over pairs of 8 channels.
// Reference op: the weight gradient autograd computes for that conv.
//
//   dW[co][s = kt * 7 + kh][p * 8 + ch] = sum over (n, to, oh, ow) of dy[n, to, oh, ow][co] * x[n, to * sT + kt - pT, 2 oh + kh - 3, ow + p - 1][ch]
//
// Why its own kernel: the general kernel (conv_wgrad.hip) gathers, per k step of 32 reduction rows, the 64 bytes every position
// contributes to one (kt, kh) row of the kernel from global memory into LDS -- the last conv stage of the step that still paid one
// gather per tap: 76 us = 0.18 of the MFMA peak on the R3D-18 stem, 550 us on the (5,7,7) one.  Here a block walks 256-position tiles
// of output planes; per tile it stages what conv_stem.hip stages (the <= 17 source rows the tile can reach, all kT planes, 16 KB each:
// 256 positions x 64 bytes of operand per k step come out of it as constant shifts) plus the dy tile (32 KB), and runs every k step
// s against them:  D_s[64 co][32] += dy^T[64 x 256] . X_s[256 x 32],  X_s[pos] = the 64 contiguous window bytes at
// (row 2 (oh - oh0) + kh, column ow) of plane kt.
//  * both operands are position-major, so MFMA fragments are transposing reads (ds_read_b64_tr_b16, as conv_wgrad_halo.hip /
//    conv_thin.hip): reduction slot (lane group g, read h, row j) is position 32 ks + 16 h + 4 g + j for both;
//  * the 8 waves share the k steps s out (wave w: s = w, w + 8, ...: 3 / 5 of the 21 / 35), every wave reads the whole dy tile; the
//    partial products (<= 5 x 64 x 32 per wave) stay in registers over all tiles of the block;
//  * at the end every block stores its slab in register order with 16-byte stores and a second kernel adds the slabs into dW in a
//    fixed order (16 waves share the slabs of 64 float4 out and meet in LDS): the same bits every run, deterministic mode takes it as is.
#include "common.h"

struct StemWGeom {
  int N, T, To, H, Ho, Wo, WpS, WPL;   // WpS = source pairs per row, WPL = window pairs per row (WpS + 2)
  int HoWo, tiles, sT, pT, total;      // total = N * To * tiles items
  unsigned plane_bytes;                // one source plane: H * WpS * 16
  FastDiv dWo, dWPL, dTo, dTiles;
};

typedef __attribute__((address_space(3))) void* sw_lds_t;
typedef __attribute__((ext_vector_type(4))) short sw_s16x4;
typedef __attribute__((ext_vector_type(8))) short sw_s16x8;

__device__ __forceinline__ auto sw_rsrc(const void* p) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7FFFFFFF, 0x00020000);
}
// XOR key on the 16-byte granule of a 128-byte dy row (bits 1, 2 of the row: unchanged by + 16 h and + 32 ks, conv_wgrad_halo.hip)
__device__ __forceinline__ int sw_swz(int row) { return row & 6; }

#define SW_TR(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))

constexpr int SW_SLAB4(int ns) { return ns * 8 * 64; }    // float4 per slab: [s][half][co tile][lane]

// KT = temporal taps (1, 3, 5); NPS = window pieces per thread and plane (a piece = 512 threads x 16 B)
template <int KT, int NPS, int STAGES>
__global__ __launch_bounds__(512, 1) void wgrad_stem_kernel(const StemWGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           float4* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NS = KT * 7, SMAX = (NS + 7) / 8;
  constexpr int PLANE = NPS * 512 * 16;
  constexpr unsigned OOB = 0x80000000u;
  // a stage = [KT][NPS * 512][16 B] input window ((row r, column c) at r * WPL + c), then [256][128 B] dy tile (granules swizzled by
  // sw_swz(row)); STAGES = 2 where two fit the 160 KB
  constexpr int STAGE = KT * PLANE + 256 * 128;
  const unsigned LdsL = (unsigned)(uintptr_t)(sw_lds_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fg = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int r_l = 4 * fg + qq;
  const auto rs_x = sw_rsrc(x);
  const auto rs_dy = sw_rsrc(dy);
  const int wpl16 = __builtin_amdgcn_readfirstlane(g.WPL * 16);

  f32x4_t acc[SMAX][2][4];
#pragma unroll
  for (int si = 0; si < SMAX; ++si)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[si][hf][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // one item's DMA: window (KT planes of NPS pieces; rows outside the plane, the pad pair either side and planes outside the clip are
  // zeros) + dy tile (256 positions x 128 B; positions behind the plane are zeros -- they multiply clamped, finite window rows)
  auto stage_item = [&](int item, int stage) {
    const int po = fdiv(item, g.dTiles), tile = item - po * g.tiles;       // po = n * To + to
    const int n = fdiv(po, g.dTo), to = po - n * g.To;
    const int p0 = tile * 256;
    const int plast = min(p0 + 255, g.HoWo - 1);
    const int oh0 = fdiv(p0, g.dWo), oh1 = fdiv(plast, g.dWo);
    const int NR = 2 * (oh1 - oh0) + 7, ih0 = 2 * oh0 - 3;   // window rows <-> source rows ih0 .. ih0 + NR - 1
    unsigned char* const base = smem + stage * STAGE;
    unsigned win_voff[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int e = ps * 512 + tid;
      const int r = fdiv(e, g.dWPL), c = e - r * g.WPL, ih = ih0 + r;
      const bool ok = r < NR && (unsigned)ih < (unsigned)g.H && c >= 1 && c <= g.WpS;     // window column c <-> source pair c - 1
      win_voff[ps] = ok ? (unsigned)((ih * g.WpS + c - 1) * 16) : OOB;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int ti = to * g.sT + kt - g.pT;
      const bool okp = (unsigned)ti < (unsigned)g.T;
      const unsigned so = __builtin_amdgcn_readfirstlane(okp ? (unsigned)(n * g.T + ti) * g.plane_bytes : 0u);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const unsigned vo = okp ? win_voff[ps] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (sw_lds_t)(base + kt * PLANE + (ps * 512 + wave * 64) * 16), 16, vo, so, 0, 0);
      }
    }
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)po * (unsigned)g.HoWo * 128u);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int row = pc * 64 + (tid >> 3), pg = tid & 7;
      const int p = p0 + row;
      const unsigned vo = p < g.HoWo ? (unsigned)(p * 128 + ((pg ^ sw_swz(row)) * 16)) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (sw_lds_t)(base + KT * PLANE + (pc * 512 + wave * 64) * 16), 16, vo, so, 0, 0);
    }
  };

  const int i0 = (int)((long)blockIdx.x * g.total / gridDim.x), i1 = (int)((long)(blockIdx.x + 1) * g.total / gridDim.x);
  int cur = 0;
  if (STAGES == 2 && i0 < i1) stage_item(i0, 0);
  for (int item = i0; item < i1; ++item) {
    const int tile = item - fdiv(item, g.dTiles) * g.tiles;
    const int p0 = tile * 256;
    const int oh0 = fdiv(p0, g.dWo);
    if (STAGES == 1) {
      __syncthreads();                                     // every wave is done reading the previous item's tiles
      stage_item(item, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      // two stages (the (3,7,7) / (1,7,7) stems: 2 x 80 KB): this item's tiles were issued one item ago; behind the barrier every wave
      // has them and is done reading the other stage, which takes the next item's.  (The transposing reads below are inline asm: the
      // compiler does not fence them against the DMA in flight into the OTHER stage.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (item + 1 < i1) stage_item(item + 1, cur ^ 1);
    }
    const unsigned HsL = LdsL + (unsigned)(cur * STAGE), DsL = HsL + (unsigned)(KT * PLANE);

    // (the k-step loop is NOT unrolled: with it unrolled the (5,7,7) stem's 160 accumulators + 16 row offsets spilled)
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {
      sw_s16x4 va[2][4];
      int posaddr[2];                                      // window offset of this lane's reduction row, per read h
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = r_l + 16 * (2 * ks + h);
        const unsigned ra = DsL + (unsigned)(row * 128 + (pp & 1) * 8);
        const int key = sw_swz(row), g0 = pp >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) SW_TR(va[h][i], ra + (unsigned)(((2 * i + g0) ^ key) * 16));
        const int p = min(p0 + row, g.HoWo - 1);           // (rows behind the plane: a valid, finite window row; their dy is zero)
        const int oh = fdiv(p, g.dWo), ow = p - oh * g.Wo;
        posaddr[h] = (2 * (oh - oh0) * g.WPL + ow) * 16 + pp * 8;
      }
      // B fragments one k step s ahead: the reads of s + 8 are issued before the MFMAs of s and may stay in flight over them (LDS
      // returns in order: lgkmcnt(4) = "everything but the four youngest reads has landed")
      sw_s16x4 vb[2][2][2];                                // [buffer][h][half]
      auto issue_b = [&](int si, int buf) {
        const int s = wave + 8 * si;
        const int kt = s / 7, kh = s - kt * 7;
        const unsigned tb = HsL + (unsigned)(kt * PLANE + kh * wpl16);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) SW_TR(vb[buf][h][hf], tb + (unsigned)(posaddr[h] + hf * 32));
      };
      if (wave < NS) issue_b(0, 0);                        // (KT = 1 has 7 k steps: wave 7 idles)
      bf16x8_t fa[4];
#pragma unroll
      for (int si = 0; si < SMAX; ++si) {
        const int s = wave + 8 * si;                       // wave-uniform
        if (s < NS) {
          const bool more = si + 1 < SMAX && s + 8 < NS;
          if (more) {
            issue_b(si + 1, (si + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          if (si == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const sw_s16x8 w8 = {va[0][i][0], va[0][i][1], va[0][i][2], va[0][i][3], va[1][i][0], va[1][i][1], va[1][i][2], va[1][i][3]};
              fa[i] = __builtin_bit_cast(bf16x8_t, w8);
            }
          }
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const sw_s16x4 b0 = vb[si & 1][0][hf], b1 = vb[si & 1][1][hf];
            const sw_s16x8 w8 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, w8);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[si][hf][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[si][hf][i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (STAGES == 2) cur ^= 1;
  }
  // ---- the block's slab, in register order: float4 index ((s * 2 + half) * 4 + co tile) * 64 + lane; a lane's float4 = co
  // 16 i + 4 (lane >> 4) + r, column 16 half + (lane & 15) (the reduce kernel undoes the order) ----
  float4* slab = slabs + (long)blockIdx.x * SW_SLAB4(NS);
#pragma unroll
  for (int si = 0; si < SMAX; ++si) {
    const int s = wave + 8 * si;
    if (s < NS) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          slab[((s * 2 + hf) * 4 + i) * 64 + lane] = make_float4(acc[si][hf][i][0], acc[si][hf][i][1], acc[si][hf][i][2], acc[si][hf][i][3]);
    }
  }
}

// dw[co][s][32] += sum over the nblk slabs, in a fixed order.  A block = 64 consecutive float4 of a slab x 16 waves: wave w sums slabs
// w, w + 16, ... (up to four loads in flight), the partial sums meet in LDS and wave 0 adds them in wave order.
__global__ __launch_bounds__(1024) void wgrad_stem_reduce_kernel(const float4* __restrict__ slabs4, float* __restrict__ dw, int nblk, int NS) {
  __shared__ float4 part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long S4 = (long)NS * 8 * 64;
  const long q = (long)blockIdx.x * 64 + lane;             // (S4 is a multiple of 64: every lane is live)
  float4 s4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int b = w; b < nblk; b += 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int bb = b + 16 * u;
      if (bb < nblk) { const float4 v = slabs4[(long)bb * S4 + q]; s4[u].x += v.x; s4[u].y += v.y; s4[u].z += v.z; s4[u].w += v.w; }
    }
  }
  part[w][lane] = make_float4((s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y),
                              (s4[0].z + s4[1].z) + (s4[2].z + s4[3].z), (s4[0].w + s4[1].w) + (s4[2].w + s4[3].w));
  __syncthreads();
  if (w != 0) return;
  float r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) { const float4 v = part[k][lane]; r4[0] += v.x; r4[1] += v.y; r4[2] += v.z; r4[3] += v.w; }
  const int u = (int)(q >> 6), i = u & 3, hf = (u >> 2) & 1, s = u >> 3;
  const int co0 = 16 * i + 4 * (lane >> 4), col = 16 * hf + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) dw[((long)(co0 + r) * NS + s) * 32 + col] += r4[r];        // one owner per element: plain adds
}

static long g_wgrad_stem_launches = 0;
extern "C" int64_t mscl_debug_wgrad_stem_launches(void) { return g_wgrad_stem_launches; }      // tests: which kernel family took a launch

static int sw_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  return cus;
}
// window pieces per plane (2 or 3), 0 when the shape is not this kernel's.  d describes the PAIRED convolution: C = 8, K = 64,
// kernel (kT, 7, 4), stride (sT, 2, 1), padding (pT, 3, 1).
static int sw_shape(const mscl_conv_desc* d) {
  if (d->C != 8 || d->K != 64 || d->kH != 7 || d->kW != 4 || d->sH != 2 || d->sW != 1 || d->pH != 3 || d->pW != 1 ||
      (d->kT != 1 && d->kT != 3 && d->kT != 5)) return 0;
  // MSCL_WGRAD_STEM: 0 off, 1 forced (tests: small planes too); default: planes of at least two 256-position tiles
  static MsclTune t("MSCL_WGRAD_STEM");
  const int sw = t.get(-1);
  const long howo = (long)d->Ho * d->Wo;
  if (sw == 0 || (sw != 1 && howo < 512)) return 0;
  if ((long)d->N * d->T * d->H * d->W * 16 >= (1L << 31) || (long)d->N * d->To * howo * 128 >= (1L << 31)) return 0;      // 32-bit offsets
  int span = (256 - 2) / d->Wo + 2; if (span > d->Ho) span = d->Ho;
  const int pairs = (2 * (span - 1) + 7) * (d->W + 2);
  return pairs <= 2 * 512 ? 2 : (pairs <= 3 * 512 ? 3 : 0);
}
static long sw_blocks(const mscl_conv_desc* d) {
  const long items = (long)d->N * d->To * (((long)d->Ho * d->Wo + 255) / 256);
  return items < sw_cus() ? items : sw_cus();
}
// floats of workspace mscl_wgrad_stem wants (0: the layer is not covered)
extern "C" int64_t mscl_wgrad_stem_ws(const mscl_conv_desc* d) {
  if (!d || !sw_shape(d)) return 0;
  return (int64_t)sw_blocks(d) * SW_SLAB4(d->kT * 7) * 4;
}

template <int KT, int NPS>
static void sw_go(const StemWGeom& g, unsigned nblk, const bf16_t* x, const bf16_t* dy, float* ws, hipStream_t st) {
  constexpr size_t stage = (size_t)KT * NPS * 512 * 16 + (size_t)256 * 128;
  constexpr int STAGES = 2 * stage <= 160 * 1024 ? 2 : 1;          // (3,7,7) with two pieces per plane: 2 x 80 KB; (5,7,7): one stage of 112-152 KB
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel<KT, NPS, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_stem_kernel<KT, NPS, STAGES>), dim3(nblk), dim3(512), STAGES * stage, st, g, x, dy, reinterpret_cast<float4*>(ws));
}

// returns 1 if launched, 0 if the shape / workspace is not covered (the caller goes on to the general kernel), <0 / >0 on error
int mscl_wgrad_stem(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats, hipStream_t st) {
  const int nps = sw_shape(d);
  if (nps == 0) return 0;
  const int NS = d->kT * 7;
  const long nblk = sw_blocks(d);
  if (ws == nullptr || nblk * SW_SLAB4(NS) * 4 > ws_floats) return 0;
  StemWGeom g{};
  const long howo = (long)d->Ho * d->Wo;
  g.N = d->N; g.T = d->T; g.To = d->To; g.H = d->H; g.Ho = d->Ho; g.Wo = d->Wo; g.WpS = d->W; g.WPL = d->W + 2;
  g.HoWo = (int)howo; g.tiles = (int)((howo + 255) / 256); g.sT = d->sT; g.pT = d->pT;
  g.total = (int)((long)d->N * d->To * g.tiles);
  g.plane_bytes = (unsigned)(d->H * d->W * 16);
  g.dWo = make_fastdiv(d->Wo); g.dWPL = make_fastdiv(g.WPL); g.dTo = make_fastdiv(d->To); g.dTiles = make_fastdiv(g.tiles);
  if (nps == 2) {
    if (d->kT == 5) sw_go<5, 2>(g, (unsigned)nblk, x, dy, ws, st);
    else if (d->kT == 3) sw_go<3, 2>(g, (unsigned)nblk, x, dy, ws, st); else sw_go<1, 2>(g, (unsigned)nblk, x, dy, ws, st);
  } else {
    if (d->kT == 5) sw_go<5, 3>(g, (unsigned)nblk, x, dy, ws, st);
    else if (d->kT == 3) sw_go<3, 3>(g, (unsigned)nblk, x, dy, ws, st); else sw_go<1, 3>(g, (unsigned)nblk, x, dy, ws, st);
  }
  MSCL_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_stem_reduce_kernel, dim3((unsigned)(SW_SLAB4(NS) / 64)), dim3(1024), 0, st, (const float4*)ws, dw, (int)nblk, NS);
  MSCL_LAUNCH_CHECK();
  ++g_wgrad_stem_launches;
  return 1;
}
