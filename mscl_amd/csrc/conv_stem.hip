// Window-resident forward of the RGB stems on bf16 MFMA, gfx950: nn.Conv3d(3, 64, (kT, 7, 7), stride (sT, 2, 2), padding (pT, 3, 3))
// of r3d.py:176-184 (torchvision r3d_18: kT = 3) and resnet3d.py conv1 (ResNet3dSlowOnly: kT = 1; mscl_r50_cosm_lr3e-2.py:18: kT = 5, sT = 2), executed on the W-PAIRED clip
// (elementwise.hip, pair_w_kernel: a position holds two neighbouring pixels as 8 channels, the conv becomes (kT, 7, 4) / stride
// (sT, 2, 1) / pad (pT, 3, 1) over pairs; one (kt, kh) row of the kernel = 4 pairs x 8 channels = ONE 32-deep MFMA k step).
//
// Why its own kernel: the implicit-GEMM kernel (conv_igemm.hip) gathers those 64 bytes per position and k step from global memory
// into LDS -- 21 x 16 KB per 256-position tile, as many LDS-staging clocks as MFMA clocks -- and measured 460 TFLOP/s on the R3D-18
// stem (75 us), 262 on the SlowOnly-50 one (351 us against an 80-us output write).  Here a block stages the input rows its 256
// consecutive output positions can reach ONCE (<= 17 rows of Wo + 3 pairs per source plane, 16 KB; all kT planes resident) and reads
// every k step's position operand straight out of that window: in pair units the 4 pairs of output column ow are window columns
// ow .. ow + 3 of row 2 oh + kh -- 64 contiguous bytes, so a lane's 16-byte fragment is window[(row, ow + (lane >> 4))], conflict-free,
// with no im2col copy at all.  The zero padding (3 rows above / below, one pair column left / right, planes outside the clip) is
// written by the buffer unit's range check while the window is staged.
//  * weights: two k steps (8 KB) per ring stage, gathered by LDS-DMA straight into MFMA fragment order (a lane's chunk sits at
//    [k step][channel tile][lane]: reads are linear, no swizzle); an odd k-step count (21) is padded with a zero step;
//  * half-step operand pipeline, one barrier per stage, two blocks per CU, epilogue (BatchNorm sums, bf16, paired 16-byte stores) as in
//    conv_halo.hip, whose structure this kernel follows.
#include "common.h"

struct StemGeom {
  int N, T, To, H, Ho, Wo, WpS, WPL;   // WpS = source pairs per row (Wo + 1), WPL = window pairs per row (Wo + 3)
  int HoWo, tiles, sT, pT;
  int stat_stride;
  unsigned plane_bytes;               // one source plane: H * WpS * 16
  FastDiv dWo, dWPL, dTo;
};

typedef __attribute__((address_space(3))) void* stem_lds_ptr_t;

__device__ __forceinline__ auto stem_rsrc(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// KT = temporal taps (1, 3, 5); NPS = window pieces per thread and plane (a piece = 512 threads x 16 B: NPS * 512 pairs hold the window
// of one plane); RING = weight-ring stages.
template <int KT, int NPS, int RING>
__global__ __launch_bounds__(512, 4) void conv_stem_kernel(const StemGeom g, const bf16_t* __restrict__ src,
                                                          const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
                                                          float* __restrict__ stat_sum, float* __restrict__ stat_sq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256, NW = 8, HC = 64;
  constexpr int NS = KT * 7, NU = (NS + 1) / 2;            // k steps, ring stages walked (two k steps each)
  constexpr int PLANE = NPS * 512 * 16;
  constexpr unsigned HOOB = 0x80000000u;
  static_assert(RING >= 2 && RING <= 4 && NU >= RING, "configuration");
  unsigned char* const Hs = smem;                          // [KT][NPS * 512][16 B] input window, (row r, column c) at r * WPL + c
  unsigned char* const Ws = smem + KT * PLANE;             // [RING][2 k steps][4 channel tiles][64 lanes][16 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = bid % g.tiles, po = bid / g.tiles;      // po = n * To + to
  const int n = fdiv(po, g.dTo), to = po - n * g.To;
  const int p0 = tile * BM;
  const int plast = min(p0 + BM - 1, g.HoWo - 1);
  const int oh0 = fdiv(p0, g.dWo), oh1 = fdiv(plast, g.dWo);
  const int NR = 2 * (oh1 - oh0) + 7, ih0 = 2 * oh0 - 3;   // window rows <-> source rows ih0 .. ih0 + NR - 1
  const auto rs_src = stem_rsrc(src, 0x7FFFFFFFu);
  const auto rs_wgt = stem_rsrc(wgt, 0x7FFFFFFFu);

  // ---- window DMA: per-piece offsets inside a source plane (the plane goes in the SGPR offset) ----
  unsigned win_voff[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int e = ps * 512 + tid;
    const int r = fdiv(e, g.dWPL), c = e - r * g.WPL, ih = ih0 + r;
    const bool ok = r < NR && (unsigned)ih < (unsigned)g.H && c >= 1 && c <= g.WpS;     // window column c <-> source pair c - 1
    win_voff[ps] = ok ? (unsigned)((ih * g.WpS + c - 1) * 16) : HOOB;
  }
  auto issue_plane = [&](int kt) {
    const int ti = to * g.sT + kt - g.pT;
    const bool okp = (unsigned)ti < (unsigned)g.T;
    const unsigned so = __builtin_amdgcn_readfirstlane(okp ? (unsigned)(n * g.T + ti) * g.plane_bytes : 0u);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const unsigned vo = okp ? win_voff[ps] : HOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (stem_lds_ptr_t)(Hs + kt * PLANE + (ps * 512 + wave * 64) * 16), 16, vo, so, 0, 0);
    }
  };
  // ---- weights [64][NS][32] -> fragment order: thread tid = (k step of the stage, channel tile, lane) fetches its own chunk ----
  const int w_ks = tid >> 8, w_j = (tid >> 6) & 3;
  const unsigned w_voff = (unsigned)((((w_j * 16 + (lane & 15)) * NS + w_ks) * 32 + (lane >> 4) * 8) * 2);
  auto issue_weights = [&](int u) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(u * 2 * 32 * 2));
    const unsigned vo = (2 * u + 1 >= NS && w_ks == 1) ? HOOB : w_voff;        // the zero step behind an odd count
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (stem_lds_ptr_t)(Ws + (u % RING) * 8192 + wave * 1024), 16, vo, so, 0, 0);
  };

  // ---- prologue: plane 0 and the first RING weight stages, then the other planes: the loop starts when plane 0 and stage 0 have
  // landed; the later planes are first read in stage 3 (k step 7), whose wait (at the end of stage 2) is for a weight stage issued
  // AFTER them and so covers them, while the waits of stages 0 and 1 leave their pieces in flight ----
  // (KT = 5, the (5,7,7) / temporal-stride-2 stem of mscl_r50: four later planes are more pieces than the counted wait below can
  // leave in flight, and five planes make ONE block per CU anyway -- every plane goes out first and has landed when the loop starts)
  constexpr bool PLANES_FIRST = KT > 3;
  issue_plane(0);
  if constexpr (PLANES_FIRST) {
#pragma unroll
    for (int kt = 1; kt < KT; ++kt) issue_plane(kt);
  }
#pragma unroll
  for (int u = 0; u < RING; ++u) issue_weights(u);
  if constexpr (!PLANES_FIRST) {
#pragma unroll
    for (int kt = 1; kt < KT; ++kt) issue_plane(kt);
  }

  const int fr = lane & 15, fq = lane >> 4;
  const int wpl16 = __builtin_amdgcn_readfirstlane(g.WPL * 16);
  int a_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = min(p0 + wave * 32 + i * 16 + fr, g.HoWo - 1);     // positions behind the plane read a valid row (not stored)
    const int oh = fdiv(p, g.dWo), ow = p - oh * g.Wo;
    a_base[i] = (2 * (oh - oh0) * g.WPL + ow + fq) * 16;
  }
  bf16x8_t fa[2][2], fb[2][4];                             // [set][fragment]: one set per half stage
  auto read_half = [&](int u, int ks, int set) {
    const int s = (2 * u + ks < NS) ? 2 * u + ks : NS - 1;  // (the zero step multiplies any valid rows)
    const int kt = s / 7, kh = s % 7;
    const unsigned char* wb = Ws + (u % RING) * 8192 + ks * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8_t*>(wb + j * 1024);
    const unsigned char* hb = Hs + kt * PLANE + kh * wpl16;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const bf16x8_t*>(hb + a_base[i]);
  };
  f32x4_t acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][i], acc[j][i], 0, 0, 0);
  };
#define ST_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory")
  // plane 0 + stage 0 landed; the other planes and RING - 1 stages stay in flight
  constexpr int LATE = PLANES_FIRST ? 0 : (KT - 1) * NPS; // pieces of the later planes still in flight behind the first weight stages
  static_assert(RING - 1 + LATE <= 8 && (KT == 1 || RING <= 3), "prologue wait");
#define ST_WAIT_N(n) do { if ((n) >= 8) ST_WAIT(8); else if ((n) == 7) ST_WAIT(7); else if ((n) == 6) ST_WAIT(6); else if ((n) == 5) ST_WAIT(5); \
    else if ((n) == 4) ST_WAIT(4); else if ((n) == 3) ST_WAIT(3); else if ((n) == 2) ST_WAIT(2); else if ((n) == 1) ST_WAIT(1); else ST_WAIT(0); } while (0)
  ST_WAIT_N(RING - 1 + LATE);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_half(0, 0, 0);

#pragma unroll
  for (int u = 0; u < NU; ++u) {
    read_half(u, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < NU) {
      // stage u + 1 landed (the RING - 2 younger ones, fewer at the end, may stay in flight); this wave's reads of stage u retired;
      // the barrier publishes the one and frees the other
      const int younger = ((RING - 2) < (NU - 2 - u) ? (RING - 2) : (NU - 2 - u)) + (u <= RING - 2 && u <= 1 ? LATE : 0);
      ST_WAIT_N(younger);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (u + RING < NU) issue_weights(u + RING);
      read_half(u + 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef ST_WAIT_N
#undef ST_WAIT
  __syncthreads();                  // the epilogue reuses the window memory

  long orow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + wave * 32 + i * 16 + fr;
    const bool ok = p < g.HoWo;
    orow[i] = ok ? ((long)po * g.HoWo + p) * HC : -1;
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- epilogue: BatchNorm statistics of the fp32 result (as conv_halo64b_kernel) ----
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [NW][2][64]
    const int wv = tid >> 6;
    // value 8 j + 4 sq + r = (channel tile j, sum / sum of squares, channel r of this lane row's quad): after the reduce-scatter
    // quad q of a lane row holds the 8 totals of channel tile q
    float sv[32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = acc[j][0][r], v1 = acc[j][1][r];
        sv[j * 8 + r] = v0 + v1; sv[j * 8 + 4 + r] = v0 * v0 + v1 * v1;
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
  // 16-byte stores: two channel tiles paired through v_permlane16_swap (igemm.h)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const auto sx = __builtin_amdgcn_permlane16_swap(pack2bf(acc[j][i][0], acc[j][i][1]), pack2bf(acc[j + 1][i][0], acc[j + 1][i][1]), false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(pack2bf(acc[j][i][2], acc[j][i][3]), pack2bf(acc[j + 1][i][2], acc[j + 1][i][3]), false, false);
      const int nn = (j + (fq & 1)) * 16 + (fq & 2) * 4;
      if (orow[i] >= 0) *reinterpret_cast<uint4*>(out + orow[i] + nn) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
  }
}

static long g_stem_launches = 0;
extern "C" int64_t mscl_debug_stem_launches(void) { return g_stem_launches; }      // tests: which kernel family took a launch

template <int KT, int NPS>
static void stem_go(const StemGeom& g, unsigned nblk, const bf16_t* x, const bf16_t* w, bf16_t* y, float* ssum, float* ssq, hipStream_t st) {
  constexpr int RING = KT == 1 ? 2 : 3;           // (KT = 5: 120 KB of window at NPS = 3 + 24 KB of ring: one block per CU)
  constexpr size_t lds = (size_t)KT * NPS * 512 * 16 + (size_t)RING * 8192;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel<KT, NPS, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_stem_kernel<KT, NPS, RING>), dim3(nblk), dim3(512), lds, st, g, x, w, y, ssum, ssq);
}

// returns 1 if launched, 0 if the shape is not covered (the caller goes on to the implicit-GEMM kernel), <0 / >0 on error.
// d describes the PAIRED convolution: C = 8, K = 64, kernel (kT, 7, 4), stride (sT, 2, 1), padding (pT, 3, 1).
int mscl_conv_stem(const mscl_conv_desc* d, const bf16_t* x, const bf16_t* w, bf16_t* y, float* ssum, float* ssq, hipStream_t st) {
  if (d->C != 8 || d->K != 64 || d->kH != 7 || d->kW != 4 || d->sH != 2 || d->sW != 1 || d->pH != 3 || d->pW != 1 ||
      (d->kT != 1 && d->kT != 3 && d->kT != 5)) return 0;
  // MSCL_STEM: 0 off, 1 forced (tests: small planes too); default: planes of at least two 256-position tiles
  static MsclTune t("MSCL_STEM");
  const int sw = t.get(-1);
  const long howo = (long)d->Ho * d->Wo;
  if (sw == 0 || (sw != 1 && howo < 512)) return 0;
  if ((long)d->N * d->T * d->H * d->W * 16 >= (1L << 31)) return 0;            // source offsets are 32-bit
  StemGeom g{};
  g.N = d->N; g.T = d->T; g.To = d->To; g.H = d->H; g.Ho = d->Ho; g.Wo = d->Wo; g.WpS = d->W; g.WPL = d->W + 2;
  g.HoWo = (int)howo; g.tiles = (int)((howo + 255) / 256); g.sT = d->sT; g.pT = d->pT;
  g.stat_stride = 2 * 64;
  g.plane_bytes = (unsigned)(d->H * d->W * 16);
  g.dWo = make_fastdiv(d->Wo); g.dWPL = make_fastdiv(g.WPL); g.dTo = make_fastdiv(d->To);
  // output rows a 256-position tile can touch (worst start column), source rows behind them, window pairs per plane
  int span = (256 - 2) / d->Wo + 2; if (span > d->Ho) span = d->Ho;
  const int pairs = (2 * (span - 1) + 7) * g.WPL;
  const unsigned nblk = (unsigned)((long)d->N * d->To * g.tiles);
  if (pairs <= 2 * 512) {
    if (d->kT == 5) stem_go<5, 2>(g, nblk, x, w, y, ssum, ssq, st);
    else if (d->kT == 3) stem_go<3, 2>(g, nblk, x, w, y, ssum, ssq, st); else stem_go<1, 2>(g, nblk, x, w, y, ssum, ssq, st);
  } else if (pairs <= 3 * 512) {
    if (d->kT == 5) stem_go<5, 3>(g, nblk, x, w, y, ssum, ssq, st);
    else if (d->kT == 3) stem_go<3, 3>(g, nblk, x, w, y, ssum, ssq, st); else stem_go<1, 3>(g, nblk, x, w, y, ssum, ssq, st);
  } else return 0;
  MSCL_LAUNCH_CHECK();
  ++g_stem_launches;
  return 1;
}
