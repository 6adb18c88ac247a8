// Thin-K 1x1x1 / stride-1 convolution into wide outputs (the write-heavy convs of the Bottleneck trunks: 64 -> 256 and 128 -> 512
// channels on the 56^2 / 28^2 maps of ResNet3dSlowOnly-50, forward and -- with the transposed kernel -- the input gradient of the
// 256 -> 64 / 512 -> 128 convs), gfx950.  Reference op: the `conv3` / `conv1` ConvModules of mmaction/models/backbones/resnet3d.py:262-296
// and their autograd input gradient.
//
// Roofline: HBM.  out[M][N] = A[M][K] . W[N][K]^T with K = 64 or 128 and N = 4 K: a row costs 2 K bytes to read and 8 K to write; the
// MFMA work (2 K N flop per row) is 4 % of the time the bytes take.  The implicit-GEMM kernel runs these shapes as one-K-step tiles,
// each a DMA round trip, a barrier, 0.2 us of MFMAs and an epilogue, two resident blocks per CU: 2.4-2.9 TB/s of map traffic where a copy
// kernel of the same bytes runs at 5.3-7 TB/s on the same box (profiles/r05_ab_sweeps.md).  Here:
//  * persistent blocks: block b walks the 128-row tiles b, b + grid, ...; the NEXT tile's rows are on their way into the other half of
//    a double-buffered LDS stage (LDS-DMA, 16-byte pieces, source-side swizzle) while the current one is multiplied and stored;
//  * the weights never touch LDS: a wave owns 64 of the block's 256 output channels and holds its [64 x K] slice as MFMA operands in
//    registers for the whole launch (32 registers at K = 64, 64 at K = 128);
//  * a 16-row tile is complete after 2 K / 32 MFMAs per channel tile -- its 4 accumulators go straight to the epilogue (optional addend,
//    bf16, 16-byte stores: a wave writes one whole 128-byte line per row), so the accumulators never outlive a row tile;
//  * BatchNorm sum / sum of squares stay in registers across ALL tiles of the block and meet the statistics slots once at the end
//    (512 atomics per block instead of 256 per tile).
#include "common.h"

struct K1Geom {
  long M;                          // rows (positions)
  int N;                           // output channels (a multiple of 256)
  int mtiles, ntiles;              // row tiles (128 rows; 64 at K = 256), 256-channel tiles
  int stat_stride;                 // floats between two statistics slots (2 * N)
};

typedef __attribute__((address_space(3))) void* k1_lds_ptr_t;
constexpr unsigned K1_OOB = 0x80000000u;

template <int KS, int BM>          // K = 64 * KS input channels, BM-row tiles
__global__ __launch_bounds__(256, 2) void conv_k1_kernel(const K1Geom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                         bf16_t* __restrict__ out, const bf16_t* __restrict__ addend,
                                                         float* __restrict__ stat_sum, float* __restrict__ stat_sq) {
  constexpr int K = 64 * KS, ROWB = K * 2;                 // bytes per staged row
  constexpr int TILE = BM * ROWB;                          // 16 / 32 KB (K = 256: 64-row tiles, 32 KB)
  constexpr int NP = TILE / 4096;                          // DMA pieces per thread per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][BM][ROWB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // the ntiles blocks that walk the same rows sit on ONE XCD (blocks are dealt round-robin over the 8 XCDs): the second reader of a
  // row tile finds it in that XCD's L2.  (The host makes the grid a multiple of 8 * ntiles.)
  const int xcd = blockIdx.x & 7, yb = blockIdx.x >> 3;
  const int nt = yb % g.ntiles;
  const int mt0 = (yb / g.ntiles) * 8 + xcd, mstep = gridDim.x / g.ntiles;
  const int n0 = nt * 256 + wave * 64;

  // this wave's weights: fb[j][ks] = rows n0 + 16 j + fr, input channels 32 ks + 8 fq .. + 7 (the MFMA's first operand: channels as rows)
  bf16x8_t fb[4][2 * KS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ks = 0; ks < 2 * KS; ++ks)
      fb[j][ks] = *reinterpret_cast<const bf16x8_t*>(wgt + (long)(n0 + j * 16 + fr) * K + ks * 32 + fq * 8);

  // staging: piece p of thread t lands at LDS byte (p * 256 + t) * 16 of the stage, i.e. row s / (8 KS), 16-byte chunk s % (8 KS); it is
  // READ from chunk c ^ (row & 7) of the source row (the swizzle lives on the source side: LDS-DMA writes a wave's 64 pieces back to back)
  const uint64_t a_addr = reinterpret_cast<uint64_t>(src);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a_addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(a_addr >> 32));
  const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFFu, 0x00020000);
  // (row and source offset of a piece are a shift and an xor of the thread index: recomputed per tile rather than held -- the K = 256
  // form keeps 128 registers of weights)
  auto issue = [&](int mt, int buf) {
    const long m0 = (long)mt * BM;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(m0 * ROWB));
    const int left = (int)(g.M - m0 < BM ? g.M - m0 : BM);          // rows of the tile that exist
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int s = p * 256 + tid;
      const int row = s / (8 * KS), c = s % (8 * KS);
      const unsigned vo = row < left ? (unsigned)(row * ROWB + (((c & ~7) | ((c & 7) ^ (row & 7))) * 16)) : K1_OOB;      // rows past the end: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (k1_lds_ptr_t)(smem + buf * TILE + p * 4096 + wave * 1024), 16, vo, so, 0, 0);
    }
  };

  float s1[4][4], s2[4][4];        // [channel tile][channel of the lane's quad]: sums over every row this lane has seen
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

  int buf = 0;
  if (mt0 < g.mtiles) issue(mt0, 0);
  for (int mt = mt0; mt < g.mtiles; mt += mstep) {
    // this tile has landed (and this wave's stores of the tile before are on their way: vmcnt counts both), every wave is done
    // reading the other stage: the next tile may overwrite it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (mt + mstep < g.mtiles) issue(mt + mstep, buf ^ 1);
    const unsigned char* st = smem + buf * TILE;
    const long m0 = (long)mt * BM;
    constexpr int UNR = KS == 4 ? 1 : 2;           // (K = 256: the weights leave no registers for a second row tile in flight)
#pragma unroll UNR
    for (int i = 0; i < BM / 16; ++i) {
      const int row = i * 16 + fr;
      const unsigned char* ab = st + row * ROWB;
      bf16x8_t fa[2 * KS];
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ++ks) {
        const int c = ks * 4 + fq;
        fa[ks] = *reinterpret_cast<const bf16x8_t*>(ab + (((c & ~7) | ((c & 7) ^ (row & 7))) * 16));
      }
      const long grow = m0 + row;
      const bool ok = grow < g.M;
      uint2 add4[4];
      if (addend != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) add4[j] = *reinterpret_cast<const uint2*>(addend + (ok ? grow : 0) * g.N + n0 + j * 16 + fq * 4);
      }
      f32x4_t acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[ks], acc[j], 0, 0, 0);
      }
      // (rows past the end were staged as zeros: they add nothing to the sums)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[j][r]; s1[j][r] += v; s2[j][r] += v * v; }
      // 16-byte stores: two channel tiles paired through v_permlane16_swap (as conv_halo.hip: an even lane row ends up with 8 consecutive
      // channels of tile j, an odd one with 8 of tile j + 1; the partner row holds the same position)
      auto quad = [&](int j) -> uint2 {
        float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        if (addend != nullptr) {
          const uint2 av = add4[j];
          v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
          v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
        }
        uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
        return pv;
      };
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const uint2 q0 = quad(j), q1 = quad(j + 1);
        const auto sx = __builtin_amdgcn_permlane16_swap(q0.x, q1.x, false, false);      // every lane takes part (no divergent branch around)
        const auto sy = __builtin_amdgcn_permlane16_swap(q0.y, q1.y, false, false);
        const int n = (j + (fq & 1)) * 16 + (fq & 2) * 4;
        if (ok) *reinterpret_cast<uint4*>(out + grow * g.N + n0 + n) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
    buf ^= 1;
  }

  if (stat_sum != nullptr) {
    // the lane rows' 16 positions meet by a reduce-scatter (common.h): quad q of a lane row ends with the 8 totals of channel tile q
    float sv[32];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sv[j * 8 + r] = s1[j][r]; sv[j * 8 + 4 + r] = s2[j][r]; }
    row16_reduce_scatter<32>(sv);
    if ((fr & 3) == 0) {
      const int jq = fr >> 2;
      const int so = (int)(blockIdx.x % MSCL_STAT_ACTIVE) * g.stat_stride;
      const int c = n0 + jq * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) { atomicAdd(&stat_sum[so + c + r], sv[r]); atomicAdd(&stat_sq[so + c + r], sv[4 + r]); }
    }
  }
}

static long g_k1_launches = 0;
extern "C" int64_t mscl_debug_k1_launches(void) { return g_k1_launches; }      // tests: which kernel family took a launch

// returns 1 if launched, 0 if the shape is not covered (the caller goes on to the implicit-GEMM kernel), < 0 / > 1 on error.
// src [M][K] bf16 rows, wgt [N][K] (the forward kernel [Cout][1][Cin], or the transposed kernel [Cin][1][Cout] for the input gradient)
int mscl_conv_k1(long M, int K, int N, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const bf16_t* addend, float* ssum, float* ssq,
                 hipStream_t st) {
  if (!src || !wgt || !out || M <= 0) return MSCL_E_ARG;
  static MsclTune t_k1("MSCL_K1");                                    // MSCL_K1=0: off (A/B against the implicit-GEMM kernel)
  if (t_k1.get(1) == 0) return 0;
  if ((K != 64 && K != 128 && K != 256) || N % 256 != 0 || N < 256) return 0;
  if (M * K * 2 >= (1L << 31) || M * N >= (1L << 31) || M < 4096) return 0;      // 32-bit staging offsets; small maps stay where they are
  const int BM = K == 256 ? 64 : 128;              // (K = 256: 128 operand registers of weights per wave, 32-KB stages of 64 rows)
  K1Geom g{};
  g.M = M; g.N = N; g.mtiles = (int)((M + BM - 1) / BM); g.ntiles = N / 256; g.stat_stride = 2 * N;
  // two blocks per CU (<= 64 KB of LDS each), every block at least two tiles so that the pipeline has something to hide
  const long unit = 8L * g.ntiles;
  long blocks = 512 / unit * unit;
  const long most = ((long)g.mtiles / 2 * g.ntiles + unit - 1) / unit * unit;
  if (blocks > most) blocks = most;
  if (blocks < unit) blocks = unit;
  const size_t lds = (size_t)2 * BM * K * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k1_kernel<2, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k1_kernel<4, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (K == 64) hipLaunchKernelGGL((conv_k1_kernel<1, 128>), dim3((unsigned)blocks), dim3(256), lds, st, g, src, wgt, out, addend, ssum, ssq);
  else if (K == 128) hipLaunchKernelGGL((conv_k1_kernel<2, 128>), dim3((unsigned)blocks), dim3(256), lds, st, g, src, wgt, out, addend, ssum, ssq);
  else hipLaunchKernelGGL((conv_k1_kernel<4, 64>), dim3((unsigned)blocks), dim3(256), lds, st, g, src, wgt, out, addend, ssum, ssq);
  MSCL_LAUNCH_CHECK();
  ++g_k1_launches;
  return 1;
}
