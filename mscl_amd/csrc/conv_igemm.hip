// Conv3d forward and input-gradient as ONE implicit-GEMM kernel family on bf16 MFMA (gfx950).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//   rows    m = positions of the "row tensor"  (forward: y positions; dgrad: dx positions)
//   columns n = channels of the row tensor     (forward: Cout;        dgrad: Cin)
//   depth   k = (tap, source channel)          gathered on the fly from the NDHWC "source tensor"
//                                               (forward: x; dgrad: dy) -- no im2col buffer.
// B is the kernel laid out [n][tap][source channel] (K-contiguous, like A), so both operands are read
// from LDS as 16-byte k-contiguous fragments of v_mfma_f32_16x16x32_bf16.
//
// MI355X mapping
//  * 256 threads = 4 waves, block tile BM x BN, depth step BK (64 = one 128-byte channel row per tap, or 32);
//    tiles are sized so that >= 3 blocks fit a CU's 160 KB LDS (a 256x64x64 tile = 80 KB + tables fits once:
//    measured 0.84 waves/SIMD and 258 TFLOP/s, vs 512 TFLOP/s for 128x64x64).
//  * Staging is LDS-DMA (buffer_load ... lds), double-buffered, one barrier per K step.  A wave-instruction
//    writes 1 KiB of LDS linearly, so the XOR swizzle that makes every ds_read_b128 lane group hit 16
//    distinct 16-byte slots is applied to the per-lane SOURCE offset.  Padding taps, the K tail and rows
//    beyond M use an out-of-range offset: the buffer unit writes zeros, no VALU select, no ds_write.
//  * The product is computed transposed (weights as the MFMA A operand): a lane owns 4 consecutive output
//    channels of one position -> 8-byte stores; BatchNorm sum / sum-of-squares are reduced in the epilogue
//    (shuffles -> LDS -> one atomic per channel per block).
//  * Strided input-gradient: output positions are split into stride-parity classes; each class only visits
//    the taps that can reach it (27/8 of the taps on average for 3x3x3 stride 2) -- one launch, class in the grid.
//  * Small-M layers (layer3/4, pyramid levels): split-K over the grid with fp32 atomic partials + a finalize
//    pass (bias / addend / ReLU / BN statistics / bf16), so >= 2 blocks per CU exist even at M = 784.
//  * blockIdx is remapped so that an XCD's L2 sees neighbouring position tiles (halo reuse).
#include "igemm.h"
#include <cstdio>
#include <cstdlib>


template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const IGemmGeom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
    const float* __restrict__ bias, const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
    float* __restrict__ stat_sq, const int relu, float* __restrict__ partial) {
  constexpr int GPR = BK / 8;                 // 16-byte granules per tile row
  constexpr int RPP = 256 / GPR;              // rows covered per staging pass
  constexpr int AP = BM / RPP;                // A staging passes
  constexpr int BP = (BN + RPP - 1) / RPP;    // B staging passes
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int IM = WM / 16, JN = WN / 16;
  constexpr int KSUB = BK / 32;
  static_assert(WAVES_M * WAVES_N == 4 && IM >= 1 && JN >= 1 && AP >= 1, "tile config");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);                       // [STAGES][BM*BK]
  bf16_t* Bs = As + STAGES * BM * BK;                                 // [STAGES][BN*BK]
  int* tap_delta = reinterpret_cast<int*>(Bs + STAGES * BN * BK);     // [ntaps] per tap-list slot
  int* tap_bits = tap_delta + g.ntaps;                                // [ntaps]
  int* tap_id = tap_bits + g.ntaps;                                   // [ntaps] actual tap (weight addressing)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases / M0 stay scalar
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % g.ksplit; bid /= g.ksplit;
  // parity class (0 when nclass == 0) is the FASTEST index after the split: xcd_remap hands each XCD a contiguous run of
  // logical ids, and with the class slowest XCD 7 got nothing but the 8-tap class (8/27 of the work on 1/8 of the chip)
  // while XCD 0 got the 1-tap one; interleaved, the 8 classes of a tile also share their dy rows in one L2
  int ci = 0;
  if (g.nclass > 0) { ci = bid % g.nclass; bid /= g.nclass; }
  const int nt = bid % g.ntiles; bid /= g.ntiles;
  const int mt = bid;
  const int m0 = mt * BM, n0 = nt * BN;

  // class geometry (dense launches: one class covering every row, identity tap list)
  int roT = 0, roH = 0, roW = 0, rsT = 1, rsH = 1, rsW = 1, TrS = g.Tr, HrS = g.Hr, WrS = g.Wr, Mc = g.M, ntl = g.ntaps;
  FastDiv dW = g.dW, dH = g.dH, dT = g.dT;
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);
  ClassInfo csel = g.cls[0];                  // scalar select instead of dynamic indexing of the kernarg table
#pragma unroll
  for (int c = 1; c < 8; ++c) if (ci == c) csel = g.cls[c];
  if (g.nclass > 0) {
    roT = csel.ro[0]; roH = csel.ro[1]; roW = csel.ro[2]; rsT = g.sT; rsH = g.sH; rsW = g.sW;
    TrS = csel.TrS; HrS = csel.HrS; WrS = csel.WrS; Mc = csel.M; ntl = csel.ntl;
    dW = csel.dW; dH = csel.dH; dT = csel.dT;
  }
  if (m0 >= Mc) return;                       // class smaller than the grid's (max) tile count

  unsigned csel_lo = 0, csel_hi = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) { csel_lo |= (unsigned)csel.taps[t] << (8 * t); csel_hi |= (unsigned)csel.taps[4 + t] << (8 * t); }
  // ---- tap tables ----
  for (int t = tid; t < ntl; t += 256) {
    const int id = (g.nclass > 0) ? (int)((t < 4 ? (csel_lo >> (8 * t)) : (csel_hi >> (8 * (t - 4)))) & 255u) : t;
    const int kw = id % g.kW, kh = (id / g.kW) % g.kH, kt = id / (g.kW * g.kH);
    const int lin = (kt * g.Hs + kh) * g.Ws + kw;
    tap_delta[t] = (mode == 0) ? lin : ((mode == 1) ? -lin : (kt | (kh << 8) | (kw << 16)));
    tap_bits[t] = (1 << kt) | (1 << (8 + kh)) | (1 << (16 + kw));
    tap_id[t] = id;
  }

  // ---- per-row gather state (one row per staging pass) ----
  const int rg = tid % GPR;                   // this thread's physical granule column within a tile row
  const int rr = tid / GPR;                   // row within a pass
  int row_base[AP], row_mask[AP], row_aux[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + p * RPP + rr;
    int mask = 0, base = 0, aux = 0;
    if (m < Mc) {
      const int q1 = fdiv(m, dW), ws_ = m - q1 * WrS;
      const int q2 = fdiv(q1, dH), hs_ = q1 - q2 * HrS;
      const int n = fdiv(q2, dT), ts_ = q2 - n * TrS;
      const int tr = ts_ * rsT + roT, hr = hs_ * rsH + roH, wr = ws_ * rsW + roW;
      int t0, h0, w0;
      if (mode == 0) { t0 = tr * g.sT - g.pT; h0 = hr * g.sH - g.pH; w0 = wr * g.sW - g.pW; }
      else { t0 = tr + g.pT; h0 = hr + g.pH; w0 = wr + g.pW; }
      for (int k = 0; k < g.kT; ++k) {
        const int d = (mode == 0) ? t0 + k : t0 - k;
        const bool ok = (mode == 2) ? (d >= 0 && (d & (g.sT - 1)) == 0 && (d >> g.lsT) < g.Ts) : ((unsigned)d < (unsigned)g.Ts);
        mask |= ok ? (1 << k) : 0;
      }
      for (int k = 0; k < g.kH; ++k) {
        const int d = (mode == 0) ? h0 + k : h0 - k;
        const bool ok = (mode == 2) ? (d >= 0 && (d & (g.sH - 1)) == 0 && (d >> g.lsH) < g.Hs) : ((unsigned)d < (unsigned)g.Hs);
        mask |= ok ? (1 << (8 + k)) : 0;
      }
      for (int k = 0; k < g.kW; ++k) {
        const int d = (mode == 0) ? w0 + k : w0 - k;
        const bool ok = (mode == 2) ? (d >= 0 && (d & (g.sW - 1)) == 0 && (d >> g.lsW) < g.Ws) : ((unsigned)d < (unsigned)g.Ws);
        mask |= ok ? (1 << (16 + k)) : 0;
      }
      if (mode == 2) { base = n * g.Ts; aux = t0 | (h0 << 10) | (w0 << 20); }
      else base = ((n * g.Ts + t0) * g.Hs + h0) * g.Ws + w0;
    }
    row_base[p] = base; row_mask[p] = mask; row_aux[p] = aux;
  }
  __syncthreads();   // tap tables visible

  const int cmask = (1 << g.cgs) - 1;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4_t acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int KGc = ntl << g.cgs;                       // K granules of this class
  const int nk_all = (KGc + GPR - 1) / GPR;
  const int k_beg = (int)((long)nk_all * split / g.ksplit), k_end = (int)((long)nk_all * (split + 1) / g.ksplit);

  const int rgl = swz<BK>(rr, rg);            // logical granule this lane fetches (pass-invariant)
  const unsigned src_bytes = (unsigned)((long)g.N * g.Ts * g.Hs * g.Ws * g.Cs * 2);
  const unsigned wgt_bytes = (unsigned)((long)g.Cr * g.KG * 16);
  // descriptors built from readfirstlane'd words: otherwise hipcc wraps every buffer op in a waterfall loop
  const auto rs_src = make_uniform_rsrc(src, src_bytes);
  const auto rs_wgt = make_uniform_rsrc(wgt, wgt_bytes);
  const int cs2 = g.Cs * 2;
  int row_byte[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) row_byte[p] = (mode == 2) ? row_base[p] : row_base[p] * cs2;
  int wrow_byte[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int r = p * RPP + rr;
    wrow_byte[p] = (r < BN && n0 + r < g.Cr) ? (n0 + r) * g.KG * 16 : -1;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto issue_tiles = [&](int kt, int buf) {
    const int kg = kt * GPR + rgl;
    const bool kin = kg < KGc;
    const int slot = kin ? (kg >> g.cgs) : 0;
    const int cgr = kg & cmask;
    const int td = tap_delta[slot], tb = tap_bits[slot];
    const int kgw = (tap_id[slot] << g.cgs) + cgr;    // granule index inside a weight row
    unsigned char* a = reinterpret_cast<unsigned char*>(As + buf * BM * BK) + wave * 1024;
    unsigned char* b = reinterpret_cast<unsigned char*>(Bs + buf * BN * BK) + wave * 1024;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool ok = kin && ((row_mask[p] & tb) == tb);
      unsigned off;
      if (mode == 2) {
        const int aa = row_aux[p];
        const int dt = ((aa & 1023) - (td & 255)) >> g.lsT;
        const int dh = (((aa >> 10) & 1023) - ((td >> 8) & 255)) >> g.lsH;
        const int dw = (((aa >> 20) & 1023) - ((td >> 16) & 255)) >> g.lsW;
        off = (unsigned)((((row_byte[p] + dt) * g.Hs + dh) * g.Ws + dw) * cs2 + (cgr << 4));
      } else {
        off = (unsigned)(row_byte[p] + td * cs2 + (cgr << 4));
      }
      off = ok ? off : src_bytes;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(a + p * (RPP * BK * 2)), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      if (p * RPP + (wave * 64) / GPR < BN) {       // wave-uniform: this wave's 1 KiB chunk lies inside the B tile
        const unsigned off = (kin && wrow_byte[p] >= 0) ? (unsigned)(wrow_byte[p] + kgw * 16) : wgt_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(b + p * (RPP * BK * 2)), 16, off, 0, 0, 0);
      }
    }
  };

  auto compute_tile = [&](int buf) {
    const unsigned char* a = reinterpret_cast<const unsigned char*>(As + buf * BM * BK);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(Bs + buf * BN * BK);
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      bf16x8_t fa[IM], fb[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        const int r = wm0 + i * 16 + fr;
        fa[i] = *reinterpret_cast<const bf16x8_t*>(a + r * (BK * 2) + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int r = wn0 + j * 16 + fr;
        fb[j] = *reinterpret_cast<const bf16x8_t*>(b + r * (BK * 2) + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
  };

  static_assert(STAGES == 2, "two LDS stages (a deeper ring never paid: see conv_igemm_fast_kernel)");
  if (k_beg < k_end) {
    issue_tiles(k_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = k_beg; kt < k_end; ++kt) {
    const int cur = (kt - k_beg) & 1;
    if (kt + 1 < k_end) issue_tiles(kt + 1, cur ^ 1);
    compute_tile(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  igemm_epilogue<BM, BN, IM, JN>(g, acc, smem, tid, fr, fq, m0, n0, wm0, wn0, split, Mc, dW, dH, dT, TrS, HrS, WrS, rsT, rsH, rsW,
                                 roT, roH, roW, out, bias, addend, stat_sum, stat_sq, relu, partial);
}

// ---------------------------------------------------------------------------------------------------------------
// Uniform-tap variant for the layers that carry the FLOPs (source channels a multiple of 64, forward or stride-1
// input gradient).  One K step = 64 channels of ONE tap, so the tap is wave-uniform: its offset travels in the
// buffer instruction's SGPR offset and the per-lane VGPR offset of a row never changes.  A K step then costs 3 VALU
// per A piece (tap-validity test + select of the zero-fill sentinel) and none per B piece, instead of the ~30 the
// general kernel spends on table lookups and address arithmetic (measured there: 7.8 VALU per MFMA, 22 % MFMA busy).
// Two LDS stages: issue tile k+1, compute tile k, drain, barrier.  (A 3- / 4-stage ring with counted vmcnt waits and a raw barrier
// was built in round 2 and never beat two blocks per CU at depth 2 -- 57 vs 82 us on 128 -> 128 x 50176 positions; removed in round
// 3.  What does beat this loop on the big maps is the ping-pong structure of conv_pp.hip.)
// WAVES_M x WAVES_N = 4 (256 threads) or 8 (512 threads, the WIDE tiles 256 x 128 / 128 x 256: 48 KB staged per 4.2 MFLOP
// instead of 32 KB per 2.1).  Measured round 2 (tools/bench_conv.py --sweep, one process): ring depth 3 / 4 never beats two
// blocks per CU at depth 2 (128 -> 128 on 50176 positions: 57 us at depth 2, 82 at depth 3 with one block per CU); the
// 256 x 128 tile ties the 128 x 128 one there (53.8 vs 54.0 us) and loses on small maps; 128 x 256 wins a little where 256 output
// channels meet a few thousand positions (40.4 -> 37.4 us).  PMC on the 128 x 128 tile: MFMA pipe busy 36 %, waves 34 % parked
// on vmcnt / barrier, 29 % issue-stalled -- the plateau of the two-barrier-per-K-step structure (cdna_hip_programming.md 5).
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_fast_kernel(
    const IGemmGeom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
    const float* __restrict__ bias, const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
    float* __restrict__ stat_sq, const int relu, float* __restrict__ partial) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BK = 64, GPR = 8, RPP = NT / GPR;
  constexpr int AP = BM / RPP, BP = BN / RPP;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int IM = WM / 16, JN = WN / 16;
  constexpr int A_STAGE = BM * BK * 2, B_STAGE = BN * BK * 2;
  constexpr unsigned OOB = 0x80000000u;       // >= num_records with or without the SGPR offset added
  static_assert((WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8) && IM >= 1 && JN >= 1 && AP >= 1 && BP >= 1 && BN % RPP == 0 &&
                BM % RPP == 0, "tile config");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;                       // [STAGES][BM*BK] bf16
  unsigned char* const Bs = smem + STAGES * A_STAGE;    // [STAGES][BN*BK] bf16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % g.ksplit; bid /= g.ksplit;
  const int nt = bid % g.ntiles; bid /= g.ntiles;
  const int mt = bid;
  const int m0 = mt * BM, n0 = nt * BN;
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);      // 0 forward, 1 stride-1 input gradient
  const int Mc = g.M;
  const int cs2 = g.Cs * 2;
  // tap offsets are kept non-negative for the SGPR operand: the descriptor base is moved back by `bias_bytes`
  const int maxlin = ((g.kT - 1) * g.Hs + (g.kH - 1)) * g.Ws + (g.kW - 1);
  const int padlin = (g.pT * g.Hs + g.pH) * g.Ws + g.pW;
  const int bias_bytes = (mode == 0 ? padlin : maxlin) * cs2;

  const int rg = tid & 7, rr = tid >> 3;
  const int rgl = swz<BK>(rr, rg);
  int row_voff[AP], row_mask[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + p * RPP + rr;
    int mask = 0, base = 0;
    if (m < Mc) {
      const int q1 = fdiv(m, g.dW), wr = m - q1 * g.Wr;
      const int q2 = fdiv(q1, g.dH), hr = q1 - q2 * g.Hr;
      const int n = fdiv(q2, g.dT), tr = q2 - n * g.Tr;
      int t0, h0, w0;
      if (mode == 0) { t0 = tr * g.sT - g.pT; h0 = hr * g.sH - g.pH; w0 = wr * g.sW - g.pW; }
      else { t0 = tr + g.pT; h0 = hr + g.pH; w0 = wr + g.pW; }
      for (int k = 0; k < g.kT; ++k) { const int d = (mode == 0) ? t0 + k : t0 - k; mask |= ((unsigned)d < (unsigned)g.Ts) ? (1 << k) : 0; }
      for (int k = 0; k < g.kH; ++k) { const int d = (mode == 0) ? h0 + k : h0 - k; mask |= ((unsigned)d < (unsigned)g.Hs) ? (1 << (8 + k)) : 0; }
      for (int k = 0; k < g.kW; ++k) { const int d = (mode == 0) ? w0 + k : w0 - k; mask |= ((unsigned)d < (unsigned)g.Ws) ? (1 << (16 + k)) : 0; }
      base = ((n * g.Ts + t0) * g.Hs + h0) * g.Ws + w0;
    }
    row_voff[p] = base * cs2 + (mode == 0 ? bias_bytes : 0) + rgl * 16;      // >= 0 for every row with a valid tap
    row_mask[p] = mask;
  }
  unsigned wrow_voff[BP];
#pragma unroll
  for (int p = 0; p < BP; ++p) {
    const int r = p * RPP + rr;
    wrow_voff[p] = (n0 + r < g.Cr) ? (unsigned)((n0 + r) * g.KG * 16 + rgl * 16) : OOB;
  }
  const unsigned char* src_b = reinterpret_cast<const unsigned char*>(src) - bias_bytes;
  const auto rs_src = make_uniform_rsrc(src_b, 0x7FFFFFFFu);
  const auto rs_wgt = make_uniform_rsrc(wgt, 0x7FFFFFFFu);

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4_t acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int sub = g.cgs - 3, submask = (1 << sub) - 1;          // K steps per tap = Cs / 64
  const int nk_all = g.ntaps << sub;
  const int k_beg = (int)((long)nk_all * split / g.ksplit), k_end = (int)((long)nk_all * (split + 1) / g.ksplit);

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto issue = [&](int kt, int buf) {
    const int slot = kt >> sub, cpart = kt & submask;
    const int q = fdiv(slot, g.dKW), kw = slot - q * g.kW;
    const int kd = fdiv(q, g.dKH), kh = q - kd * g.kH;
    const int lin = (kd * g.Hs + kh) * g.Ws + kw;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((mode == 0 ? lin : maxlin - lin) * cs2 + cpart * 128));
    const int tb = __builtin_amdgcn_readfirstlane((1 << kd) | (1 << (8 + kh)) | (1 << (16 + kw)));
    const unsigned woff = __builtin_amdgcn_readfirstlane((unsigned)kt * 128u);
    unsigned char* a = As + buf * A_STAGE + wave * 1024;
    unsigned char* b = Bs + buf * B_STAGE + wave * 1024;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const unsigned off = ((row_mask[p] & tb) == tb) ? (unsigned)row_voff[p] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(a + p * (RPP * BK * 2)), 16, off, soff, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const unsigned wv = wrow_voff[p];         // (a captured array element passed straight to the builtin loses the host stub)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(b + p * (RPP * BK * 2)), 16, wv, woff, 0, 0);
    }
  };
  // per-lane fragment byte offsets inside a stage (row * 128 + swizzled granule * 16), loop-invariant
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ra = wm0 + fr, rb = wn0 + fr;     // rows of fragment i / j differ by multiples of 16: same swizzle key
    a_off[ks] = ra * 128 + swz<BK>(ra, ks * 4 + fq) * 16;
    b_off[ks] = rb * 128 + swz<BK>(rb, ks * 4 + fq) * 16;
  }
  auto compute = [&](int buf) {
    const unsigned char* a = As + buf * A_STAGE;
    const unsigned char* b = Bs + buf * B_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[IM], fb[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(a + a_off[ks] + i * (16 * 128));
#pragma unroll
      for (int j = 0; j < JN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(b + b_off[ks] + j * (16 * 128));
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
  };

  // STAGES == 1: layers whose whole reduction is ONE K step (1x1x1 convs from 64 channels: the write-heavy convs of the Bottleneck
  // trunks).  Nothing is ever staged behind the first tile, so half the LDS serves and twice the blocks share a CU -- for a launch
  // that is a staging round trip, 0.15 us of MFMA work and an epilogue per block, resident blocks are what hides the latency.
  static_assert(STAGES == 2 || STAGES == 1, "LDS stages");
  if (k_beg < k_end) {
    issue(k_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = k_beg; kt < k_end; ++kt) {
    const int cur = STAGES == 2 ? ((kt - k_beg) & 1) : 0;
    if constexpr (STAGES == 2) { if (kt + 1 < k_end) issue(kt + 1, cur ^ 1); }
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  igemm_epilogue<BM, BN, IM, JN>(g, acc, smem, tid, fr, fq, m0, n0, wm0, wn0, split, Mc, g.dW, g.dH, g.dT, g.Tr, g.Hr, g.Wr, 1, 1, 1,
                                 0, 0, 0, out, bias, addend, stat_sum, stat_sq, relu, partial);
}

// split-K epilogue: out = bf16( relu?( sum of slabs + bias + addend ) ), BN statistics of the sum
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ partial, bf16_t* __restrict__ out,
                                                              const float* __restrict__ bias, const bf16_t* __restrict__ addend,
                                                              int relu, float* __restrict__ ssum, float* __restrict__ ssq,
                                                              long rows, int C, int nslab, long grp_rows) {
  __shared__ float red[8 * 512];
  const int G = C >> 3;                       // requires 256 % G == 0 and C <= 512
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 256 / G;
  const int c0 = tg * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // statistics groups: blockIdx.y owns rows [y * grp_rows, (y + 1) * grp_rows) and the y-th [slots][2][C] block of sums
  const long rbeg = grp_rows > 0 ? (long)blockIdx.y * grp_rows : 0, rend = grp_rows > 0 ? rbeg + grp_rows : rows;
  if (ssum != nullptr) { ssum += (long)blockIdx.y * MSCL_STAT_SLOTS * 2 * C; ssq += (long)blockIdx.y * MSCL_STAT_SLOTS * 2 * C; }
  for (long r = rbeg + (long)blockIdx.x * RP + tr; r < rend; r += (long)gridDim.x * RP) {
    const long o = r * C + c0;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four slabs (and the addend) in flight per trip: with the loads inside a loop over the slabs every slab was its own memory
    // round trip (the compiler waits at the bottom of the loop), 2-16 of them per row on a pass whose rows make one or two trips
    uint4 av = make_uint4(0, 0, 0, 0);
    if (addend) av = *reinterpret_cast<const uint4*>(addend + o);
    const long sstride = rows * C;
    for (int sl = 0; sl < nslab; sl += 4) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* pp = partial + (long)(sl + u < nslab ? sl + u : sl) * sstride + o;      // (a short trip repeats its first slab; dropped below)
        a[u] = *reinterpret_cast<const float4*>(pp); b[u] = *reinterpret_cast<const float4*>(pp + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sl + u < nslab) {
          v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += v[i]; q[i] += v[i] * v[i]; }
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (addend) { float e[8]; unpack8(av, e);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += e[i]; }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f); }
    *reinterpret_cast<uint4*>(out + o) = pack8(v);
  }
  if (ssum == nullptr) return;
  block_channel_sum(s, red, G, C, 2, 0);
  block_channel_sum(q, red, G, C, 2, 1);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    const int so = (int)(blockIdx.x % MSCL_STAT_ACTIVE) * 2 * C;
    atomicAdd(&ssum[so + i], red[i] + red[C + i] + red[2 * C + i] + red[3 * C + i]);
    atomicAdd(&ssq[so + i], red[4 * C + i] + red[5 * C + i] + red[6 * C + i] + red[7 * C + i]);
  }
}

// ---------------------------------------------------------------------------------------- host side
// the pass that follows a split-K launch (also used by conv_pp.hip)
int mscl_launch_splitk_finalize(const float* partial, bf16_t* out, const float* bias, const bf16_t* addend, int relu, float* ssum,
                                float* ssq, long rows, int C, int nslab, long grp_rows, hipStream_t st) {
  const int RP = 256 / (C / 8);
  const int ngrp = grp_rows > 0 ? (int)(rows / grp_rows) : 1;
  constexpr long fin_cap = 2048;
  long fb = (rows / ngrp + RP - 1) / RP; if (fb > fin_cap / ngrp) fb = fin_cap / ngrp; if (fb < 1) fb = 1;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3((unsigned)fb, ngrp), dim3(256), 0, st, partial, out, bias, addend, relu, ssum, ssq,
                     rows, C, nslab, grp_rows);
  MSCL_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
static constexpr bool fast_tile() { return BK == 64 && STAGES == 2 && BM % (8 * WAVES_M * WAVES_N) == 0 && BN % (8 * WAVES_M * WAVES_N) == 0; }
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
static int launch_cfg(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias,
                      const bf16_t* addend, float* ssum, float* ssq, int relu, float* ws, long ws_floats, hipStream_t st) {
  int maxM = g.M;
  if (g.nclass > 0) { maxM = 0; for (int c = 0; c < g.nclass; ++c) maxM = g.cls[c].M > maxM ? g.cls[c].M : maxM; }
  g.mtiles = (maxM + BM - 1) / BM;
  g.ntiles = (g.Cr + BN - 1) / BN;
  const int ncls = g.nclass > 0 ? g.nclass : 1;
  const long blocks = (long)g.mtiles * g.ntiles * ncls;
  int maxtl = g.ntaps;
  if (g.nclass > 0) { maxtl = 0; for (int c = 0; c < g.nclass; ++c) maxtl = g.cls[c].ntl > maxtl ? g.cls[c].ntl : maxtl; }
  const int nk = ((maxtl << g.cgs) + BK / 8 - 1) / (BK / 8);
  g.ksplit = 1;
  const long out_elems = (long)g.N * g.Tr * g.Hr * g.Wr * g.Cr;
  if (ws != nullptr && blocks <= 256 && nk >= 32) {           // too few tiles for 256 CUs and a long K loop
    // measured on the 6272- and 784-position layers (a sweep of round 2, its tuning hooks removed in round 3): one round of <= 2 blocks per CU beats more
    // splits (98 tiles x 6 = 588 blocks ran 52 us, x 4 = 392 blocks 44 us), and a block wants >= 12 K steps
    constexpr long ks_target = 448, ks_steps = 12;        // swept in round 2 (320 / 448 / 640 blocks, 8 / 12 / 18 steps): kept
    long want = (WAVES_M * WAVES_N == 8 ? ks_target * 256 / 448 : ks_target) / blocks;      // wide tiles: one 96-KB block per CU
    const long minsteps = WAVES_M * WAVES_N == 8 ? (ks_steps + 1) / 2 : ks_steps;
    if (want > nk / minsteps) want = nk / minsteps;
    if (want > 16) want = 16;
    if (want * out_elems > ws_floats) want = ws_floats / out_elems;
    if (want > 1) g.ksplit = (int)want;
  }
  float* partial = g.ksplit > 1 ? ws : nullptr;
  bool launched = false;
  if constexpr (fast_tile<BM, BN, BK, WAVES_M, WAVES_N, STAGES>()) {
    // uniform-tap kernel: whole 64-channel K steps of one tap, no parity classes, offsets below 2^31
    const long span = ((long)g.N * g.Ts * g.Hs * g.Ws + 2L * (((long)g.kT * g.Hs + g.kH) * g.Ws + g.kW)) * g.Cs * 2;
    if (g.mode != 2 && g.cgs >= 3 && span < (1L << 31)) {
      g.dKW = make_fastdiv(g.kW); g.dKH = make_fastdiv(g.kH);
      const long nblk = blocks * g.ksplit;
      // one LDS stage for one-step layers into <= 64 channels (measured on the SlowOnly-50 shapes, A/B in one process: 64 -> 64 1x1x1
      // forward 32.9 -> 29.7 us, its input gradient 26.7 -> 23.8; the write-heavy 64 -> 256 lost 4 %, so wide outputs keep two stages)
      if (nk == 1 && g.ksplit == 1 && g.Cr <= 64) {       // (round 4 A/B: 29.7 vs 32.9 us on the SlowOnly-50 64 -> 64 1x1x1 conv; wider outputs lose)
        auto kern1 = conv_igemm_fast_kernel<BM, BN, WAVES_M, WAVES_N, 1>;
        hipLaunchKernelGGL(kern1, dim3((unsigned)nblk), dim3(64 * WAVES_M * WAVES_N), (size_t)(BM + BN) * BK * 2, st, g, src, wgt, out, bias,
                           addend, ssum, ssq, relu, partial);
      } else {
        auto kern = conv_igemm_fast_kernel<BM, BN, WAVES_M, WAVES_N, 2>;
        static bool attr_done_f = false;
        if (!attr_done_f) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_done_f = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WAVES_M * WAVES_N), (size_t)2 * (BM + BN) * BK * 2, st, g, src, wgt, out, bias,
                           addend, ssum, ssq, relu, partial);
      }
      MSCL_LAUNCH_CHECK();
      launched = true;
    }
  }
  if constexpr (WAVES_M * WAVES_N != 4) {
    if (!launched) return MSCL_E_SHAPE;              // wide tiles exist in the uniform-tap kernel only (callers check wide_ok first)
  } else if (!launched) {
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * 2 + (size_t)g.ntaps * 12;
    auto kern = conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, STAGES>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks * g.ksplit)), dim3(256), lds, st, g, src, wgt, out, bias, addend, ssum, ssq,
                       relu, partial);
    MSCL_LAUNCH_CHECK();
  }
  if (g.ksplit > 1)
    return mscl_launch_splitk_finalize(partial, out, bias, addend, relu, ssum, ssq, out_elems / g.Cr, g.Cr, g.ksplit, (long)g.grp_rows, st);
  return 0;
}

static int launch_igemm(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias,
                        const bf16_t* addend, float* ssum, float* ssq, int relu, float* ws, long ws_floats, hipStream_t st) {
  if ((long)g.N * g.Ts * g.Hs * g.Ws * g.Cs >= (1L << 31) || (long)g.N * g.Tr * g.Hr * g.Wr * g.Cr >= (1L << 31)) return MSCL_E_SHAPE;
  if ((long)g.Cr * g.KG * 16 >= (1L << 31)) return MSCL_E_SHAPE;
  if (g.Cr > 512 || ilog2_exact(g.Cr / 8) < 0) ws = nullptr;          // finalize kernel's thread layout
  const bool bk64 = (g.Cs % 64) == 0;
  const int Cr = g.Cr;
  int rowsM = g.M;
  if (g.nclass > 0) { rowsM = 0; for (int c = 0; c < g.nclass; ++c) rowsM += g.cls[c].M; }
  auto blocks = [&](int bm, int bn) { return (long)((rowsM + bm - 1) / bm) * ((Cr + bn - 1) / bn); };
#define GO3(BM, BN, BK, WMv, WNv, ST) return launch_cfg<BM, BN, BK, WMv, WNv, ST>(g, src, wgt, out, bias, addend, ssum, ssq, relu, ws, ws_floats, st)
#define GO(BM, BN, BK, WMv, WNv) GO3(BM, BN, BK, WMv, WNv, 2)
  // ping-pong kernel with shared W taps (conv_pp.hip): kW = 3, stride 1 along W, source channels a multiple of 64, output
  // channels a multiple of 128.  MSCL_PP: 0 = off, 1 (default) = layers of >= 400 k outputs (784 positions x 512 channels and up:
  // measured faster on every such shape of the step; the 784 x 128 pyramid level is not), 2 = wherever it applies.
  {
    static MsclTune t_pp("MSCL_PP");
    const int pp_level = t_pp.get(1);
    if (pp_level > 0 && (pp_level >= 2 || (long)rowsM * Cr >= 400000L)) {
      const int r = mscl_launch_conv_pp(g, src, wgt, out, bias, addend, ssum, ssq, relu, ws, ws_floats, st);
      if (r != MSCL_PP_SKIP) return r;
    }
  }
  const bool can_split = ws != nullptr;
  // wide tiles (uniform-tap kernel only: whole 64-channel steps of one tap, no parity classes, 32-bit offsets)
  const long span_w = ((long)g.N * g.Ts * g.Hs * g.Ws + 2L * (((long)g.kT * g.Hs + g.kH) * g.Ws + g.kW)) * g.Cs * 2;
  const bool wide_ok = bk64 && g.mode != 2 && g.cgs >= 3 && span_w < (1L << 31);
  if (wide_ok && Cr >= 128) {
    const int nk = g.ntaps << (g.cgs - 3);
    // 128 output channels: 256 positions x 128 (196 tiles on the 50176-position maps) -- a tie at best, A/B only
    if (Cr == 128 && (blocks(256, 128) >= 160 || (can_split && nk >= 32 && blocks(256, 128) >= 8))) GO(256, 128, 64, 4, 2);
    // 256 output channels on maps of a few thousand positions: 128 positions x 256 channels, K split over the grid
    if (Cr >= 256 && Cr % 256 == 0 && can_split && nk >= 32 && blocks(128, 256) <= 128)
      GO(128, 256, 64, 2, 4);
  }
  if (bk64) {
    if (Cr >= 128) {
      // a few dozen 128 x 128 tiles (784-position maps): 64-row tiles double the tiles per split, so fewer fp32 slabs
      // make the same number of blocks (512 -> 512 3x3x3 on 784 positions: 34 -> 32 us, 128 -> 128 on 6272: 30 -> 25 us)
      if (can_split && g.nclass == 0 && blocks(128, 128) <= 64) GO(64, 128, 64, 2, 2);
      if (blocks(128, 128) >= 384 || can_split) GO(128, 128, 64, 2, 2);
      GO(64, 128, 64, 2, 2);
    }
    if (Cr > 32) {
      // strided input gradient into 64 channels: a class has 1-8 taps, i.e. 2-16 K steps of 64; half-depth steps pipeline
      // these short loops better (97 -> 84 us on the layer-2 entry conv)
      // Round 3: every tile of either kernel runs this launch in 77-92 us (128 x 64 x 32 general 79, 128 x 64 x 64 uniform-tap 79,
      // 256 x 64 77-84, 64 x 64 83-92): with 64 output channels a K step stages 24 KB for 0.5 M MACs = 22 MAC/B, and 11.1 GMAC at the
      // ~6.3 TB/s of LDS-DMA fill is 81 us.  Only keeping dy resident across a cell's 27 (tap, class) uses would lift it.  (The
      // uniform-tap kernel with parity classes was built: 42 -> 38 us and 37 -> 36 us on the 128- / 256-channel strided layers, but
      // the class logic cost its dense uses more inside the step: 1013-1017 vs 1021-1023 clip-pairs/s.  Not kept.)
      if (g.nclass > 0 && Cr <= 64) GO(128, 64, 32, 2, 2);
      if (blocks(128, 64) >= 384 || can_split) GO(128, 64, 64, 2, 2);   // 48 KB LDS: 3 blocks/CU
      GO(64, 64, 64, 2, 2);
    }
    if (Cr > 16) GO(256, 32, 64, 4, 1);
    GO(256, 16, 64, 4, 1);
  }
  if (Cr >= 128) GO(128, 128, 32, 2, 2);
  if (Cr > 32) GO(256, 64, 32, 4, 1);
  if (Cr > 16) GO(256, 32, 32, 4, 1);
  GO(256, 16, 32, 4, 1);
#undef GO
#undef GO3
}

static int check_desc(const mscl_conv_desc* d) {
  if (!d) return MSCL_E_ARG;
  if (d->N <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return MSCL_E_ARG;
  if (d->C % 8 || d->K % 8) return MSCL_E_SHAPE;
  if (ilog2_exact(d->C / 8) < 0 || ilog2_exact(d->K / 8) < 0) return MSCL_E_SHAPE;
  if (d->kT < 1 || d->kH < 1 || d->kW < 1 || d->kT > 8 || d->kH > 8 || d->kW > 8) return MSCL_E_SHAPE;
  if (d->sT < 1 || d->sH < 1 || d->sW < 1) return MSCL_E_ARG;
  if (d->To != (d->T + 2 * d->pT - d->kT) / d->sT + 1 || d->Ho != (d->H + 2 * d->pH - d->kH) / d->sH + 1 ||
      d->Wo != (d->W + 2 * d->pW - d->kW) / d->sW + 1) return MSCL_E_SHAPE;
  if (d->T + d->pT >= 1000 || d->H + d->pH >= 1000 || d->W + d->pW >= 1000) return MSCL_E_SHAPE;
  return 0;
}

extern "C" int mscl_conv_halo64(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                                const uint16_t* addend, float* ssum, float* ssq, void* stream);
// layer-1 shape (3x3x3 s1 p1, 64 -> 64): halo-resident kernel, 131 / 109 us vs 156 / 135 us (fwd / dgrad) for the
// implicit-GEMM kernel; MSCL_HALO=0 switches it off
static bool halo_enabled(const mscl_conv_desc* d) {
  static MsclTune t("MSCL_HALO");
  if (t.read() && t.c0 == '0') return false;
  if (t.read() && t.c0 == '1') return true;                 // forced (tests: small planes too)
  return (long)d->H * (d->W + 2) >= 1024;                   // 256-position tiles: planes of a few hundred positions waste them
}

int mscl_conv_thin(int planes, int H, int W, int C, int K, int flip, const bf16_t* x, const bf16_t* w, bf16_t* y, const float* bias,
                   const bf16_t* addend, int relu, float* ssum, float* ssq, int stat_groups, hipStream_t st);      // conv_thin.hip
int mscl_conv_stem(const mscl_conv_desc* d, const bf16_t* x, const bf16_t* w, bf16_t* y, float* ssum, float* ssq, hipStream_t st);   // conv_stem.hip
int mscl_conv_k1(long M, int K, int N, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const bf16_t* addend, float* ssum, float* ssq,
                 hipStream_t st);                                                                                      // conv_k1.hip
int mscl_launch_dgrad_s2(const mscl_conv_desc* d, const bf16_t* dy, const bf16_t* wT, bf16_t* dx, const bf16_t* addend, hipStream_t st);   // conv_dgrad_s2.hip
static bool unit_1x1x1(const mscl_conv_desc* d) {
  return d->kT == 1 && d->kH == 1 && d->kW == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 && d->pH == 0 && d->pW == 0;
}
static bool thin_shape(const mscl_conv_desc* d) {
  return d->kT == 1 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 && d->pH == 1 && d->pW == 1 &&
         (d->C == 16 || d->C == 32) && (d->K == 16 || d->K == 32);
}

extern "C" int mscl_conv3d_fwd(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* w, uint16_t* y,
                               const float* bias, const uint16_t* addend, int relu, float* ssum, float* ssq,
                               float* splitk_ws, int64_t splitk_ws_floats, void* stream) {
  return mscl_conv3d_fwd_groups(d, x, w, y, bias, addend, relu, ssum, ssq, 1, splitk_ws, splitk_ws_floats, stream);
}

extern "C" int mscl_conv3d_fwd_groups(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* w, uint16_t* y,
                                      const float* bias, const uint16_t* addend, int relu, float* ssum, float* ssq,
                                      int stat_groups, float* splitk_ws, int64_t splitk_ws_floats, void* stream) {
  int e = check_desc(d); if (e) return e;
  if (!x || !w || !y) return MSCL_E_ARG;
  if ((ssum == nullptr) != (ssq == nullptr)) return MSCL_E_ARG;
  if (stat_groups < 1 || d->N % stat_groups != 0) return MSCL_E_ARG;
  if (stat_groups > 2) return MSCL_E_SHAPE;         // the epilogue splits a tile over two statistics groups at most
  if (ssum != nullptr && mscl_det()) {
    // deterministic mode: no statistics in the epilogue (LDS and global float atomics); a fixed-order pass over the stored map
    // fills slot 0 instead (per-block partials in the workspace, folded in index order: bn_act.hip) -- the statistics of the bf16
    // values the next layer reads.  The workspace is free again by then: the split-K finalize ran before on this stream.
    e = mscl_conv3d_fwd_groups(d, x, w, y, bias, addend, relu, nullptr, nullptr, 1, splitk_ws, splitk_ws_floats, stream);
    if (e) return e;
    return mscl_bn_stats(y, ssum, ssq, (int64_t)d->N * d->To * d->Ho * d->Wo, d->K, stat_groups, splitk_ws, splitk_ws_floats, stream);
  }
  if (thin_shape(d)) {             // 1x3x3 s1 between 16- / 32-channel maps: window-resident direct kernel (conv_thin.hip)
    const int h = mscl_conv_thin(d->N * d->T, d->H, d->W, d->C, d->K, 0, x, w, y, bias, addend, relu, ssum, ssq, stat_groups,
                                 (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  if (stat_groups == 1 && bias == nullptr && addend == nullptr && !relu) {  // W-paired RGB stem: window-resident kernel (conv_stem.hip)
    const int h = mscl_conv_stem(d, x, w, y, ssum, ssq, (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  if (stat_groups == 1 && bias == nullptr && !relu && halo_enabled(d)) {   // 3x3x3 s1 64->64: halo-resident kernel (conv_halo.hip)
    const int h = mscl_conv_halo64(d, 0, x, w, y, addend, ssum, ssq, stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  if (stat_groups == 1 && bias == nullptr && !relu && unit_1x1x1(d)) {     // thin-K, wide-N 1x1x1: persistent streaming kernel (conv_k1.hip)
    const int h = mscl_conv_k1((long)d->N * d->T * d->H * d->W, d->C, d->K, x, w, y, addend, ssum, ssq, (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  IGemmGeom g{};
  g.N = d->N; g.Ts = d->T; g.Hs = d->H; g.Ws = d->W; g.Cs = d->C;
  g.Tr = d->To; g.Hr = d->Ho; g.Wr = d->Wo; g.Cr = d->K;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  g.M = d->N * d->To * d->Ho * d->Wo; g.ntaps = d->kT * d->kH * d->kW;
  g.cgs = ilog2_exact(d->C / 8); g.KG = g.ntaps * (d->C / 8); g.mode = 0; g.nclass = 0;
  g.grp_rows = (stat_groups > 1 && ssum != nullptr) ? (d->N / stat_groups) * d->To * d->Ho * d->Wo : 0;
  g.dW = make_fastdiv(g.Wr); g.dH = make_fastdiv(g.Hr); g.dT = make_fastdiv(g.Tr);
  return launch_igemm(g, x, w, y, bias, addend, ssum, ssq, relu, splitk_ws, (long)splitk_ws_floats, (hipStream_t)stream);
}

extern "C" int mscl_conv3d_dgrad(const mscl_conv_desc* d, const uint16_t* dy, const uint16_t* wT, uint16_t* dx,
                                 const uint16_t* addend, float* splitk_ws, int64_t splitk_ws_floats, void* stream) {
  int e = check_desc(d); if (e) return e;
  if (!dy || !wT || !dx) return MSCL_E_ARG;
  if (thin_shape(d)) {
    const int h = mscl_conv_thin(d->N * d->T, d->H, d->W, d->K, d->C, 1, dy, wT, dx, nullptr, addend, 0, nullptr, nullptr, 1,
                                 (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  if (halo_enabled(d)) {
    const int h = mscl_conv_halo64(d, 1, dy, wT, dx, addend, nullptr, nullptr, stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  if (unit_1x1x1(d)) {
    const int h = mscl_conv_k1((long)d->N * d->T * d->H * d->W, d->K, d->C, dy, wT, dx, addend, nullptr, nullptr, (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  {   // 3x3x3 / stride 2 entry conv of a stage on a large map: dy window resident in LDS, all eight parity classes from it (conv_dgrad_s2.hip)
    const int h = mscl_launch_dgrad_s2(d, dy, wT, dx, addend, (hipStream_t)stream);
    if (h != 0) return h == 1 ? 0 : h;
  }
  IGemmGeom g{};
  g.N = d->N; g.Ts = d->To; g.Hs = d->Ho; g.Ws = d->Wo; g.Cs = d->K;
  g.Tr = d->T; g.Hr = d->H; g.Wr = d->W; g.Cr = d->C;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  g.M = d->N * d->T * d->H * d->W; g.ntaps = d->kT * d->kH * d->kW;
  g.cgs = ilog2_exact(d->K / 8); g.KG = g.ntaps * (d->K / 8);
  const bool unit = d->sT == 1 && d->sH == 1 && d->sW == 1;
  g.mode = unit ? 1 : 2; g.nclass = 0;
  g.dW = make_fastdiv(g.Wr); g.dH = make_fastdiv(g.Hr); g.dT = make_fastdiv(g.Tr);
  if (!unit) {
    g.lsT = ilog2_exact(d->sT); g.lsH = ilog2_exact(d->sH); g.lsW = ilog2_exact(d->sW);
    if (g.lsT < 0 || g.lsH < 0 || g.lsW < 0 || d->sT > 2 || d->sH > 2 || d->sW > 2) return MSCL_E_STRIDE;
    // stride-parity classes: input position i receives tap k only if (i + p - k) % s == 0
    // IN PLACE (addend == dx): positions of a class without taps keep what dx holds, so such a class is not launched at all --
    // the strided 1x1x1 shortcut of a stage entry then touches 1/8 (1/4) of the map it adds to instead of writing a map of
    // mostly zeros that the entry conv's input gradient reads back as its addend (nn._BlockFn / _BottleneckFn)
    const bool inplace = addend != nullptr && addend == dx;
    int nc = 0;
    bool dropped = false;
    for (int a = 0; a < d->sT; ++a) for (int b = 0; b < d->sH; ++b) for (int c = 0; c < d->sW; ++c) {
      ClassInfo& ci = g.cls[nc];
      ci.ro[0] = (unsigned char)a; ci.ro[1] = (unsigned char)b; ci.ro[2] = (unsigned char)c;
      ci.TrS = (d->T - a + d->sT - 1) / d->sT; ci.HrS = (d->H - b + d->sH - 1) / d->sH; ci.WrS = (d->W - c + d->sW - 1) / d->sW;
      ci.M = d->N * ci.TrS * ci.HrS * ci.WrS;
      ci.dW = make_fastdiv(ci.WrS > 0 ? ci.WrS : 1); ci.dH = make_fastdiv(ci.HrS > 0 ? ci.HrS : 1); ci.dT = make_fastdiv(ci.TrS > 0 ? ci.TrS : 1);
      int n = 0;
      for (int kt = 0; kt < d->kT; ++kt) for (int kh = 0; kh < d->kH; ++kh) for (int kw = 0; kw < d->kW; ++kw) {
        if ((a + d->pT - kt) % d->sT == 0 && (b + d->pH - kh) % d->sH == 0 && (c + d->pW - kw) % d->sW == 0) {
          if (n >= 8) return MSCL_E_SHAPE;            // kernels wider than 4 taps per parity are not needed here
          ci.taps[n++] = (unsigned char)((kt * d->kH + kh) * d->kW + kw);
        }
      }
      ci.ntl = (unsigned char)n;
      if (ci.M > 0 && inplace && n == 0) dropped = true;
      if (ci.M > 0 && !(inplace && n == 0)) ++nc;
    }
    if (nc == 0) return 0;                            // (in place and no position receives a tap: nothing to add)
    // classes were dropped: no split-K -- its finalize pass walks EVERY position of dx and would add slab rows nobody wrote
    // (reachable from 1024 dy channels up: K steps >= 32 and few tiles, i.e. the small-map Bottleneck shortcuts)
    if (dropped) { splitk_ws = nullptr; splitk_ws_floats = 0; }
    g.nclass = nc;
  }
  return launch_igemm(g, dy, wT, dx, nullptr, addend, nullptr, nullptr, 0, splitk_ws, (long)splitk_ws_floats, (hipStream_t)stream);
}

__global__ void weight_transpose_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wT, int Cout, int taps, int Cin) {
  const long total = (long)Cout * taps * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout); const long r = i / Cout;       // i indexes wT[ci][tap][co]
    const int tap = (int)(r % taps); const int ci = (int)(r / taps);
    wT[i] = w[((long)co * taps + tap) * Cin + ci];
  }
}

extern "C" int mscl_weight_transpose(const uint16_t* w, uint16_t* wT, int Cout, int taps, int Cin, void* stream) {
  if (!w || !wT || Cout <= 0 || taps <= 0 || Cin <= 0) return MSCL_E_ARG;
  const long total = (long)Cout * taps * Cin;
  const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wT, Cout, taps, Cin);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// all kernels of a model in one launch: table[i] = {src ptr, dst ptr, Cout, taps, Cin, first block}
struct TransposeEntry { const bf16_t* w; bf16_t* wT; int Cout, taps, Cin, first_block; };
// Entry i owns blocks [first_block_i, first_block_{i+1}).  Kernels whose channel counts are multiples of 64 are
// moved as 64x64 (co x ci) tiles of one tap through LDS: 128-byte rows in, 128-byte rows out (the element-wise
// form reads 2-byte elements at a stride of taps*Cin and took 460 us for the 37 M weights of a step).
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const TransposeEntry* __restrict__ table, int n) {
  __shared__ bf16_t tile[64][66];
  int e = 0;
  for (int i = 1; i < n; ++i) if ((int)blockIdx.x >= table[i].first_block) e = i;      // uniform scan, n ~ 40
  const TransposeEntry t = table[e];
  const int lb = blockIdx.x - t.first_block;
  if ((t.Cout & 63) == 0 && (t.Cin & 63) == 0) {
    const int cit = t.Cin >> 6, cot = t.Cout >> 6;
    const int tiles = cit * cot * t.taps;
    if (lb >= tiles) return;
    const int ci0 = (lb % cit) << 6; const int r = lb / cit;
    const int co0 = (r % cot) << 6; const int tap = r / cot;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {            // 64 rows (co) x 8 granules (ci)
      const int row = i >> 3, gq = i & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(t.w + ((long)(co0 + row) * t.taps + tap) * t.Cin + ci0 + gq * 8);
      const bf16_t* pv = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[row][gq * 8 + k] = pv[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {            // 64 rows (ci) x 8 granules (co)
      const int row = i >> 3, gq = i & 7;
      bf16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = tile[gq * 8 + k][row];
      *reinterpret_cast<uint4*>(t.wT + ((long)(ci0 + row) * t.taps + tap) * t.Cout + co0 + gq * 8) = *reinterpret_cast<uint4*>(o);
    }
    return;
  }
  const long total = (long)t.Cout * t.taps * t.Cin;
  const long i = (long)lb * 256 + threadIdx.x;
  if (i >= total) return;
  const int co = (int)(i % t.Cout); const long r = i / t.Cout;
  const int tap = (int)(r % t.taps); const int ci = (int)(r / t.taps);
  t.wT[i] = t.w[((long)co * t.taps + tap) * t.Cin + ci];
}
extern "C" int mscl_weight_transpose_batched(const void* table, int n, int total_blocks, void* stream) {
  if (!table || n <= 0 || total_blocks <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const TransposeEntry*>(table), n);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_abi_version(void) { return 1; }

int g_mscl_deterministic = 0;
int g_mscl_tune_gen = 0;
extern "C" int mscl_tuning_reload(void) { ++g_mscl_tune_gen; return 0; }
extern "C" int mscl_set_deterministic(int on) { g_mscl_deterministic = on ? 1 : 0; return 0; }
extern "C" int mscl_get_deterministic(void) { return g_mscl_deterministic; }
