// Conv3d forward / stride-1 input gradient for the 128-and-wider channel layers: implicit GEMM, bf16 MFMA, one 512-thread block per
// CU, ping-pong schedule (replaces the two-barrier-per-K-step loop of conv_igemm_fast_kernel on these shapes).
// Reference ops: the 3x3x3 convolutions of BasicBlock / Conv3DSimple (mmaction/models/backbones/r3d.py:16-34,95-127), the SEPC
// PConv3D convolutions (necks/sepc.py:57-135) and the FPN output convolutions (necks/fpn.py:130-152).
//
// What bounds an im2col-style conv on a CU is the global -> LDS path: ~30 B/clk/CU (MI355X_MICROARCH.md, "Indexed rows: gather into
// LDS"), i.e. ~100+ issue cycles per 1-KiB LDS-DMA instruction, against 16 cycles per v_mfma_f32_16x16x32_bf16.  A 256 x 128 x 64
// step re-staged per tap needs 6 DMA instructions per wave per 32 MFMAs.  Two changes take that to 3.3:
//  * SHARED W TAPS.  Rows of the GEMM are positions in PADDED-LINEAR order: q = (n,t,h) * (W + 2) + wp, wp = 0 and W + 1 being
//    zero columns (the buffer unit writes the zeros: out-of-range offset).  The three kw taps of a (kt, kh) pair are then the
//    SAME LDS rows read one row up / in place / one row down, so the position tile is staged once per (kt, kh, 64 channels) and
//    only the three 128 x 64 weight tiles differ.  A block stages 256 rows and stores the 254 inner ones (tiles overlap by 2).
//    Cost: 2 / (W + 2) of the MFMA work lands on padding columns (6.7 % at W = 28).
//  * PING-PONG.  8 waves = 2 per SIMD.  Every wave runs [L: 8 ds_read_b128 + its share of the DMA issue | barrier | M: 16 MFMAs
//    at raised priority | barrier]; waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave feeds the matrix pipe
//    while its partner loads.  DMA stays in flight across the (raw) barriers behind counted vmcnt waits, never 0 inside the loop.
// Hazard rules (cdna_hip_programming.md 5, "Read a staged buffer one phase AFTER the wait that retires it"), phases numbered per
// wave:  (R1) a unit waited for in L_p is first read in L_{p+1};  (R2) a slot last read in L_p is re-issued in L_{p+1} or later --
// one phase suffices because every wave retires its fragment reads (lgkmcnt(0)) BEFORE the barrier that ends its L section, so no
// read of L_p is in flight once any wave starts L_{p+1}.  With the one-barrier stagger both hold for either wave group.
// Schedule per group g = (kt, kh, 64-channel part): 3 phases Q0..Q2 = the kw taps, each a whole 64-deep K tile (16 ds_read_b128,
// 32 MFMAs per wave); UB = BN / 64 DMA instructions per weight tile, 2 per half position tile; ring of NB = 4 weight slots (tile
// index mod 4), 2 position slots:
//   issue for group g+1:   Q0: A0' A1'   Q1: B0' B1'   Q2: B2'
//   waits:  Q0: B1 (of g) landed = vmcnt(UB + 4)   Q1: B2 landed = vmcnt(2 UB + 4)   Q2: A0', A1', B0' landed = vmcnt(2 UB)
// Round 4, measured and dropped: issuing the DMA pieces from the M section instead of L -- after the MFMAs (in the time a wave
// otherwise spends at the closing barrier) or ahead of them -- with the waits left in L and their counts lowered by the pieces no
// longer ahead of them.  Every shape lost 1-8 % (A/B in one process: 128 -> 128 forward 46.1 vs 50.0 / 50.5 us, input gradient
// 42.1 vs 45.2 / 45.4; 256 -> 256 37.5 vs 39.0 / 39.3; 512 -> 512 28.8 vs 29.8 / 29.2): the L section is not what the pieces'
// issue cost lengthens -- a piece issued half a phase later lands half a phase later, and the counted waits of the next L
// sections then find it still in flight.
#include "igemm.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> struct IC { static constexpr int value = N; };

#define PP_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define PP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

// Timing-study build (-DPP_STAMP, tools/pp_stamps.py; never the shipped library): per block and wave group the cycles spent in the
// L sections, at the barrier behind them, in the M sections and at the barrier behind those (s_memtime; the stamps themselves cost
// ~10 % of the wave cycles).  The values go to a buffer of their own and feed no output.
#ifdef PP_STAMP
__device__ unsigned long long g_pp_stamps[2048 * 2 * 8];
extern "C" int mscl_debug_pp_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

template <int BN>
__global__ __launch_bounds__(512) void conv_pp_kernel(
    const IGemmGeom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
    const float* __restrict__ bias, const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
    float* __restrict__ stat_sq, const int relu, float* __restrict__ partial) {
  constexpr int BM = 256, KW = 3, HALO = 1, OUT_ROWS = BM - 2 * HALO, NB = 4;
  constexpr int A_SLOT = BM * 128, B_SLOT = BN * 128;
  constexpr int UB = BN / 64;                     // DMA instructions per thread per weight tile (64 rows per pass of 512 threads)
  constexpr int A_BASE = NB * B_SLOT;             // weight ring first: the row "-1" read of position fragment 0 stays inside LDS
  // 8 waves: 4 (positions) x 2 (channels) of 64 x 64 at BN = 128; 8 x 1 of 32 x 64 at BN = 64
  constexpr int WAVES_N = BN / 64, WAVES_M = 8 / WAVES_N, WM = BM / WAVES_M, WN = 64;
  constexpr int IM = WM / 16, JN = WN / 16;
  constexpr unsigned OOB = 0x80000000u;
  static_assert((BN == 128 || BN == 64) && JN == 4 && (IM == 4 || IM == 2), "tile config");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
#ifdef PP_STAMP
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                      // 0: waves 0-3, 1: waves 4-7 (the SIMD partners of 0-3), one barrier behind
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % g.ksplit; bid /= g.ksplit;
  const int nt = bid % g.ntiles; const int mt = bid / g.ntiles;
  const int n0 = nt * BN;
  const int q0 = mt * OUT_ROWS - HALO;            // padded-linear position of LDS row 0
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);      // 0 forward, 1 stride-1 input gradient
  const int Wp = g.Wr + 2 * HALO, Mp = g.M;
  const int cs2 = g.Cs * 2;
  // tap offsets (kt, kh only: kw is a row shift at read time) stay non-negative in the SGPR operand: descriptor base moved back
  const int maxlin = ((g.kT - 1) * g.Hs + (g.kH - 1)) * g.Ws;
  const int padlin = (g.pT * g.Hs + g.pH) * g.Ws + HALO;
  const int bias_bytes = (mode == 0 ? padlin : maxlin) * cs2;

  const int rg = tid & 7, rr = tid >> 3;          // granule column / row inside a 64-row staging pass
  // LDS images are [row][granule ^ (row & 7)]: conflict-free ds_read_b128 fragments under the -1 / 0 / +1 row shifts of the shared
  // taps (measured against (row >> 1) & 7, which the two-barrier kernels use: 2-way conflicts on the odd shifts, +16 % on the
  // read-only loop; no swizzle: 3.3x)
  auto skey = [&](int row) { return row & 7; };
  const int rgl = rg ^ skey(rr);                  // logical granule this lane fetches
  unsigned wrow_voff[UB];
#pragma unroll
  for (int p = 0; p < UB; ++p) {
    const int r = p * 64 + rr;
    wrow_voff[p] = (n0 + r < g.Cr) ? (unsigned)((n0 + r) * g.KG * 16 + rgl * 16) : OOB;
  }
  const unsigned char* src_b = reinterpret_cast<const unsigned char*>(src) - bias_bytes;
  const auto rs_src = make_uniform_rsrc(src_b, 0x7FFFFFFFu);
  const auto rs_wgt = make_uniform_rsrc(wgt, 0x7FFFFFFFu);

  // groups of this split
  const int subs = g.cgs - 3, submask = (1 << subs) - 1;        // 64-channel parts per tap = Cs / 64
  const int ng_all = (g.kT * g.kH) << subs;
  const int g_beg = (int)((long)ng_all * split / g.ksplit), g_end = (int)((long)ng_all * (split + 1) / g.ksplit);
  // scalar state of one group: SGPR offset of the position tile, tap-validity bits, SGPR offset of its first weight tile
  auto group_soff = [&](int gi, unsigned& soff, int& tb, unsigned& woff) {
    const int tk = gi >> subs, cpart = gi & submask;
    const int kt = fdiv(tk, g.dKH), kh = tk - kt * g.kH;
    const int lin = (kt * g.Hs + kh) * g.Ws;
    soff = __builtin_amdgcn_readfirstlane((unsigned)((mode == 0 ? lin : maxlin - lin) * cs2 + cpart * 128));
    tb = __builtin_amdgcn_readfirstlane((1 << kt) | (1 << (8 + kh)));
    woff = __builtin_amdgcn_readfirstlane((unsigned)((((tk * KW) << subs) + cpart) * 128));      // kw = 0; kw adds Cs * 2 bytes
  };
  auto issue_b = [&](unsigned slot, unsigned woff) {
    unsigned char* b = smem + slot + wave * 1024;
#pragma unroll
    for (int p = 0; p < UB; ++p) {
      const unsigned wv = wrow_voff[p];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(b + p * 8192), 16, wv, woff, 0, 0);
    }
  };
  unsigned soff_n = 0, woff_n = 0; int tb_n = 0;
  const unsigned wstep = (unsigned)cs2;           // bytes from a (kt, kh, kw) weight tile to the (kt, kh, kw + 1) one
  // ---- prologue, part 1: the first group's first weight tile needs no row state; it travels while the rows are set up ----
  group_soff(g_beg, soff_n, tb_n, woff_n);
  issue_b(0 * B_SLOT, woff_n);
  int row_voff[4], row_mask[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = q0 + p * 64 + rr;
    int mask = 0, base = 0;
    if (q >= 0 && q < Mp) {
      const int nth = fdiv(q, g.dW), w0 = q - nth * Wp - HALO;
      if ((unsigned)w0 < (unsigned)g.Ws) {
        const int q2 = fdiv(nth, g.dH), hr = nth - q2 * g.Hr;
        const int n = fdiv(q2, g.dT), tr = q2 - n * g.Tr;
        int t0, h0;
        if (mode == 0) { t0 = tr * g.sT - g.pT; h0 = hr * g.sH - g.pH; }
        else { t0 = tr + g.pT; h0 = hr + g.pH; }
        // taps k with 0 <= t0 + k < Ts (forward) / 0 <= t0 - k < Ts (gradient) form a range [lo, hi): bits without a loop
        const int tlo = (mode == 0) ? max(0, -t0) : max(0, t0 - g.Ts + 1), thi = (mode == 0) ? min(g.kT, g.Ts - t0) : min(g.kT, t0 + 1);
        const int hlo = (mode == 0) ? max(0, -h0) : max(0, h0 - g.Hs + 1), hhi = (mode == 0) ? min(g.kH, g.Hs - h0) : min(g.kH, h0 + 1);
        mask = (thi > tlo ? ((1 << thi) - (1 << tlo)) : 0) | (hhi > hlo ? (((1 << hhi) - (1 << hlo)) << 8) : 0);
        base = ((n * g.Ts + t0) * g.Hs + h0) * g.Ws + w0;
      }
    }
    row_voff[p] = base * cs2 + (mode == 0 ? bias_bytes : 0) + rgl * 16;      // >= 0 for every row with a valid tap
    row_mask[p] = mask;
  }
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4_t acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment byte offsets inside a slot; fragments i / j are 16 rows = 2048 bytes apart (same swizzle key)
  unsigned a_off[KW][2], b_off[2];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = wm0 + fr + (mode == 0 ? kw - HALO : HALO - kw);        // may be -1 / 256 on the two rows that are not stored
      a_off[kw][ks] = lds_base + A_BASE + (unsigned)(row * 128 + (((ks * 4 + fq) ^ skey(row)) * 16));
    }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int row = wn0 + fr;
    b_off[ks] = lds_base + (unsigned)(row * 128 + (((ks * 4 + fq) ^ skey(row)) * 16));
  }

  auto issue_a = [&](int half, unsigned slot, unsigned soff, int tb) {     // rows [128 half, 128 half + 128) of a position tile
    unsigned char* a = smem + A_BASE + slot + wave * 1024;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int p = half * 2 + pp;
      const unsigned off = ((row_mask[p] & tb) == tb) ? (unsigned)row_voff[p] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(a + p * 8192), 16, off, soff, 0, 0);
    }
  };
  unsigned bq = 0;                                // ring slot (0..NB-1) of the current group's first weight tile
  unsigned a_cur = 0;                             // byte offset of the current group's position slot (0 / A_SLOT)

  // ---- prologue, part 2: the first position tile, then weight tiles 1 and 2 of the first group.  Only what phase 0 reads is waited
  // for (the position tile and weight tile 0: 48 of the 80 KB); tiles 1 and 2 stay in flight exactly where the steady-state waits
  // of Q0 / Q1 expect them (older than everything those phases issue).  Round 4: the stamps put this prologue at 5.5-7.4 k cycles
  // of a block's 32-87 k. ----
  issue_a(0, 0, soff_n, tb_n);
  issue_a(1, 0, soff_n, tb_n);
  issue_b(1 * B_SLOT, woff_n + wstep);
  issue_b(2 * B_SLOT, woff_n + 2 * wstep);
  PP_VMCNT(2 * UB);
  __builtin_amdgcn_s_barrier();

  u32x4_t fa[2][IM], fb[2][JN];
  // per-group scalar state: weight ring slots of this group's tiles (bq, bq+1, bq+2 mod 4) and of the next group's (bq+3, bq,
  // bq+1 mod 4), position slots, and the next group's SGPR offsets (has_next)
  int gi = g_beg;
  bool has_next = gi + 1 < g_end;
  if (has_next) group_soff(gi + 1, soff_n, tb_n, woff_n);
  auto advance = [&]() {
    a_cur ^= (unsigned)A_SLOT;
    bq = (bq + 3) & 3;
    ++gi;
    has_next = gi + 1 < g_end;
    if (has_next) group_soff(gi + 1, soff_n, tb_n, woff_n);
  };
  // ---- L: fragments of K tile kw of the current group, this wave's share of the next group's DMA, the counted wait; the
  // fragment reads are retired before the section ends (rule R2) ----
  auto Lsec = [&](auto PC) {
    constexpr int kw = decltype(PC)::value;
    const unsigned a_nxt = a_cur ^ (unsigned)A_SLOT;
    const unsigned bs1 = (bq + 1) & 3, bs2 = (bq + 2) & 3, bn0 = (bq + 3) & 3, bn1 = bq, bn2 = bs1;
    const unsigned bslot = (kw == 0 ? bq : (kw == 1 ? bs1 : bs2)) * (unsigned)B_SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned aa = a_off[kw][ks] + a_cur, ba = b_off[ks] + bslot;
      PP_DSR(fa[ks][0], aa, 0); PP_DSR(fa[ks][1], aa, 2048);
      if constexpr (IM == 4) { PP_DSR(fa[ks][2], aa, 4096); PP_DSR(fa[ks][3], aa, 6144); }
      PP_DSR(fb[ks][0], ba, 0); PP_DSR(fb[ks][1], ba, 2048); PP_DSR(fb[ks][2], ba, 4096); PP_DSR(fb[ks][3], ba, 6144);
    }
    if (has_next) {
      if constexpr (kw == 0) { issue_a(0, a_nxt, soff_n, tb_n); issue_a(1, a_nxt, soff_n, tb_n); PP_VMCNT(UB + 4); }
      if constexpr (kw == 1) { issue_b(bn0 * B_SLOT, woff_n); issue_b(bn1 * B_SLOT, woff_n + wstep); PP_VMCNT(2 * UB + 4); }
      if constexpr (kw == 2) { issue_b(bn2 * B_SLOT, woff_n + 2 * wstep); PP_VMCNT(2 * UB); }
    } else {
      if constexpr (kw == 0) PP_VMCNT(UB);
      if constexpr (kw == 1) PP_VMCNT(0);
    }
    if constexpr (IM == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]),
                     "+v"(fb[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1]),
                     "+v"(fb[1][2]), "+v"(fb[1][3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fa[1][0]),
                     "+v"(fa[1][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]));
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- M: one WM x 64 x 64 product from the fragments the last L section read ----
  auto Msec = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[ks][j]),
                                                              __builtin_bit_cast(bf16x8_t, fa[ks][i]), acc[j][i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const int ng = g_end - g_beg;
  // Two barriers per phase: [L | barrier | M | barrier], waves 4-7 one barrier behind.  (Measured and dropped: ONE barrier per
  // phase with the second group's sections rotated -- waves 0-3 [L_p M_p | barrier], waves 4-7 [M_{p-1} L_p | barrier] -- which
  // would make an interval L + M instead of 2 max(L, M): 52.4 vs 46.9 us on the 128 -> 128 layer, slower on every shape; without
  // the barrier between the sections the partners of a SIMD drift into loading together and multiplying together.)
  if (grp == 1) __builtin_amdgcn_s_barrier();
#ifdef PP_STAMP
  unsigned long long tL = 0, tW1 = 0, tM = 0, tW2 = 0, ta, tb;
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#define PP_PHASE(KW) ta = __builtin_amdgcn_s_memtime(); Lsec(IC<KW>{}); tb = __builtin_amdgcn_s_memtime(); tL += tb - ta; \
    __builtin_amdgcn_s_barrier(); ta = __builtin_amdgcn_s_memtime(); tW1 += ta - tb; Msec(); tb = __builtin_amdgcn_s_memtime(); tM += tb - ta; \
    __builtin_amdgcn_s_barrier(); ta = __builtin_amdgcn_s_memtime(); tW2 += ta - tb;
#else
#define PP_PHASE(KW) Lsec(IC<KW>{}); __builtin_amdgcn_s_barrier(); Msec(); __builtin_amdgcn_s_barrier();
#endif
  for (int n = 0; n < ng; ++n) {
    PP_PHASE(0) PP_PHASE(1) PP_PHASE(2)
    advance();
  }
#undef PP_PHASE
  if (grp == 0) __builtin_amdgcn_s_barrier();         // pairs with the last barrier of waves 4-7
#ifdef PP_STAMP
  const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue (shared): rows back from padded-linear order; the halo rows and the padding columns are not stored ----
  long orow[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int idx = wm0 + i * 16 + fr;
    const int q = q0 + idx;
    long o = -1;
    if (idx >= HALO && idx < BM - HALO && q < Mp) {
      const int nth = fdiv(q, g.dW), w0 = q - nth * Wp - HALO;
      if ((unsigned)w0 < (unsigned)g.Wr) o = ((long)nth * g.Wr + w0) * g.Cr;
    }
    orow[i] = o;
  }
  igemm_epilogue_rows<BM, BN, IM, JN, true>(g, acc, smem, tid, fr, fq, 0, n0, wm0, wn0, split, 0, orow, out, bias, addend, stat_sum,
                                            stat_sq, relu, partial);
#ifdef PP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 2048) {
    unsigned long long* o = g_pp_stamps + ((size_t)blockIdx.x * 2 + grp) * 8;
    o[0] = tL; o[1] = tW1; o[2] = tM; o[3] = tW2; o[4] = t_loop - t_begin; o[5] = t_end - t_loop_end; o[6] = t_end - t_begin; o[7] = 0;
  }
#endif
}

// Round 4, measured and dropped: the same kernel with TWO blocks per CU (128-row blocks, three weight slots, 80 KB of LDS, 104
// registers), built because the stamps of profiles/r04_pp_stamps.md put the main loop at ~80 % matrix-pipe busy and the launch's loss
// outside it (212 blocks on 256 CUs, the 80-KB prologue, the epilogue; 27 / 14 phases per block on the split-K layers).  Correct
// (parity + race screen) and a tie: 128 -> 128 forward / input gradient 47.5 / 43.5 vs 47.0 / 43.8 us, 256 -> 256 37.8 / 32.8 vs
// 38.6 / 33.7, 512 -> 512 32.5 / 30.7 vs 28.6 / 26.9, SEPC 44.3 / 41.4 vs 42.1 / 39.5, the 1x3x3 pyramid level 17.4 / 15.0 vs 20.7 /
// 18.1; step 1080-1083 vs 1078-1097 clip-pairs/s.  A half-size block halves the MFMAs of a phase but not its DMA instructions (the
// weight tile of a phase stays 16 KB) and reads 1.5 x the LDS bytes per MFMA: what the second block hides, the longer L sections and
// the doubled weight traffic give back.  (Unlike the layer-1 kernel, conv_halo.hip, whose window is read once whatever the block size.)

// ------------------------------------------------------------------------------------------------------------- host side
static long g_pp_launches = 0;
extern "C" int64_t mscl_debug_pp_launches(void) { return g_pp_launches; }     // tests: which kernel family took a launch

// Returns 0 when launched, MSCL_PP_SKIP when the shape is outside this kernel (the caller falls back to conv_igemm.hip).
int mscl_launch_conv_pp(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias, const bf16_t* addend,
                        float* ssum, float* ssq, int relu, float* ws, long ws_floats, hipStream_t st) {
  constexpr int OUT_ROWS = 254;
  const int BN = (g.Cr % 128 == 0) ? 128 : 64;
  constexpr long slots = 256;                     // blocks of one round
  if (g.mode == 2 || g.nclass != 0 || g.grp_rows != 0) return MSCL_PP_SKIP;
  if (g.kW != 3 || g.pW != 1 || g.sW != 1 || g.Wr != g.Ws) return MSCL_PP_SKIP;
  if (g.cgs < 3 || (g.Cs & 63) != 0 || (g.Cr % 64) != 0) return MSCL_PP_SKIP;
  if (g.kT > 8 || g.kH > 8) return MSCL_PP_SKIP;
  const long span = ((long)g.N * g.Ts * g.Hs * g.Ws + 2L * (((long)g.kT * g.Hs + g.kH) * g.Ws + g.kW)) * g.Cs * 2;
  if (span >= (1L << 31) || (long)g.Cr * g.KG * 16 >= (1L << 31)) return MSCL_PP_SKIP;
  const long Mp = (long)g.N * g.Tr * g.Hr * (g.Wr + 2);
  if (Mp >= (1L << 30)) return MSCL_PP_SKIP;
  g.M = (int)Mp;
  g.dW = make_fastdiv(g.Wr + 2);
  g.dKH = make_fastdiv(g.kH);
  g.mtiles = (int)((Mp + OUT_ROWS - 1) / OUT_ROWS);
  g.ntiles = g.Cr / BN;
  const long blocks = (long)g.mtiles * g.ntiles;
  const int ng = g.kT * g.kH * (g.Cs / 64);
  g.ksplit = 1;
  const long out_elems = (long)g.N * g.Tr * g.Hr * g.Wr * g.Cr;
  if (ws != nullptr && g.Cr <= 512 && ilog2_exact(g.Cr / 8) >= 0 && blocks <= slots / 2 && ng >= 4) {
    // too few tiles for the chip: split the (kt, kh, channel part) groups over the grid, one round of blocks, >= 2 groups each
    long want = slots / blocks;
    if (want > ng / 2) want = ng / 2;
    if (want > 16) want = 16;
    if (want * out_elems > ws_floats) want = ws_floats / out_elems;
    if (want > 1) g.ksplit = (int)want;
  }
  static MsclTune t_ks("MSCL_PP_KSPLIT");               // tuning aid
  if (t_ks.read()) {
    const long want = t_ks.val;
    if (want >= 1 && want <= ng && (want == 1 || (ws != nullptr && want * out_elems <= ws_floats && g.Cr <= 512))) g.ksplit = (int)want;
  }
  float* partial = g.ksplit > 1 ? ws : nullptr;
  const size_t lds = 4 * (size_t)BN * 128 + 2 * 256 * 128;
  const int which = BN == 64 ? 1 : 0;
  static bool attr_done[2] = {false, false};    // per kernel (the instantiations have the same function type: ONE lambda body)
  auto go = [&](auto kern) {
    if (!attr_done[which]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done[which] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks * g.ksplit)), dim3(512), lds, st, g, src, wgt, out, bias, addend, ssum, ssq, relu,
                       partial);
  };
  if (which == 1) go(conv_pp_kernel<64>); else go(conv_pp_kernel<128>);
  MSCL_LAUNCH_CHECK();
  ++g_pp_launches;
  if (g.ksplit > 1) return mscl_launch_splitk_finalize(partial, out, bias, addend, relu, ssum, ssq, out_elems / g.Cr, g.Cr, g.ksplit, 0, st);
  return 0;
}
