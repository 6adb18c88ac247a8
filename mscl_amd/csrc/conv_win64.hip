// 3x3 in-plane convolution, 64 -> 64 channels, stride 1, pad 1, kT taps along time (R3D-18 layer 1: 46 % of the step's FLOPs;
// forward and stride-1 input gradient) -- persistent, window-resident, ping-pong.  Reference op: the 3x3x3 convolutions of
// BasicBlock / Conv3DSimple in torchvision's r3d_18 == mmaction/models/backbones/r3d.py:16-34,95-127.
//
// What the previous layer-1 kernel (conv_halo.hip) left on the table, by its own timing builds: a dispatch gap, a window
// prologue and an epilogue per 256-position block (45 % of a block slot at one 144-KB block per CU), one barrier per tap with the
// LDS reads, the DMA issue and the MFMAs of a wave in series, and 12 ds_read_b128 per 16 MFMAs (with 64 output channels every
// wave re-reads the whole 8-KB weight tile: 187 B/clk of LDS reads at full MFMA rate against ~220 delivered).  Here:
//  * PERSISTENT: one 512-thread block per CU walks tiles b, b + grid, ...  The (tile, kt) "groups" form ONE stream: the window of
//    group u+1 and the weight tiles three taps ahead travel by LDS-DMA while group u computes, across tile boundaries; BatchNorm
//    statistics stay in registers until the block ends.
//  * WINDOW: rows are positions in a GLOBAL padded-linear order, G = plane * PL + hp * Wp + wp with Wp = W + 2 (a zero column
//    each side) and PL = (H + 1) * Wp (one zero row between planes), so every in-plane tap is a constant row shift, padding is
//    data (written by the buffer unit's range check) and tiles of 256 consecutive G may straddle planes.  The window
//    [G0 - Wp - 1, G0 + 256 + Wp + 1) of ONE source plane is staged once per (tile, kt) and serves its 9 taps.
//  * PING-PONG: waves 4-7 run one barrier behind waves 0-3; per tap a wave runs [L: 8 ds_read_b128 + its DMA share | barrier |
//    M: 16 MFMAs | barrier], so on every SIMD one wave feeds the matrix pipe while its partner loads (as conv_pp.hip).
//  * K SPLIT BETWEEN SIMD PARTNERS: waves w and w + 4 own the SAME 64 x 64 output tile and half of each tap's 64-deep reduction
//    (k 0..31 / 32..63): 4 + 4 fragment reads per 16 MFMAs instead of 4 + 8.  At the end of a tile the partners exchange half
//    of their partial sums through 32 KB of LDS (w gives rows 32..63, w + 4 gives rows 0..31) and each stores 32 rows.
// Status (round 3): OPT-IN (MSCL_WIN64=1).  Alone: 101 vs 107 us forward, 96 vs 94 us gradient against conv_halo.hip on the
// (8,16,56,56,64) map; inside the three-stream step 970 vs 1007 clip-pairs/s -- 256 long-lived 160-KB blocks leave the other two
// chains nothing to run on, the pattern the earlier persistent variant of conv_halo.hip showed.  Timing builds (compile-time
// switches, since removed): without the DMA issue 77 us, without the fragment reads 80, without the MFMAs 70 -- three comparable
// costs that overlap poorly at 16 MFMAs per barrier pair (an interval lasts ~590 cycles for 256 cycles of MFMA); dropping all
// memory traffic but keeping the DMA instructions saves 11 us, so issue cost, not bandwidth, is what the loads cost.
// Hazard rules as in conv_pp.hip: (R1) a DMA unit waited for in L_p is first read in L_{p+1}; (R2) a slot last read in L_p is
// re-issued in L_{p+1} or later (every wave retires its reads before the barrier that ends its L section).
// DMA schedule, tap counter c (9 per group), weight ring of 4 slots (c & 3), window slots u & 1:
//   L_c issues window piece p of group u+1 (p = c % 9 < 6), then weight tile c + 3; waits until weight tile c + 1 has landed:
//   vmcnt = instructions issued after it = 3, 4, 4, 4, 4, 4, 3, 2, 2 for p = 0..8 (2, 2, 2, 2, 2, 2, 1, 0, - in the last group).
#include "igemm.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> struct WIC { static constexpr int value = N; };

#define WN_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define WN_DSW(addr, src, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(src), "i"(OFF) : "memory")
#define WN_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

struct Win64Geom {
  int NT, T, H, W, Wp, PL, HW;     // planes (N * T), frames per clip, plane size, padded row length, padded plane length
  int kT, pT, mode;                // taps along time and their padding; 0 forward, 1 stride-1 input gradient (taps mirrored)
  int ntiles, Mg;                  // 256-row tiles over Mg = NT * PL padded-linear rows
  int KG;                          // 16-byte granules per weight row (taps * 8)
  FastDiv dPL, dWp, dT;
};

constexpr int WN_NST = 4;          // output store instructions per wave at the end of a tile
constexpr int WN_BM = 256, WN_ROWS = 384, WN_WSLOT = WN_ROWS * 128, WN_BSLOT = 64 * 128, WN_NB = 4;
constexpr int WN_B_BASE = 0, WN_W_BASE = WN_NB * WN_BSLOT, WN_X_BASE = WN_W_BASE + 2 * WN_WSLOT, WN_LDS = WN_X_BASE + 32 * 1024;
static_assert(WN_LDS == 160 * 1024, "LDS plan");

__global__ __launch_bounds__(512) void conv_win64_kernel(const Win64Geom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt,
                                                         bf16_t* __restrict__ out, const bf16_t* __restrict__ addend,
                                                         float* __restrict__ stat_sum, float* __restrict__ stat_sq) {
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, pr = wave & 3;        // ping-pong group (= reduction half) and tile-row quarter of this wave
  const int nblk = (int)gridDim.x;
  const int b0 = xcd_remap(blockIdx.x, nblk);      // blocks of one XCD walk neighbouring tiles (shared halo rows in its L2)
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);
  const int Wp = g.Wp, HALO = g.Wp + 1;
  const int kT = g.kT;
  const int my_tiles = (g.ntiles - b0 + nblk - 1) / nblk;
  const int n_groups = my_tiles * kT;
  if (n_groups <= 0) return;

  // ---- DMA addressing ----
  const int rg = tid & 7, rr = tid >> 3;
  const int rgl = rg ^ (rr & 7);                   // LDS images are [row][granule ^ (row & 7)] (conflict-free under +-1 / +-Wp row shifts)
  const int plane_bytes = g.HW * 128;
  // time taps travel in the SGPR offset, kept non-negative by moving the descriptor base back
  const int bias_bytes = (mode == 0 ? g.pT : (kT - 1 - g.pT)) * plane_bytes;
  const auto rs_src = make_uniform_rsrc(reinterpret_cast<const unsigned char*>(src) - bias_bytes, 0x7FFFFFFFu);
  const auto rs_wgt = make_uniform_rsrc(wgt, 0x7FFFFFFFu);
  const auto rs_out = make_uniform_rsrc(out, 0x7FFFFFFFu);
  const unsigned wrow_voff = (unsigned)(rr * g.KG * 16 + rgl * 16);          // weight row rr (output channel / input channel for wT)
  unsigned win_voff[6]; int win_mask[6];
  auto setup_rows = [&](int tile) {                // window rows p * 64 + rr of `tile`: source offset and valid time taps
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int j = p * 64 + rr;
      const int G = tile * WN_BM - HALO + j;
      int mask = 0; unsigned voff = 0;
      if (j < WN_BM + 2 * HALO && G >= 0 && G < g.Mg) {
        const int plane = fdiv(G, g.dPL), r = G - plane * g.PL;
        const int hp = fdiv(r, g.dWp), wp = r - hp * Wp;
        if (hp >= 1 && wp >= 1 && wp <= g.W) {
          const int t0 = plane - fdiv(plane, g.dT) * g.T;
          // taps k whose source frame t0 + (k - pT) (forward) / t0 - (k - pT) (gradient) lies inside the clip: a range [lo, hi)
          const int lo = (mode == 0) ? max(0, g.pT - t0) : max(0, t0 + g.pT - g.T + 1);
          const int hi = (mode == 0) ? min(kT, g.T - t0 + g.pT) : min(kT, t0 + g.pT + 1);
          mask = hi > lo ? ((1 << hi) - (1 << lo)) : 0;
          voff = (unsigned)(((plane * g.H + hp - 1) * g.W + wp - 1) * 128 + rgl * 16);
        }
      }
      win_voff[p] = voff; win_mask[p] = mask;
    }
  };
  auto kt_soff = [&](int kt) -> unsigned {         // source plane of time tap kt relative to the moved-back base
    return (unsigned)((mode == 0 ? kt : (kT - 1 - kt)) * plane_bytes);
  };
  auto issue_win = [&](int p, unsigned slot, unsigned soff, int ktbit) {
    const unsigned off = (win_mask[p] & ktbit) ? win_voff[p] : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(smem + WN_W_BASE + slot + p * 8192 + wave * 1024), 16, off, soff, 0, 0);
  };
  auto issue_wgt = [&](int c, int tapw) {          // weight tile of tap counter c (ring slot c & 3); tapw = kt * 9 + kh * 3 + kw
    const unsigned woff = __builtin_amdgcn_readfirstlane((unsigned)(tapw * 128));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(smem + WN_B_BASE + (c & 3) * WN_BSLOT + wave * 1024), 16, wrow_voff, woff, 0, 0);
  };

  // ---- fragment addressing: wave pair pr owns tile rows [64 pr, 64 pr + 64); this wave reduces k in [32 grp, 32 grp + 32) ----
  const int fr = lane & 15, fq = lane >> 4;
  const int gq = grp * 4 + fq;                     // 16-byte granule of this lane's fragments inside a 128-byte row
  const int R0 = HALO + pr * 64 + fr;              // window row of tile row pr * 64 + fr at shift 0
  const int sgn = mode == 0 ? 1 : -1;
  const unsigned b_off = lds_base + WN_B_BASE + (unsigned)(fr * 128 + ((gq ^ (fr & 7)) << 4));
  f32x4_t acc[4][4];                               // [channel fragment j][row fragment i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float st_s[4][4], st_q[4][4];                    // BatchNorm partial sums of this wave's stored rows: [j][channel in quad]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[j][r] = 0.f; st_q[j][r] = 0.f; }

  // ---- prologue: weight tiles 0..2 and the whole first window, all landed behind vmcnt(0) ----
  setup_rows(b0);
  issue_wgt(0, 0); issue_wgt(1, 1); issue_wgt(2, 2);
  {
    const unsigned so = __builtin_amdgcn_readfirstlane(kt_soff(0));
#pragma unroll
    for (int p = 0; p < 6; ++p) issue_win(p, 0, so, 1);
  }
  WN_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();      // the stagger

  u32x4_t fa[4], fb[4];
  int tile = b0, kt = 0;
  for (int u = 0; u < n_groups; ++u) {
    const bool has_next = u + 1 < n_groups;
    const bool last_kt = kt == kT - 1;
    const int kt_n = last_kt ? 0 : kt + 1;
    const int tile_n = last_kt ? tile + nblk : tile;
    if (has_next && last_kt) setup_rows(tile_n);   // this tile's windows are all issued: the row state now serves the next tile
    const unsigned so_n = __builtin_amdgcn_readfirstlane(kt_soff(kt_n));
    const int ktbit_n = 1 << kt_n;
    const unsigned wslot = (unsigned)(u & 1) * WN_WSLOT, wslot_n = wslot ^ (unsigned)WN_WSLOT;
    const int c0 = u * 9;                          // tap counter of this group's first tap (9 = 1 mod 4: ring slot (u + p) & 3)
    const bool after_tile = u > 0 && kt == 0;      // the previous group ended a tile: its WN_NST output stores are in the vmcnt queue,
                                                   // younger than the weight tiles phases 0 and 1 wait for

    auto phase = [&](auto PC) {
      constexpr int P = decltype(PC)::value;
      constexpr int kh = P / 3, kw = P % 3;
      // ---- L ----
      const int row = R0 + sgn * ((kh - 1) * Wp + (kw - 1));
      const unsigned aa = lds_base + WN_W_BASE + wslot + (unsigned)(row * 128 + ((gq ^ (row & 7)) << 4));
      const unsigned ba = b_off + (unsigned)((c0 + P) & 3) * WN_BSLOT;
      auto reads = [&]() {
        WN_DSR(fa[0], aa, 0); WN_DSR(fa[1], aa, 2048); WN_DSR(fa[2], aa, 4096); WN_DSR(fa[3], aa, 6144);
        WN_DSR(fb[0], ba, 0); WN_DSR(fb[1], ba, 2048); WN_DSR(fb[2], ba, 4096); WN_DSR(fb[3], ba, 6144);
      };
      reads();
      if (has_next) {
        if constexpr (P < 6) issue_win(P, wslot_n, so_n, ktbit_n);
        if constexpr (P + 3 <= 8) issue_wgt(c0 + P + 3, kt * 9 + P + 3); else issue_wgt(c0 + P + 3, kt_n * 9 + P - 6);
        if constexpr (P == 0) { if (after_tile) WN_VMCNT(3 + WN_NST); else WN_VMCNT(3); }
        else if constexpr (P == 1) { if (after_tile) WN_VMCNT(4 + WN_NST); else WN_VMCNT(4); }
        else if constexpr (P == 6) WN_VMCNT(3);
        else if constexpr (P <= 5) WN_VMCNT(4);
        else WN_VMCNT(2);
      } else {
        if constexpr (P + 3 <= 8) issue_wgt(c0 + P + 3, kt * 9 + P + 3);
        if constexpr (P <= 1) { if (after_tile) WN_VMCNT(2 + WN_NST); else WN_VMCNT(2); }
        else if constexpr (P <= 5) WN_VMCNT(2);
        else if constexpr (P == 6) WN_VMCNT(1);
        else if constexpr (P == 7) WN_VMCNT(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M: 64 positions x 64 channels x 32 deep ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[j]), __builtin_bit_cast(bf16x8_t, fa[i]),
                                                              acc[j][i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    };
    phase(WIC<0>{}); phase(WIC<1>{}); phase(WIC<2>{}); phase(WIC<3>{}); phase(WIC<4>{});
    phase(WIC<5>{}); phase(WIC<6>{}); phase(WIC<7>{}); phase(WIC<8>{});

    if (last_kt) {
      // ---- end of a tile: the partners' partial sums meet.  Wave w (group 0) gives row fragments 2, 3 and keeps 0, 1; wave w + 4
      // gives 0, 1 and keeps 2, 3.  One 8-KB exchange area per pair, used in turn.  Intervals (one barrier apart), group 1 being
      // one interval behind:  g0 write | g1 read + add + write | g0 read + add + store, g1 store | (g1 idles one interval) ----
      const unsigned xa = lds_base + WN_X_BASE + (unsigned)(pr * 8192 + lane * 16);
      auto xwrite = [&](int ibase) {
        WN_DSW(xa, acc[0][ibase], 0); WN_DSW(xa, acc[0][ibase + 1], 1024); WN_DSW(xa, acc[1][ibase], 2048); WN_DSW(xa, acc[1][ibase + 1], 3072);
        WN_DSW(xa, acc[2][ibase], 4096); WN_DSW(xa, acc[2][ibase + 1], 5120); WN_DSW(xa, acc[3][ibase], 6144); WN_DSW(xa, acc[3][ibase + 1], 7168);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      auto xread_add = [&](int ibase) {
        f32x4_t t[8];
        WN_DSR(t[0], xa, 0); WN_DSR(t[1], xa, 1024); WN_DSR(t[2], xa, 2048); WN_DSR(t[3], xa, 3072);
        WN_DSR(t[4], xa, 4096); WN_DSR(t[5], xa, 5120); WN_DSR(t[6], xa, 6144); WN_DSR(t[7], xa, 7168);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j][ibase] += t[2 * j]; acc[j][ibase + 1] += t[2 * j + 1]; }
      };
      // rows this wave stores: fragments IB, IB + 1 of its pair's 64 rows (a compile-time index: a runtime one would put the
      // accumulators in scratch memory)
      auto store_rows = [&](auto IBC) {
        constexpr int IB = decltype(IBC)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = IB + ii;
          const int G = tile * WN_BM + pr * 64 + i * 16 + fr;
          long o = -1;
          if (G < g.Mg) {
            const int plane = fdiv(G, g.dPL), r = G - plane * g.PL;
            const int hp = fdiv(r, g.dWp), wp = r - hp * Wp;
            if (hp >= 1 && wp >= 1 && wp <= g.W) o = ((long)(plane * g.H + hp - 1) * g.W + wp - 1) * 64;
          }
          const bool ok = o >= 0;
          uint2 pk[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (ok) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { st_s[j][r] += v[r]; st_q[j][r] += v[r] * v[r]; }   // statistics of the raw fp32 result
              if (addend != nullptr) {
                const uint2 av = *reinterpret_cast<const uint2*>(addend + o + j * 16 + fq * 4);
                v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
                v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
              }
            }
            pk[j].x = pack2bf(v[0], v[1]); pk[j].y = pack2bf(v[2], v[3]);
            acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          }
          // pair channel fragments (j, j + 1) between lane rows fq, fq ^ 1: one 16-byte store per lane (igemm_epilogue_rows)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const auto sx = __builtin_amdgcn_permlane16_swap(pk[j].x, pk[j + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(pk[j].y, pk[j + 1].y, false, false);
            const u32x4_t w = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            const int n = (j + (fq & 1)) * 16 + (fq & 2) * 4;
            // a buffer store with an out-of-range offset for rows that are not stored: EXACTLY four store instructions per wave
            // and tile, whatever the rows -- the counted vmcnt waits of the next two phases allow for them (WN_NST)
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, ok ? (int)((o + n) * 2) : (int)OOB, 0, 0);
          }
        }
      };
      if (grp == 0) {
        xwrite(2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        xread_add(0);
        store_rows(WIC<0>{});
        __builtin_amdgcn_s_barrier();
      } else {
        xread_add(2);
        xwrite(0);
        __builtin_amdgcn_s_barrier();
        store_rows(WIC<2>{});
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
      // the other half of each accumulator set was given away: start the next tile from zero
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    kt = kt_n; tile = tile_n;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // pairs with the last barrier of waves 4-7

  // ---- BatchNorm statistics: one reduction per block ----
  if (stat_sum != nullptr) {
    WN_VMCNT(0);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [2][64]
    if (tid < 128) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(st_s[j][r]), q = row16_sum(st_q[j][r]);
        if (fr == 0) { atomicAdd(&red[j * 16 + fq * 4 + r], s); atomicAdd(&red[64 + j * 16 + fq * 4 + r], q); }
      }
    __syncthreads();
    if (tid < 64) {
      const int so = (int)(blockIdx.x % MSCL_STAT_ACTIVE) * 2 * 64;
      atomicAdd(&stat_sum[so + tid], red[tid]); atomicAdd(&stat_sq[so + tid], red[64 + tid]);
    }
  }
}

static long g_win64_launches = 0;
extern "C" int64_t mscl_debug_win64_launches(void) { return g_win64_launches; }

// Returns 1 if launched, 0 if the shape is not covered (nothing written), <0 / >0 on error (the convention of mscl_conv_halo64).
extern "C" int mscl_conv_win64(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                               const uint16_t* addend, float* ssum, float* ssq, void* stream) {
  if (!d || !src || !w || !out) return MSCL_E_ARG;
  if ((ssum == nullptr) != (ssq == nullptr)) return MSCL_E_ARG;
  if (d->C != 64 || d->K != 64 || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pH != 1 || d->pW != 1) return 0;
  if (d->kT < 1 || d->kT > 3 || d->pT != (d->kT - 1) / 2 || d->To != d->T) return 0;
  Win64Geom g{};
  g.NT = d->N * d->T; g.T = d->T; g.H = d->H; g.W = d->W; g.Wp = d->W + 2; g.PL = (d->H + 1) * g.Wp; g.HW = d->H * d->W;
  g.kT = d->kT; g.pT = d->pT; g.mode = mode;
  if (WN_BM + 2 * (g.Wp + 1) > WN_ROWS) return 0;                             // W <= 61
  if ((long)(g.NT + 2) * g.HW * 128 >= (1L << 31) || (long)g.NT * g.PL >= (1L << 30)) return 0;
  g.Mg = g.NT * g.PL;
  g.ntiles = (g.Mg + WN_BM - 1) / WN_BM;
  g.KG = d->kT * 9 * 8;
  g.dPL = make_fastdiv(g.PL); g.dWp = make_fastdiv(g.Wp); g.dT = make_fastdiv(d->T);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  const int grid = g.ntiles < cus ? g.ntiles : cus;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_win64_kernel, dim3((unsigned)grid), dim3(512), WN_LDS, (hipStream_t)stream, g, src, w, out, addend, ssum, ssq);
  MSCL_LAUNCH_CHECK();
  ++g_win64_launches;
  return 1;
}
