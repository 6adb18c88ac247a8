// Halo-resident 3x3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (R3D-18 layer 1: 44 % of the
// trunk's FLOPs; forward AND stride-1 input gradient) on bf16 MFMA, gfx950.
//
// Why a second conv kernel: the implicit-GEMM kernels (conv_igemm.hip) re-stage every input row once per tap, 27x.
// With only 64 output channels a staged byte feeds 64 FLOP; a CU moves global memory into LDS at <= 64 B/clk, the
// MFMA pipe wants 4096 FLOP/clk, so re-staging caps those kernels near 40 % of MFMA peak (measured 23 %).
//
// Here a block owns BM = 256 consecutive positions of ONE (n, t) plane in PADDED-LINEAR order: the plane is walked
// as H rows of W + 2 columns (one zero column each side), q = hp * (W + 2) + wp.  In that order every tap is a
// constant shift of q -- (kh - 1) * (W + 2) + (kw - 1) -- and the zero padding is part of the data, so there is no
// per-tap masking at all.  The block stages, ONCE, the window [q0 - (W+2) - 1, q0 + BM + (W+2) + 1) of each of the
// three source planes (384 rows of 128 B each = 144 KB of LDS); rows that fall on padding or outside the clip are
// zero-filled by the buffer unit's out-of-range rule.  The 2 pad columns cost 3.4 % extra MFMA work at W = 56.
//  * weights (8 KB per tap) stream through a 2-stage LDS ring by LDS-DMA; one barrier per tap, placed between the
//    tap's two 32-deep k steps;
//  * right after that barrier a wave reads BOTH operands of the NEXT tap into registers (A from the resident
//    window, B from the ring stage that just landed) and only then issues the DMA for the tap after next, so MFMAs
//    never wait on LDS or on memory in steady state;
//  * planes 2 and 3 of the window stream in under the first taps (2 pieces per tap), ordered before the weight
//    pieces so that the per-tap `vmcnt(0)` never waits for anything issued less than a full tap ago;
//  * 4 waves, 64 positions x 64 channels each: 16 ds_read_b128 per 32 MFMAs (half the LDS read rate);
//  * epilogue as in conv_igemm.hip: BatchNorm sum / sum-of-squares, optional addend, bf16, 8-byte stores.
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
constexpr int NPASS = NH / 32;     // DMA pieces per thread per plane (a pass = 256 threads x 16 B = 32 rows)
constexpr int PLANE_BYTES = NH * 128;
constexpr unsigned HOOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// NW = waves per block.  4: one wave per SIMD, 64 positions x 64 channels each.  8: two waves per SIMD, 32 positions each,
// so one wave's barrier / LDS waits are covered by its neighbour's MFMAs (at 12 instead of 8 LDS reads per 16 MFMAs).
// BM = positions per block, RING = weight-ring stages.  <4, 256, 4>: one 128-KB block per CU.  <4, 128, 2>: 80 KB, TWO
// blocks per CU -- the second block computes through the first one's dispatch gap, window prologue, barriers and epilogue
// (those cost 45 % of a block slot at one block per CU), at the price of twice the weight traffic per position.
template <int NW, int BM, int RING>
__global__ __launch_bounds__(64 * NW, (BM <= 128 ? 2 : 1)) void conv_halo64_kernel(const HaloGeom g, const bf16_t* __restrict__ src,
                                                             const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
                                                             const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                                             float* __restrict__ stat_sq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // The three source planes are visited one after the other, so TWO window slots suffice: the third plane streams into
  // the first one's slot while the second is in use.  The freed LDS holds a deeper weight ring (tiles issued RING taps
  // before use) or, with 128-position tiles, a second block per CU.
  constexpr int RPP = 8 * NW;                              // window rows per DMA pass (64 * NW threads x 16 B)
  constexpr int NHK = (BM + 128 + RPP - 1) / RPP * RPP;    // window rows per plane (BM + 2 * (W + 2) + 2, W <= 61)
  constexpr int PLANE = NHK * 128;
  constexpr int NPS = NHK / RPP;                           // DMA pieces per thread per plane
  constexpr int PPT1 = (NPS + 4) / 5;                      // plane 2 streams in under taps 0..4 (landed before tap 9's operands are read)
  constexpr int PPT2 = (NPS + 5) / 6;                      // plane 3 under taps 8..13, into plane 1's slot
  constexpr int WPOS = BM / NW, IM = WPOS / 16;            // positions per wave, position tiles per wave
  constexpr int WP = NW == 4 ? 2 : 1;                      // DMA instructions per weight tile per wave
  static_assert((NW == 4 || NW == 8) && (RING == 2 || RING == 4) && WPOS % 16 == 0, "configuration");
  unsigned char* const Hs = smem;                          // [2][NHK][128 B] input window slots, row j <-> q0 - Wp - 1 + j
  unsigned char* const Ws = smem + 2 * PLANE;              // [RING][64][128 B] weight ring
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = bid % g.tiles, plane = bid / g.tiles;   // plane = n*T + t
  const int t = plane % g.T;
  const int q0 = g.Wp + tile * BM;                         // first padded-linear position of this tile (hp = 1, wp = 0)
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);

#ifdef HALO_PROBE      // timing-only builds: 1 = weight loads dropped by the range check, 2 = window loads dropped, 3 = both
  const auto rs_src = halo_rsrc(src, (HALO_PROBE & 2) ? 0u : 0x7FFFFFFFu);
  const auto rs_wgt = halo_rsrc(wgt, (HALO_PROBE & 1) ? 0u : 0x7FFFFFFFu);
#else
  const auto rs_src = halo_rsrc(src, 0x7FFFFFFFu);
  const auto rs_wgt = halo_rsrc(wgt, 0x7FFFFFFFu);
#endif

  // ---- window DMA: per-pass VGPR offsets inside a source plane (same for the 3 planes; the plane goes in the SGPR) ----
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
  auto plane_soff = [&](int hp) -> unsigned {              // byte offset of source plane t + hp - 1 (HOOB-safe: invalid -> rows zero)
    const int tt = t + hp - 1;
    return (unsigned)tt < (unsigned)g.T ? (unsigned)((plane + hp - 1) * g.HW) * (HC * 2) : 0u;
  };
  auto plane_ok = [&](int hp) -> bool { return (unsigned)(t + hp - 1) < (unsigned)g.T; };
  auto issue_plane_piece = [&](int hp, int ps) {
    const unsigned so = __builtin_amdgcn_readfirstlane(plane_soff(hp));
    const unsigned vo = plane_ok(hp) ? win_voff[ps] : HOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(Hs + (hp == 1 ? PLANE : 0) + (ps * 64 * NW + wave * 64) * 16), 16, vo, so, 0, 0);
  };
  // ---- weights: tap `tap` -> ring stage tap & 3; rows = output channel n (forward) / input channel (gradient, wT) ----
  // LDS-DMA into a 4-stage ring, issued three taps ahead.  (Staging them through registers instead -- 16-byte loads plus
  // ds_write_b128 -- measured 7 % slower.)
#ifndef HALO_EXP
#define HALO_EXP 0       // timing-study switches (wrong results): 1 no barrier, 2 no operand reads, 4 no weight moves, 8 no plane pieces
#endif
  const int w_row = tid >> 3, w_lg = tid & 7;
  const unsigned w_voff0 = (unsigned)((w_row * 27 * HC + (w_lg ^ (w_row & 7)) * 8) * 2);     // source-side swizzle
  const unsigned w_voff1 = w_voff0 + (unsigned)(32 * 27 * HC * 2);
  auto issue_weights = [&](int tap) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(tap * HC * 2));
    unsigned char* dst = Ws + (tap % RING) * (64 * 128) + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(dst), 16, w_voff0, so, 0, 0);
    if constexpr (NW == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(dst + 4096), 16, w_voff1, so, 0, 0);
  };

  // window plane visited by the kt-th group of taps, and the row shift of tap (kh, kw)
  auto tap_plane = [&](int kt) { return mode ? 2 - kt : kt; };
  auto tap_shift = [&](int kh, int kw) { return mode ? (2 - kh) * g.Wp + (2 - kw) : kh * g.Wp + kw; };

  // ---- prologue: first plane, weights of taps 0 and 1 ----
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) issue_plane_piece(tap_plane(0), ps);
#pragma unroll
  for (int w0 = 0; w0 < RING; ++w0) issue_weights(w0);

  const int fr = lane & 15, fq = lane >> 4;
  const int arow0 = wave * WPOS + fr;                      // window row of fragment 0 at shift 0
  const int brow = fr;                                     // weight row of fragment 0 (j adds 16 rows = 2 KB)
  const int b_addr0 = brow * 128 + ((0 + fq) ^ (brow & 7)) * 16;
  const int b_addr1 = brow * 128 + ((4 + fq) ^ (brow & 7)) * 16;

  bf16x8_t fa[2][2][IM], fb[2][2][4];                      // [buffer][ks][fragment]
  auto read_operands = [&](int tap, int buf) {
    const int kt = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
    const unsigned char* wb = Ws + (tap % RING) * (64 * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[buf][0][j] = *reinterpret_cast<const bf16x8_t*>(wb + b_addr0 + j * 2048);
      fb[buf][1][j] = *reinterpret_cast<const bf16x8_t*>(wb + b_addr1 + j * 2048);
    }
    const int row = arow0 + tap_shift(kh, kw);
    const int key = row & 7;
    const unsigned char* hb = Hs + (kt == 1 ? PLANE : 0) + row * 128;
    const int g0 = (fq ^ key) * 16, g1 = ((4 + fq) ^ key) * 16;
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      fa[buf][0][i] = *reinterpret_cast<const bf16x8_t*>(hb + g0 + i * 2048);
      fa[buf][1][i] = *reinterpret_cast<const bf16x8_t*>(hb + g1 + i * 2048);
    }
  };

  f32x4_t acc[4][IM];                                      // [j: channel tile][i: position tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // plane + tap-0 weights landed; the other RING - 1 weight tiles stay in flight
  if constexpr (WP * (RING - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (WP * (RING - 1) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (WP * (RING - 1) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_operands(0, 0);
  if (HALO_EXP & 2) read_operands(0, 1);

#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int cur = tap & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][0][j], fa[cur][0][i], acc[j][i], 0, 0, 0);
    if (tap + 1 < 27) {
      // Wait for the weight tile of tap + 1, then barrier: it publishes that tile and says every wave holds its tap-`tap`
      // fragments in registers, so their ring stage may be refilled with tap + RING.  (Raw s_barrier: __syncthreads()
      // would drain vmcnt as well.)  Everything issued AFTER the awaited tile may stay in flight -- later weight tiles
      // and the window pieces of the last RING - 1 taps; the count is a compile-time constant per tap (fully unrolled
      // loop).  The piece schedule ends three taps before a plane's first use, so in-order retirement has it landed.
      {
        auto pieces_at = [](int tp) {
          if (tp >= 0 && tp < 5) { const int lo = PPT1 * tp, hi = PPT1 * (tp + 1) < NPS ? PPT1 * (tp + 1) : NPS; return hi > lo ? hi - lo : 0; }
          if (tp >= 8 && tp < 14) { const int lo = PPT2 * (tp - 8), hi = PPT2 * (tp - 7) < NPS ? PPT2 * (tp - 7) : NPS; return hi > lo ? hi - lo : 0; }
          return 0;
        };
        int younger = 0;
        {
          const int wy = RING - 2 < 25 - tap ? RING - 2 : 25 - tap;         // weight tiles tap+2 .. issued so far
          younger = WP * (wy > 0 ? wy : 0);
          for (int sp = (tap + 1 - RING > 0 ? tap + 1 - RING : 0); sp <= tap - 1; ++sp) younger += pieces_at(sp);
        }
        switch (younger) {
#define HW_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
          HW_CASE(0) HW_CASE(1) HW_CASE(2) HW_CASE(3) HW_CASE(4) HW_CASE(5) HW_CASE(6) HW_CASE(7) HW_CASE(8) HW_CASE(9) HW_CASE(10)
          HW_CASE(11) HW_CASE(12) HW_CASE(13) HW_CASE(14)
#undef HW_CASE
          default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
      }
      if (!(HALO_EXP & 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(HALO_EXP & 2)) read_operands(tap + 1, cur ^ 1);
      // weight tile of tap + RING first, then this tap's share of window planes 2 and 3 (in-order retirement: see the wait)
      if (!(HALO_EXP & 4)) {
        if (tap + RING < 27) issue_weights(tap + RING);
      }
      if (!(HALO_EXP & 8)) {
        if (tap < 5) {
#pragma unroll
          for (int u = 0; u < PPT1; ++u) if (PPT1 * tap + u < NPS) issue_plane_piece(tap_plane(1), PPT1 * tap + u);
        } else if (tap >= 8 && tap < 14) {
#pragma unroll
          for (int u = 0; u < PPT2; ++u) if (PPT2 * (tap - 8) + u < NPS) issue_plane_piece(tap_plane(2), PPT2 * (tap - 8) + u);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the next tap's operand reads ABOVE this tap's second k step
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][1][j], fa[cur][1][i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                  // the epilogue reuses the window memory

  // ---- output rows of this lane: padded-linear q -> (hp, wp); pad columns and rows past the plane are dropped ----
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
  // ---- epilogue: BatchNorm statistics ----
  // Every wave plain-stores its 64-channel sums as one LDS row ([wave][2][64], eight 16-byte writes per wave); 128 threads add the
  // rows (no LDS atomics).  This section costs the forward 12-13 us on the layer-1 map (108 vs 95 us for the same kernel as input
  // gradient; ~1.7 us per block, and a CU takes its 6-7 blocks one after the other).  Measured NOT to be the cause: the global float
  // atomics (per-block rows + a fold launch: the same), the LDS atomics this form replaced (the same), slot count 1 .. 1024, slot
  // stride, placement, statistics after the output stores.  What is left is its own arithmetic at the end of every block: 64 FMAs
  // and 32 sixteen-lane reductions (128 dependent DPP adds) per wave between two barriers.
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [NW][2][64]
    const int wv = tid >> 6;
    __syncthreads();                                  // the tiles are dead
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[j][i][r]; s[r] += v; q[r] += v * v; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]);
      }
      if (fr == 0) {
        *reinterpret_cast<float4*>(&red[(wv * 2 + 0) * HC + j * 16 + fq * 4]) = make_float4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<float4*>(&red[(wv * 2 + 1) * HC + j * 16 + fq * 4]) = make_float4(q[0], q[1], q[2], q[3]);
      }
    }
    __syncthreads();
    if (tid < 2 * HC) {
      float t = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) t += red[w8 * 2 * HC + tid];
      const int so = (int)(blockIdx.x % MSCL_STAT_ACTIVE) * g.stat_stride;
      atomicAdd(tid < HC ? &stat_sum[so + tid] : &stat_sq[so + tid - HC], t);
    }
  }
  const bool plain_add = addend != nullptr;
  uint2 add4[4][IM];                                   // all of the lane's addend loads in flight at once (see the fused pass)
  if (plain_add) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < IM; ++i)
        add4[j][i] = *reinterpret_cast<const uint2*>(addend + (orow[i] < 0 ? 0 : orow[i]) + j * 16 + fq * 4);
  }
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    if (orow[i] < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = j * 16 + fq * 4;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if (plain_add) {
        const uint2 av = add4[j][i];
        v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
        v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
      }
      uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(out + orow[i] + n) = pv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same convolution with TWO blocks per CU.  The kernel above keeps one 128-KB block per CU, and what that block
// does besides its 27 taps -- dispatch gap, the exposed window prologue (48 KB before the first MFMA), the statistics epilogue and
// the output stores -- was 43 % of a block slot by the in-kernel stamps of round 1; persistent forms removed it and lost inside the
// three-stream step three times.  Here a block is small enough for a second one beside it (80 KB of LDS, <= 128 registers per
// lane), so one block's prologue / epilogue / plane switches run under the other's taps, with the same weight traffic per
// position as the one-block form (the 128-position two-block tile of round 2 doubled it):
//  * ONE window slot: the three source planes are staged one after the other into the same 48 KB; a plane switch (taps 8 -> 9,
//    17 -> 18) is [all waves done with the old plane | 6 DMA pieces per thread | second k step of the old tap | landed | barrier];
//  * half-tap operand pipeline: the fragments of (tap, k step 1) are read while (tap, k step 0) multiplies and those of
//    (tap + 1, k step 0) while (tap, k step 1) multiplies -- two sets of 6 fragments instead of two sets of 12: 48 operand
//    registers instead of 96, which is what brings the kernel under 128 registers;
//  * one barrier per tap as before (publishes the next tap's weight tile, frees this tap's ring stage), RING stages of 8 KB.
// NW = 8: 32 positions x 64 channels per wave, 12 fragment reads per 16 MFMAs -- with two blocks on the CU the LDS then moves 110 KB per
// tap and block against 512 MFMA-clocks: 84 % as busy as the matrix pipes.  NW = 4: 64 x 64 per wave, 16 reads per 32 MFMAs, 75 KB.
template <int NW, int RING, int TPS = 1>
__global__ __launch_bounds__(64 * NW, NW / 2) void conv_halo64b_kernel(const HaloGeom g, const bf16_t* __restrict__ src,
                                                              const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                                              float* __restrict__ stat_sq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 256;
  constexpr int RPP = 8 * NW;                              // window rows per DMA pass
  constexpr int NHK = (BM + 128 + RPP - 1) / RPP * RPP;    // 384 window rows (BM + 2 * (W + 2) + 2, W <= 61)
  constexpr int PLANE = NHK * 128;
  constexpr int NPS = NHK / RPP;                           // 6 / 12 DMA pieces per thread per plane
  constexpr int WPOS = BM / NW, IM = WPOS / 16;            // 32 / 64 positions per wave, 2 / 4 position tiles
  constexpr int WP = 8 / NW;                               // DMA instructions per weight tile per wave
  static_assert(RING >= 2 && RING <= 4 && (NW == 8 || NW == 4) && (TPS == 1 || TPS == 2) && RING * TPS <= 4, "configuration");
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
  auto issue_plane = [&](int hp) {                         // source plane t + hp - 1 into the window slot (rows outside the clip: zeros)
    const bool okp = (unsigned)(t + hp - 1) < (unsigned)g.T;
    const unsigned so = __builtin_amdgcn_readfirstlane(okp ? (unsigned)((plane + hp - 1) * g.HW) * (HC * 2) : 0u);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const unsigned vo = okp ? win_voff[ps] : HOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(Hs + (ps * 64 * NW + wave * 64) * 16), 16, vo, so, 0, 0);
    }
  };
  const int w_row = tid >> 3, w_lg = tid & 7;
  const unsigned w_voff0 = (unsigned)((w_row * 27 * HC + (w_lg ^ (w_row & 7)) * 8) * 2);     // source-side swizzle
  const unsigned w_voff1 = w_voff0 + (unsigned)(32 * 27 * HC * 2);
  auto issue_weights = [&](int tap) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(tap * HC * 2));
    unsigned char* dst = Ws + (((tap / TPS) % RING) * TPS + tap % TPS) * (64 * 128) + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(dst), 16, w_voff0, so, 0, 0);
    if constexpr (NW == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_ptr_t)(dst + 4096), 16, w_voff1, so, 0, 0);
  };
  auto tap_plane = [&](int kt) { return mode ? 2 - kt : kt; };
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
  if constexpr (TPS == 1) {
    // plane + tap-0 weights landed; the other RING - 1 weight tiles (WP instructions each) stay in flight
    if constexpr (WP * (RING - 1) == 6) HB_WAIT(6); else if constexpr (WP * (RING - 1) == 4) HB_WAIT(4);
    else if constexpr (WP * (RING - 1) == 3) HB_WAIT(3); else if constexpr (WP * (RING - 1) == 2) HB_WAIT(2); else HB_WAIT(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_half(0, 0, 0);

  #pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      read_half(tap, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 27) {
        if ((tap + 1) % 9 == 0) {
          // plane switch: every wave holds the last fragments of the old plane in registers
          HB_WAIT(0);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          issue_plane(tap_plane((tap + 1) / 9));
          if (tap + RING < 27) issue_weights(tap + RING);    // (its ring stage was last read for this tap: free behind the barrier too)
          __builtin_amdgcn_sched_barrier(0);
          mma(1);                                            // the old tap's second k step runs under the DMA
          __builtin_amdgcn_sched_barrier(0);
          HB_WAIT(0);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          read_half(tap + 1, 0, 0);
          continue;
        }
        // wait for the weight tile of tap + 1 (everything issued after it may stay in flight: RING - 2 tiles, fewer at the end),
        // retire this wave's reads of tap's stage; the barrier publishes the one and frees the other
        const int younger = WP * ((RING - 2) < (25 - tap) ? (RING - 2) : (25 - tap > 0 ? 25 - tap : 0));
        if (younger >= 4) HB_WAIT(4); else if (younger == 2) HB_WAIT(2); else if (younger == 1) HB_WAIT(1); else HB_WAIT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tap + RING < 27) issue_weights(tap + RING);
        read_half(tap + 1, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
    }

  } else {
    // ---- a ring stage = TPS taps: one barrier per STAGE (it publishes the next stage and frees the one just used) instead of one per
    // tap -- 17 barriers per block instead of 31; the DMA of a stage has a whole stage of MFMAs to land under ----
    constexpr int NST = (27 + TPS - 1) / TPS;
    auto taps_in = [](int st) { return st < 0 || st >= NST ? 0 : (27 - st * TPS < TPS ? 27 - st * TPS : TPS); };
    auto issue_stage = [&](int st) {
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) if (st < NST && st * TPS + tt < 27) issue_weights(st * TPS + tt);
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
    for (int tap = 0; tap < 27; ++tap) {
      read_half(tap, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 27) {
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
// the time, slower inside the three-stream step, 785-793 vs 804-808 clip-pairs/s, and was removed in round 3.  Its successor is
// a window-resident ping-pong kernel with the K split between SIMD partners (conv_win64.hip, round 3), which showed the same pattern --
// 101 vs 107 us alone, 970 vs 1007 clip-pairs/s in the step -- and was removed in round 4.)

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
  if (d->C != HC || d->K != HC || d->kT != 3 || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 ||
      d->pT != 1 || d->pH != 1 || d->pW != 1) return 0;
  HaloGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.HW = d->H * d->W; g.Wp = d->W + 2;
  if (HBM + 2 * g.Wp + 2 > NH || (long)d->N * d->T * g.HW * HC * 2 >= (1L << 31)) return 0;
  g.tiles = (d->H * g.Wp + HBM - 1) / HBM; g.mode = mode;
  g.dWp = make_fastdiv(g.Wp);
  g.dT = make_fastdiv(d->T); g.dTiles = make_fastdiv(g.tiles);
  g.stat_stride = stat_stride;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64_kernel<8, 256, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  // Two waves per SIMD, 256-position tiles, 4-stage weight ring.  Measured and dropped (all inside the three-stream step, alternating
  // pairs in one call): one wave per SIMD (920-923 vs 927-931 clip-pairs/s), 128-position tiles at two blocks per CU (901-902 vs
  // 903-907), a 2-stage ring that leaves room for another chain's block on the CU (no gain).
  // MSCL_HALO_BLOCKS: 2 (default) = two blocks per CU with one window slot each (conv_halo64b_kernel), 1 = one 128-KB block per CU
  // (conv_halo64_kernel, the round-1..3 form).  Measured in one process (fwd / dgrad of the layer-1 map): 108-111 / 95-96 us -> 92-93 /
  // 81-82 us; step 1078-1082 -> 1097-1101 clip-pairs/s.  MSCL_HALO_RING: weight ring stages of the two-block form (2, 3, 4 measured
  // alike: 93.3 / 92.1 / 92.8 us; 3 = 72 KB per block is the default)
  static MsclTune t_blocks("MSCL_HALO_BLOCKS"), t_ring("MSCL_HALO_RING");
  const unsigned nblk = (unsigned)(d->N * d->T * g.tiles);
  if (t_blocks.get(2) == 2) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr2 = true;
    }
    // (4 waves per block, 64 x 64 per wave -- 16 fragment reads per 32 MFMAs instead of 12 per 16 -- measured like 8: 91.6 vs 95.4 us
    // forward, 83.2 vs 82.1 input gradient, step 1074-1090 vs 1076-1082; the template still takes NW = 4)
    const int ring = t_ring.get(3);
    const size_t lds2 = (size_t)384 * 128 + (size_t)(ring < 2 ? 2 : (ring > 4 ? 4 : ring)) * 64 * 128;
    hipStream_t hs = (hipStream_t)stream;
    // MSCL_HALO_TPS: 2 (default) = a ring of two 2-tap stages (80 KB per block), one barrier per STAGE: 17 barriers per block instead
    // of 31; 1 = one tap per stage with MSCL_HALO_RING stages.  Measured in one process: forward 85.3 -> 83.3 us, input gradient 76.1 ->
    // 75.1; step 1106-1112 -> 1112-1113 clip-pairs/s.
    static MsclTune t_tps("MSCL_HALO_TPS");
    if (t_tps.get(2) == 2) {
      static bool attr3 = false;
      if (!attr3) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64b_kernel<8, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr3 = true; }
      hipLaunchKernelGGL((conv_halo64b_kernel<8, 2, 2>), dim3(nblk), dim3(512), (size_t)384 * 128 + 4 * 64 * 128, hs, g, src, w, out, addend, ssum, ssq);
    } else
    if (ring <= 2) hipLaunchKernelGGL((conv_halo64b_kernel<8, 2>), dim3(nblk), dim3(512), lds2, hs, g, src, w, out, addend, ssum, ssq);
    else if (ring == 3) hipLaunchKernelGGL((conv_halo64b_kernel<8, 3>), dim3(nblk), dim3(512), lds2, hs, g, src, w, out, addend, ssum, ssq);
    else hipLaunchKernelGGL((conv_halo64b_kernel<8, 4>), dim3(nblk), dim3(512), lds2, hs, g, src, w, out, addend, ssum, ssq);
    MSCL_LAUNCH_CHECK();
    ++g_halo_launches;
    return 1;
  }
  const size_t lds = (size_t)2 * 384 * 128 + 4 * 64 * 128;
  hipLaunchKernelGGL((conv_halo64_kernel<8, 256, 4>), dim3(nblk), dim3(512), lds, (hipStream_t)stream, g, src, w,
                     out, addend, ssum, ssq);
  MSCL_LAUNCH_CHECK();
  ++g_halo_launches;
  return 1;
}
