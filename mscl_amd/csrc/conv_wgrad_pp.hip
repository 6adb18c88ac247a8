// Conv3d weight gradient for the 128-and-wider channel layers with kW = 3 (stride 1 along W): shared W taps, ping-pong schedule,
// a three-slot LDS ring filled from the M sections.
//   dW[co][(kt,kh,kw)][ci] += sum_q dy[q][co] * x[src(q; kt, kh) + kw - 1][ci]
// Reference op: the weight gradient autograd computes for the 3x3x3 convolutions of BasicBlock / Conv3DSimple
// (mmaction/models/backbones/r3d.py:16-34,95-127), the SEPC PConv3D convolutions (necks/sepc.py:57-135) and the FPN output
// convolutions (necks/fpn.py:130-152).
//
// conv_wgrad_kernel<128,128,2> stages a 16-KB dy tile and a 16-KB x tile per 64 positions and TAP.  Here, as in conv_pp.hip, the
// reduction index walks positions in PADDED-LINEAR order, q = (n, to, ho) * (Wo + 2) + wp with a zero column each side, so the
// three kw taps of a (kt, kh) pair read the SAME x rows shifted by one: one dy tile and one x tile per 62 positions serve
// 3 x (128 x 128 x 64) products -- 4 DMA instructions per wave per 48 MFMAs.
//  * block = (co tile, ci tile, kt, kh, position split): 8 waves as 2 (co) x 4 (ci), a wave owns 64 co x 32 ci x 3 taps = 96
//    accumulator registers; fragments by ds_read_b64_tr_b16 (both operands are position-major: the reduction index is the LDS
//    row, as in conv_wgrad.hip), the x fragments of tap kw read rows r + kw;
//  * a K tile = 62 positions: 64 dy rows (the last two zero-filled by the range check) against x rows 0 .. 65.  The x slots hold
//    64 rows; rows 64, 65 of a slot are the first two rows of the NEXT slot (two zeroed rows behind the last one): they only ever
//    meet the two zero dy rows, so what they hold does not matter as long as it is finite -- a tile of a non-finite map is
//    non-finite anyway;
//  * ping-pong as conv_pp.hip: phases of one 32-deep k half (24 MFMAs, 20 transposing reads), waves 4-7 one barrier behind
//    waves 0-3;
//  * THREE slots (offsets are immediates of the reads: the loop is unrolled by three; layout [dy0 dy1 dy2 | x0 x1 x2] keeps every
//    immediate below 64 K) and LATE issue: the tile after next travels while this one is multiplied, and its DMA instructions
//    (with their row decode) are issued AFTER the MFMAs of the M sections -- dy in phase 0, x in phase 1 -- i.e. while the wave
//    would otherwise wait at the closing barrier for its SIMD partner's L section.  Round 3's form issued all four pieces in L of
//    phase 0 with two slots; its probes put the L sections (row decode 16 us, DMA issue 22 us, fragment reads 26 us of an
//    82-us launch), not the MFMAs (12 us), in charge of the phase length;
//  * a block stores its 3 x 128 x 128 fp32 partial with plain stores to a slab and wgrad_pp_reduce_kernel adds the slabs of all
//    position splits into dW in split order (fixed order: the same bits every run) -- or, when the layer has enough tiles for
//    ONE split (512 -> 512 on 784 positions: 144 tiles), adds it straight into dW (every element has one owner): no slab
//    round trip, no reduce launch.  Float atomics would move the same bytes at a fifth of the rate (MI355X_MICROARCH.md, Global
//    float atomics).
// Hazard rules as conv_pp.hip (R1: a unit waited for in L_p is first read in L_{p+1}; R2: a slot last read in L_p is re-issued in
// L_{p+1} or later).  Tile t + 2 goes into the slot tile t - 1 left in L of its phase 1 and is issued from M of tile t: later
// than R2 asks.  Tile t + 1 is complete behind the wait of L in phase 1 of tile t (everything but the two dy pieces of tile
// t + 2 issued a section earlier) and is first read in L of phase 0 of tile t + 1.
#include "igemm.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> struct WPC { static constexpr int value = N; };

struct WPGeom {
  int N, T, H, W, C;            // x
  int To, Ho, Wo, K;            // dy
  int kT, kH, sT, sH, pT, pH;   // kW = 3, pW = 1, sW = 1
  int Wp, Mp;                   // Wo + 2, padded-linear positions N * To * Ho * Wp
  int co_tiles, ci_tiles, splits, per_split;    // per_split: padded positions per split, a multiple of 62
  int direct;                   // 1: one split, partials added straight into dW
  FastDiv dWp, dHo, dTo;
};

constexpr int WP_QT = 62;                      // positions per K tile
constexpr int WP_NS = 3;                       // ring slots
constexpr int WP_TILE = 64 * 256;              // bytes of a dy or x tile (64 rows of 128 channels)
constexpr int WP_XBASE = WP_NS * WP_TILE;      // x slots behind the dy slots
constexpr int WP_LDS = 2 * WP_NS * WP_TILE + 2 * 256;      // + two zero rows behind the last x slot
static_assert((WP_NS - 1) * WP_TILE < 65536, "slot offsets are 16-bit immediates of the fragment reads");
constexpr unsigned WP_OOB = 0x80000000u;

// XOR on the 16-byte granule index of a 256-byte row (conv_wgrad.hip wswz<16>)
__device__ __forceinline__ int wp_swz(int row) { return (row & 2) | ((row >> 1) & 4) | ((row & 1) << 3); }

#define WP_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define WP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

// LATE 1: the DMA pieces of tile t + 2 are issued after the MFMAs of the M sections (dy in phase 0, x in phase 1).
// LATE 0: all four in the L section of phase 0 (A/B arm: the round-3 placement on the three-slot ring).
template <int LATE>
__global__ __launch_bounds__(512) void wgrad_pp_kernel(const WPGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       float* __restrict__ slab, float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // ping-pong group
  // blocks that walk the same positions (all tiles of one split) get consecutive logical ids: one XCD's L2 serves their rows
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.co_tiles * g.ci_tiles * g.kT * g.kH;
  const int tile = bid % ntile, split = bid / ntile;
  int tt = tile;
  const int kh = tt % g.kH; tt /= g.kH;
  const int kt = tt % g.kT; tt /= g.kT;
  const int cit = tt % g.ci_tiles, cot = tt / g.ci_tiles;
  const int co0 = cot * 128, ci0 = cit * 128;
  const int mbeg = split * g.per_split;
  const int mend = min(g.Mp, mbeg + g.per_split);
  const int ntiles = (mend > mbeg) ? (mend - mbeg + WP_QT - 1) / WP_QT : 0;

  // the two rows behind the last x slot stay zero (rows 64, 65 of that slot; see the header for the other slots)
  if (tid < 32) *reinterpret_cast<uint4*>(smem + 2 * WP_NS * WP_TILE + tid * 16) = make_uint4(0, 0, 0, 0);
  const auto rs_x = make_uniform_rsrc(x, 0x7FFFFFFFu);
  const auto rs_dy = make_uniform_rsrc(dy, 0x7FFFFFFFu);

  // ---- DMA: a wave moves 4 consecutive rows per pass (row = wave * 4 + lane / 16, granule = lane % 16), two passes per operand
  // tile.  The 4 rows of a wave are consecutive padded positions: their (n, to, ho, wp) decode is done ONCE per wave on the scalar
  // unit for the first of them -- and for the next padded row, which a lane takes when its position wraps past Wp.
  const int lr = lane >> 4, rg = lane & 15;
  const int k2 = g.K * 2, c2 = g.C * 2;
  auto issue_dy = [&](int t, int slot) {
    const int q0 = mbeg + t * WP_QT;
    unsigned char* base = smem + slot * WP_TILE + wave * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p) {                 // dy rows q0 + r
      const int rb = p * 32 + wave * 4;           // first row of this wave in the pass (uniform)
      const int qb = q0 + rb;
      const int nth = fdiv(qb, g.dWp), wpb = qb - nth * g.Wp;      // scalar: qb is wave-uniform
      const int wpl = wpb + lr;
      const bool wrap = wpl >= g.Wp;
      const int wo = (wrap ? wpl - g.Wp : wpl) - 1, nrow = nth + (wrap ? 1 : 0);
      const int r = rb + lr;
      const int lg = rg ^ wp_swz(r);              // logical granule fetched into physical slot rg
      const bool ok = r < WP_QT && qb + lr < mend && (unsigned)wo < (unsigned)g.Wo;
      const unsigned off = ok ? (unsigned)((nrow * g.Wo + wo) * k2 + (co0 + lg * 8) * 2) : WP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(base + p * 8192), 16, off, 0, 0, 0);
    }
  };
  auto issue_x = [&](int t, int slot) {
    const int q0 = mbeg + t * WP_QT;
    unsigned char* base = smem + WP_XBASE + slot * WP_TILE + wave * 1024;
#pragma unroll
    for (int p = 0; p < 2; ++p) {                 // x rows q0 - 1 + r, source plane / row shifted by (kt, kh)
      const int rb = p * 32 + wave * 4;
      const int qb = q0 - 1 + rb;                 // >= -1
      const int qc = qb < 0 ? 0 : qb;
      const int nth = fdiv(qc, g.dWp), wpb = qb - nth * g.Wp;      // (qb = -1: nth 0, wpb -1 -> w = -2 + lr, invalid for lr = 0)
      // source row offset (bytes, w = 0) and validity of padded rows nth and nth + 1, all scalar
      int rowoff[2]; bool rowok[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int nn = nth + e;
        const int nto = fdiv(nn, g.dHo), ho = nn - nto * g.Ho;
        const int n = fdiv(nto, g.dTo), to = nto - n * g.To;
        const int ts = to * g.sT - g.pT + kt, hs = ho * g.sH - g.pH + kh;
        rowok[e] = (unsigned)ts < (unsigned)g.T && (unsigned)hs < (unsigned)g.H && n < g.N;
        rowoff[e] = ((n * g.T + ts) * g.H + hs) * g.W * c2;
      }
      const int wpl = wpb + lr;
      const bool wrap = wpl >= g.Wp;
      const int w = (wrap ? wpl - g.Wp : wpl) - 1;
      const int lg = rg ^ wp_swz(rb + lr);
      const bool ok = (wrap ? rowok[1] : rowok[0]) && (unsigned)w < (unsigned)g.W && qb + lr >= 0 && qb + lr < g.Mp;
      const unsigned off = ok ? (unsigned)((wrap ? rowoff[1] : rowoff[0]) + w * c2 + (ci0 + lg * 8) * 2) : WP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + p * 8192), 16, off, 0, 0, 0);
    }
  };

  // ---- fragment addresses (conv_wgrad.hip): 16 columns x 32 reduction rows = two transposing reads; the slot offset is a
  // compile-time immediate of the read: no address arithmetic in the loop ----
  const int fg = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int wco = (wave & 3) >> 1, wci = ((wave & 3) & 1) | ((wave >> 2) << 1);    // 2 (co) x 4 (ci): partners w, w + 4 differ in ci
  // dy fragment i (16 co), k half ks, read h: row r = ks*32 + 8*fg + 4*h + qq, granule (wco*64)/8 + i*2 + (pp >> 1)
  unsigned a_dy[2][2][4], a_x[2][2][3][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = ks * 32 + 8 * fg + 4 * h + qq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gq = wco * 8 + i * 2 + (pp >> 1);
        a_dy[ks][h][i] = lds_base + (unsigned)(r * 256 + ((gq ^ wp_swz(r)) * 16) + (pp & 1) * 8);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int rx = r + kw;                    // 64, 65: the next slot's first rows (row & 63 keeps the swizzle key a row of its own slot would have)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gq = wci * 4 + j * 2 + (pp >> 1);
          a_x[ks][h][kw][j] = lds_base + WP_XBASE + (unsigned)(rx * 256 + ((gq ^ wp_swz(rx & 63)) * 16) + (pp & 1) * 8);
        }
      }
    }

  f32x4_t acc[3][4][2];                           // [kw][co fragment][ci fragment]
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[kw][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  __syncthreads();                                // the zero rows are written
  if (ntiles > 0) {
    issue_dy(0, 0); issue_x(0, 0);
    if (ntiles > 1) { issue_dy(1, 1); issue_x(1, 1); }
    WP_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger

    s16x4_t va[2][4], vb[2][3][2];
    // one K tile from slot S; tile t + 2 goes into slot (S + 2) % 3
    auto ktile = [&](int t, auto SC) {
      constexpr int S = decltype(SC)::value, S2 = (S + 2) % WP_NS;
      const bool more2 = t + 2 < ntiles;
      auto phase = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        // ---- L ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 4; ++i) WP_TR(va[h][i], a_dy[ks][h][i], S * WP_TILE);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int j = 0; j < 2; ++j) WP_TR(vb[h][kw][j], a_x[ks][h][kw][j], S * WP_TILE);
        }
        if constexpr (LATE == 0 && ks == 0) { if (more2) { issue_dy(t + 2, S2); issue_x(t + 2, S2); } }
        if constexpr (ks == 1) {
          // tile t + 1 complete.  LATE: only the two dy pieces of tile t + 2 (issued in M of phase 0) may stay in flight;
          // LATE 0: all four pieces of tile t + 2 (issued in L of phase 0)
          if (more2) { if constexpr (LATE == 1) WP_VMCNT(2); else WP_VMCNT(4); }
          else WP_VMCNT(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(va[0][0]), "+v"(va[0][1]), "+v"(va[0][2]), "+v"(va[0][3]), "+v"(va[1][0]), "+v"(va[1][1]), "+v"(va[1][2]),
                       "+v"(va[1][3]), "+v"(vb[0][0][0]), "+v"(vb[0][0][1]), "+v"(vb[0][1][0]), "+v"(vb[0][1][1]), "+v"(vb[0][2][0]),
                       "+v"(vb[0][2][1]), "+v"(vb[1][0][0]), "+v"(vb[1][0][1]), "+v"(vb[1][1][0]), "+v"(vb[1][1][1]), "+v"(vb[1][2][0]),
                       "+v"(vb[1][2][1]));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M: 3 taps x (64 co x 32 ci x 32 deep) ----
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        bf16x8_t fa[4], fb[3][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const s16x8_t w8 = {va[0][i][0], va[0][i][1], va[0][i][2], va[0][i][3], va[1][i][0], va[1][i][1], va[1][i][2], va[1][i][3]};
          fa[i] = __builtin_bit_cast(bf16x8_t, w8);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const s16x8_t w8 = {vb[0][kw][j][0], vb[0][kw][j][1], vb[0][kw][j][2], vb[0][kw][j][3],
                                vb[1][kw][j][0], vb[1][kw][j][1], vb[1][kw][j][2], vb[1][kw][j][3]};
            fb[kw][j] = __builtin_bit_cast(bf16x8_t, w8);
          }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[kw][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[kw][j], acc[kw][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LATE == 1) {
          if (more2) { if constexpr (ks == 0) issue_dy(t + 2, S2); else issue_x(t + 2, S2); }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();
      };
      phase(WPC<0>{}); phase(WPC<1>{});
    };
    for (int t = 0; t < ntiles; t += 3) {
      ktile(t, WPC<0>{});
      if (t + 1 < ntiles) ktile(t + 1, WPC<1>{});
      if (t + 2 < ntiles) ktile(t + 2, WPC<2>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // pairs with the last barrier of waves 4-7
  }

  if (g.direct) {
    // ---- one split: every element of this tile has one owner; dW[co][(kt,kh,kw)][ci] += partial (loads of a tap in flight together)
    const int ntaps = g.kT * g.kH * 3;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int tap = (kt * g.kH + kh) * 3 + kw;
      float old[4][2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 64 + i * 16 + (lane >> 4) * 4 + r, ci = ci0 + wci * 32 + j * 16 + (lane & 15);
            old[i][j][r] = dw[((long)co * ntaps + tap) * g.C + ci];
          }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 64 + i * 16 + (lane >> 4) * 4 + r, ci = ci0 + wci * 32 + j * 16 + (lane & 15);
            dw[((long)co * ntaps + tap) * g.C + ci] = old[i][j][r] + acc[kw][i][j][r];
          }
    }
    return;
  }
  // ---- this block's partial: slab[(split * ntile + tile)][kw][co 128][ci 128] fp32, plain stores ----
  float* out = slab + ((long)split * ntile + tile) * (3L * 128 * 128);
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = wco * 64 + i * 16 + (lane >> 4) * 4 + r, ci = wci * 32 + j * 16 + (lane & 15);
          out[(kw * 128 + co) * 128 + ci] = acc[kw][i][j][r];
        }
}

// dW[co][(kt,kh,kw)][ci] += sum over the position splits, in split order.  One block per (tile, kw, 8 co rows).
__global__ __launch_bounds__(256) void wgrad_pp_reduce_kernel(const WPGeom g, const float* __restrict__ slab, float* __restrict__ dw) {
  const int ntile = g.co_tiles * g.ci_tiles * g.kT * g.kH;
  int b = blockIdx.x;
  const int rowblk = b % 16; b /= 16;             // 8 co rows each
  const int kw = b % 3; const int tile = b / 3;
  int tt = tile;
  const int kh = tt % g.kH; tt /= g.kH;
  const int kt = tt % g.kT; tt /= g.kT;
  const int cit = tt % g.ci_tiles, cot = tt / g.ci_tiles;
  const int ntaps = g.kT * g.kH * 3, tap = (kt * g.kH + kh) * 3 + kw;
  const int co_l = rowblk * 8 + (threadIdx.x >> 5), ci4 = (threadIdx.x & 31) * 4;
  const long e = ((long)tile * 3 + kw) * (128L * 128) + (long)co_l * 128 + ci4;
  const long stride = (long)ntile * (3L * 128 * 128);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int sp = 0;
  for (; sp + 4 <= g.splits; sp += 4) {           // four loads in flight, added in split order
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(slab + (long)(sp + u) * stride + e);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; sp < g.splits; ++sp) {
    const float4 v = *reinterpret_cast<const float4*>(slab + (long)sp * stride + e);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = dw + ((long)(cot * 128 + co_l) * ntaps + tap) * g.C + cit * 128 + ci4;
  float4 d = *reinterpret_cast<float4*>(o);
  d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
  *reinterpret_cast<float4*>(o) = d;
}

static long g_wgrad_pp_launches = 0;
extern "C" int64_t mscl_debug_wgrad_pp_launches(void) { return g_wgrad_pp_launches; }

static bool wpp_shape(const mscl_conv_desc* d) {
  return d->kW == 3 && d->pW == 1 && d->sW == 1 && d->Wo == d->W && (d->C % 128) == 0 && (d->K % 128) == 0 && d->kT <= 8 && d->kH <= 8;
}
// position splits of a layer: one round of blocks over the chip, at least 4 K tiles per block
static long wpp_splits(const mscl_conv_desc* d) {
  const long Mp = (long)d->N * d->To * d->Ho * (d->Wo + 2);
  const int ntile = (d->K / 128) * (d->C / 128) * d->kT * d->kH;
  long splits = 256 / ntile; if (splits < 1) splits = 1;
  const long maxs = (Mp + 4 * WP_QT - 1) / (4 * WP_QT);
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  return splits;
}

// floats of workspace mscl_wgrad_pp wants for this layer (0: the layer is not covered, or needs none -- see mscl_wgrad_pp_covers)
extern "C" int64_t mscl_wgrad_pp_ws(const mscl_conv_desc* d) {
  if (!d || !wpp_shape(d)) return 0;
  const long splits = wpp_splits(d);
  if (splits == 1) return 0;                      // straight into dW
  const int ntile = (d->K / 128) * (d->C / 128) * d->kT * d->kH;
  return splits * ntile * (3L * 128 * 128);
}
extern "C" int mscl_wgrad_pp_covers(const mscl_conv_desc* d) { return d && wpp_shape(d) ? 1 : 0; }

// returns 1 if launched (dw updated), 0 if the shape / workspace is not covered (nothing written), <0 / >0 on error
int mscl_wgrad_pp(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats, hipStream_t st) {
  if (!wpp_shape(d)) return 0;
  const int64_t need = mscl_wgrad_pp_ws(d);
  if (need > 0 && (ws == nullptr || ws_floats < need)) return 0;
  if ((long)d->N * d->T * d->H * d->W * d->C * 2 >= (1L << 31) || (long)d->N * d->To * d->Ho * d->Wo * d->K * 2 >= (1L << 31)) return 0;
  WPGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.C = d->C; g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.K = d->K;
  g.kT = d->kT; g.kH = d->kH; g.sT = d->sT; g.sH = d->sH; g.pT = d->pT; g.pH = d->pH;
  g.Wp = d->Wo + 2;
  const long Mp = (long)d->N * d->To * d->Ho * g.Wp;
  if (Mp >= (1L << 30)) return 0;
  g.Mp = (int)Mp;
  g.co_tiles = d->K / 128; g.ci_tiles = d->C / 128;
  const int ntile = g.co_tiles * g.ci_tiles * g.kT * g.kH;
  g.splits = (int)wpp_splits(d);
  g.direct = g.splits == 1 ? 1 : 0;
  const long per = (Mp + g.splits - 1) / g.splits;
  g.per_split = (int)((per + WP_QT - 1) / WP_QT * WP_QT);
  g.dWp = make_fastdiv(g.Wp); g.dHo = make_fastdiv(d->Ho); g.dTo = make_fastdiv(d->To);
  static MsclTune t_late("MSCL_WGRAD_PP_LATE");
  const int late = t_late.get(1) != 0;
  static bool attr_done[2] = {false, false};
  if (!attr_done[late]) {
    const void* k = late ? reinterpret_cast<const void*>(wgrad_pp_kernel<1>) : reinterpret_cast<const void*>(wgrad_pp_kernel<0>);
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[late] = true;
  }
  if (late) hipLaunchKernelGGL(wgrad_pp_kernel<1>, dim3((unsigned)(ntile * g.splits)), dim3(512), WP_LDS, st, g, x, dy, ws, dw);
  else hipLaunchKernelGGL(wgrad_pp_kernel<0>, dim3((unsigned)(ntile * g.splits)), dim3(512), WP_LDS, st, g, x, dy, ws, dw);
  MSCL_LAUNCH_CHECK();
  if (!g.direct) {
    hipLaunchKernelGGL(wgrad_pp_reduce_kernel, dim3((unsigned)(ntile * 3 * 16)), dim3(256), 0, st, g, (const float*)ws, dw);
    MSCL_LAUNCH_CHECK();
  }
  ++g_wgrad_pp_launches;
  return 1;
}
