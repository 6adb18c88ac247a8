// Conv3d weight gradient for the 128-and-wider channel layers with kW = 3 (stride 1 along W): shared W taps, ping-pong schedule.
//   dW[co][(kt,kh,kw)][ci] += sum_q dy[q][co] * x[src(q; kt, kh) + kw - 1][ci]
// Reference op: the weight gradient autograd computes for the 3x3x3 convolutions of BasicBlock / Conv3DSimple
// (mmaction/models/backbones/r3d.py:16-34,95-127), the SEPC PConv3D convolutions (necks/sepc.py:57-135) and the FPN output
// convolutions (necks/fpn.py:130-152).
//
// conv_wgrad_kernel<128,128,2> stages a 16-KB dy tile and a 16-KB x tile per 64 positions and TAP: 8 LDS-DMA instructions per
// wave per 32 MFMAs, two and a half times what the forward kernels issue, and the DMA issue rate is what bounds these loops
// (conv_pp.hip).  Here, as there, the reduction index walks positions in PADDED-LINEAR order, q = (n, to, ho) * (Wo + 2) + wp with
// a zero column each side, so the three kw taps of a (kt, kh) pair read the SAME x rows shifted by one: one dy tile and one x tile
// per 62 positions serve 3 x (128 x 128 x 64) products -- 4 DMA instructions per wave per 48 MFMAs.
//  * block = (co tile, ci tile, kt, kh, position split): 8 waves as 2 (co) x 4 (ci), a wave owns 64 co x 32 ci x 3 taps = 96
//    accumulator registers; fragments by ds_read_b64_tr_b16 (both operands are position-major: the reduction index is the LDS
//    row, as in conv_wgrad.hip), the x fragments of tap kw read rows r + kw;
//  * a K tile = 62 positions: 64 dy rows (the last two zero-filled) against x rows 0 .. 65 (rows 64, 65 of every slot are zeroed
//    once; the DMA fills rows 0 .. 63 = positions q0 - 1 .. q0 + 62);
//  * ping-pong as conv_pp.hip: phases of one 32-deep k half (24 MFMAs, 20 transposing reads), waves 4-7 one barrier behind
//    waves 0-3; two LDS slots (their offsets are immediates of the reads: the loop is unrolled by two), tile t + 1 travels while
//    tile t is multiplied;
//  * every block stores its 3 x 128 x 128 fp32 partial with plain stores to a slab; wgrad_pp_reduce_kernel adds the slabs of all
//    position splits into dW in split order (fixed order: the same bits every run).  Float atomics would move the same bytes at
//    a fifth of the rate (MI355X_MICROARCH.md, Global float atomics).
//
// Status: OPT-IN (MSCL_WGRAD_PP=1 / 2, conv_wgrad.hip), kept for its fixed-order result and as the record of the measurement.
// Layer 2 (128 -> 128, 50176 positions, 27 taps): 82.5 us against 81.0 of conv_wgrad_kernel<128,128,2> with float atomics; layer 3
// 54.5 vs 53.9; layer 4 56 vs 40 (9 tiles x 28 splits of a small map); inside the step 1010 vs 1033 clip-pairs/s.  Where the time
// goes on layer 2 (compile-time probes of the three-slot form of this kernel, 81.8 us whole): the slab reduce 10.5 us; the slab
// stores ~10 (252 blocks x 196 KB, all at the end of the launch); without the row decode 65, without DMA 60, without fragment reads
// 56, without MFMAs 70 -- the L sections, not the MFMAs, set the phase length.  The scalar row decode and immediate slot offsets
// of this form removed the vector work of L and did not move the total: what is left is the L2 -> LDS fill itself.  A tile moves
// 33 KB for 3 x 62 x 128 x 128 MACs = 93 MAC/B; at the ~6.4 TB/s the chip's LDS-DMA reaches (MI355X_MICROARCH.md, ldsdma-fill; the
// forward kernel of conv_pp.hip runs at that rate at 0.39-0.43 of the MFMA peak) that bounds the main loop at 0.48 of peak = 37 us,
// and the 50 MB of slabs (1.77 MB of dW x 28 splits, written once and read once) add ~20 us that no schedule removes: 252 blocks
// need 28 splits of 9 tiles.  More reuse per byte (all nine (kh, kw) taps against one dy tile) needs 288 accumulator registers at
// 128 x 128 or halves the tile and doubles the splits: not pursued.
#include "igemm.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> struct WPC { static constexpr int value = N; };

struct WPGeom {
  int N, T, H, W, C;            // x
  int To, Ho, Wo, K;            // dy
  int kT, kH, sT, sH, pT, pH;   // kW = 3, pW = 1, sW = 1
  int Wp, Mp;                   // Wo + 2, padded-linear positions N * To * Ho * Wp
  int co_tiles, ci_tiles, splits, per_split;    // per_split: padded positions per split, a multiple of 62
  FastDiv dWp, dHo, dTo;
};

constexpr int WP_QT = 62;                      // positions per K tile
constexpr int WP_DY = 64 * 256, WP_X = 66 * 256, WP_SLOT = WP_DY + WP_X, WP_NS = 2;
static_assert(WP_SLOT < 65536, "the second slot's offset is a 16-bit immediate of the fragment reads");
constexpr unsigned WP_OOB = 0x80000000u;

// XOR on the 16-byte granule index of a 256-byte row (conv_wgrad.hip wswz<16>)
__device__ __forceinline__ int wp_swz(int row) { return (row & 2) | ((row >> 1) & 4) | ((row & 1) << 3); }

#define WP_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define WP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

__global__ __launch_bounds__(512) void wgrad_pp_kernel(const WPGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       float* __restrict__ slab) {
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

  // rows 64, 65 of both x slots stay zero: the fragments of tap kw = 2 read them against the two zero dy rows
  if (tid < 64) {
    const int s = tid >> 5, i = tid & 31;
    *reinterpret_cast<uint4*>(smem + s * WP_SLOT + WP_DY + 64 * 256 + i * 16) = make_uint4(0, 0, 0, 0);
  }
  const auto rs_x = make_uniform_rsrc(x, 0x7FFFFFFFu);
  const auto rs_dy = make_uniform_rsrc(dy, 0x7FFFFFFFu);

  // ---- DMA: a wave moves 4 consecutive rows per pass (row = wave * 4 + lane / 16, granule = lane % 16), two passes per operand
  // tile.  The 4 rows of a wave are consecutive padded positions: their (n, to, ho, wp) decode is done ONCE per wave on the scalar
  // unit for the first of them -- and for the next padded row, which a lane takes when its position wraps past Wp -- so a lane
  // spends ~8 vector instructions per row instead of three divisions (the per-lane decode cost 16 us of an 82-us launch).
  const int lr = lane >> 4, rg = lane & 15;
  const int k2 = g.K * 2, c2 = g.C * 2;
  auto issue_tile = [&](int t, int slot) {
    const int q0 = mbeg + t * WP_QT;
    unsigned char* base = smem + slot * WP_SLOT + wave * 1024;
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
      const int r = rb + lr;
      const int lg = rg ^ wp_swz(r);
      const bool ok = (wrap ? rowok[1] : rowok[0]) && (unsigned)w < (unsigned)g.W && qb + lr >= 0 && qb + lr < g.Mp;
      const unsigned off = ok ? (unsigned)((wrap ? rowoff[1] : rowoff[0]) + w * c2 + (ci0 + lg * 8) * 2) : WP_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + WP_DY + p * 8192), 16, off, 0, 0, 0);
    }
  };

  // ---- fragment addresses (conv_wgrad.hip): 16 columns x 32 reduction rows = two transposing reads; the slot offset is a
  // compile-time immediate of the read (two slots, loop unrolled by two): no address arithmetic in the loop ----
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
        const int rx = r + kw;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gq = wci * 4 + j * 2 + (pp >> 1);
          a_x[ks][h][kw][j] = lds_base + WP_DY + (unsigned)(rx * 256 + ((gq ^ wp_swz(rx)) * 16) + (pp & 1) * 8);
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
    issue_tile(0, 0);
    WP_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger

    s16x4_t va[2][4], vb[2][3][2];
    // one K tile from slot S: phase 0 issues the whole next tile into the other slot (last read one tile ago: rule R2) and
    // multiplies k half 0, phase 1 waits for it (rule R1: read from the next tile's phase 0 on) and multiplies k half 1
    auto ktile = [&](int t, auto SC) {
      constexpr int S = decltype(SC)::value;
      const bool more = t + 1 < ntiles;
      auto phase = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        // ---- L ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 4; ++i) WP_TR(va[h][i], a_dy[ks][h][i], S * WP_SLOT);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int j = 0; j < 2; ++j) WP_TR(vb[h][kw][j], a_x[ks][h][kw][j], S * WP_SLOT);
        }
        if constexpr (ks == 0) { if (more) issue_tile(t + 1, S ^ 1); }
        else WP_VMCNT(0);
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
        __builtin_amdgcn_s_barrier();
      };
      phase(WPC<0>{}); phase(WPC<1>{});
    };
    for (int t = 0; t < ntiles; t += 2) {
      ktile(t, WPC<0>{});
      if (t + 1 < ntiles) ktile(t + 1, WPC<1>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // pairs with the last barrier of waves 4-7
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
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < g.splits; ++sp) {
    const float4 v = *reinterpret_cast<const float4*>(slab + (long)sp * ntile * (3L * 128 * 128) + e);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = dw + ((long)(cot * 128 + co_l) * ntaps + tap) * g.C + cit * 128 + ci4;
  float4 d = *reinterpret_cast<float4*>(o);
  d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
  *reinterpret_cast<float4*>(o) = d;
}

static long g_wgrad_pp_launches = 0;
extern "C" int64_t mscl_debug_wgrad_pp_launches(void) { return g_wgrad_pp_launches; }

// floats of workspace mscl_wgrad_pp wants for this layer (0: the layer is not covered)
extern "C" int64_t mscl_wgrad_pp_ws(const mscl_conv_desc* d) {
  if (!d) return 0;
  if (d->kW != 3 || d->pW != 1 || d->sW != 1 || d->Wo != d->W || (d->C % 128) || (d->K % 128) || d->kT > 8 || d->kH > 8) return 0;
  const long Mp = (long)d->N * d->To * d->Ho * (d->Wo + 2);
  const int ntile = (d->K / 128) * (d->C / 128) * d->kT * d->kH;
  long splits = 256 / ntile; if (splits < 1) splits = 1;
  const long maxs = (Mp + 4 * WP_QT - 1) / (4 * WP_QT);        // at least 4 K tiles per block
  if (splits > maxs) splits = maxs;
  return splits * ntile * (3L * 128 * 128);
}

// returns 1 if launched (dw updated), 0 if the shape / workspace is not covered (nothing written), <0 / >0 on error
int mscl_wgrad_pp(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats, hipStream_t st) {
  const int64_t need = mscl_wgrad_pp_ws(d);
  if (need == 0 || ws == nullptr || ws_floats < need) return 0;
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
  g.splits = (int)(need / (ntile * (3L * 128 * 128)));
  const long per = (Mp + g.splits - 1) / g.splits;
  g.per_split = (int)((per + WP_QT - 1) / WP_QT * WP_QT);
  g.dWp = make_fastdiv(g.Wp); g.dHo = make_fastdiv(d->Ho); g.dTo = make_fastdiv(d->To);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(wgrad_pp_kernel, dim3((unsigned)(ntile * g.splits)), dim3(512), WP_NS * WP_SLOT, st, g, x, dy, ws);
  MSCL_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_pp_reduce_kernel, dim3((unsigned)(ntile * 3 * 16)), dim3(256), 0, st, g, (const float*)ws, dw);
  MSCL_LAUNCH_CHECK();
  ++g_wgrad_pp_launches;
  return 1;
}
