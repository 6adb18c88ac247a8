// Input gradient of the 3x3x3 / stride 2 / pad 1 entry convolution of a stage on a LARGE map (R3D-18 layer 2: dy 8 x 8 x 28 x 28 x 128
// -> dx 8 x 16 x 56 x 56 x 64), with the dy window resident in LDS.  Reference op: the input gradient autograd computes for the
// strided Conv3DSimple of BasicBlock (mmaction/models/backbones/r3d.py:16-34, 95-127; torchvision r3d_18 layer2.0.conv1).
//
// Why.  Output position i = 2 m + p of one axis receives tap a from dy position t' with 2 t' - 1 + a = i: p = 0 (even) takes a = 1
// from t' = m; p = 1 (odd) takes a = 2 from t' = m and a = 0 from t' = m + 1.  So a CELL m = (t', h', w') of the dy grid owns the
// 2 x 2 x 2 block of dx positions (2 t' + pt, 2 h' + ph, 2 w' + pw) -- eight parity CLASSES -- and class (pt, ph, pw) sums over the
// dy positions m + (dt, dh, dw) with d <= p componentwise: 1 2 2 4 2 4 4 8 = 27 (class, tap) pairs, each a dense [cells x Cout] x
// [Cout x Cin] product.  The implicit-GEMM kernel (conv_igemm.hip, mode 2) runs the classes as separate tile sets and re-stages the
// dy rows for every pair: at 64 output channels a K step stages 24 KB for 0.5 M MACs, and rounds 3-5 measured every tiling of it
// at 77-92 us = 0.085 of the MFMA peak ("only keeping dy resident across a cell's 27 uses would lift it").  This kernel does that.
//
// A block = 256 cells of one (n, t') plane in PADDED-LINEAR order (rows of W' + 1, the extra column is zeros written by the buffer
// unit's range check, rows past the plane too) x 64 dx channels, 4 waves of 64 x 64, one wave per SIMD.  The window of a plane --
// 256 + W' + 2 rows of one 64-channel chunk of dy, 36 KB -- serves the nine (class, dh, dw) pairs of one dt as constant row shifts
// dh (W' + 1) + dw; the 64 x 64 weight tile of a (tap, chunk) is all that changes between pairs (8 KB through a ring of six
// slots, LDS-DMA, five steps ahead).  Two passes: pt = 0 (4 classes, 9 pairs per chunk, plane t' only) and pt = 1 (4 classes, 18 pairs per chunk,
// planes t' and t' + 1); the four classes of a pass keep their accumulators in registers (4 x 64 x 64 per block = 256 registers
// per lane), so every staged byte is used by every pair that needs it.  A SEGMENT = (chunk, dt) = nine steps on one window
// buffer; the next segment's window is issued into the other buffer at the segment's first step (two buffers ping-pong: dt = 0 / dt = 1
// of a chunk in pass 1, consecutive chunks in pass 0).  A step = [counted vmcnt | barrier | DMA issue for the next segment and for
// the weight tile three steps ahead | fragment reads of the NEXT step | 32 MFMAs of this step]: reads and DMA of one wave run under
// its own MFMAs (one wave per SIMD), one barrier per 512 MFMA cycles.
// Per block 56.6 MMAC = 6912 MFMAs; LDS reads 16 ds_read_b128 per 32 MFMAs; DMA 8 KB of weights + 4 KB of window per step.
// Measured (layer-2 entry, 22.2 GFLOP): 75.7 us on the parity-class kernel -> 55.9 us alone (tools/bench_conv.py); WITH the addend the step
// passes (the shortcut's input gradient: 51 MB more to read) 87.9 -> 56.8 us.  (A form that fetched the addend inside the store loop ran
// 46.7 us without an addend and 86.9 us with one: eight dependent HBM round trips per class.)  With parts switched off (a study build): loop skeleton + prologues + epilogues 27 us, + DMA and fragment reads 36 us,
// + MFMAs 48 us -- the three add up, because ONE block per CU (256 accumulators per lane) leaves nothing to run under a block's
// store-heavy epilogue or its prologue.  On the way: the shared implicit-GEMM epilogue inlined four times beside the accumulators
// spilled 388 bytes per lane to scratch (64.5 -> 48.1 us when the rows went through LDS instead); weight tiles three steps ahead
// instead of five, or the window's pieces spread over the steps, changed nothing.
#include "igemm.h"

typedef __attribute__((ext_vector_type(4))) unsigned s2_u32x4;
typedef __attribute__((address_space(3))) void* s2_lds_t;
template <int N> struct S2C { static constexpr int value = N; };

struct S2Geom {
  int N, Tp, Hp, Wp;        // dy: T', H', W'
  int T, H, W;              // dx
  int Co, Ci;               // dy / dx channels
  int Wq, cells, tiles;     // padded row length W' + 1; padded cells per plane H' Wq; 256-cell tiles per plane
  int nct, nchunk;          // Ci / 64, Co / 64
  FastDiv dWq, dTiles, dTp, dNct;
};

constexpr int S2_CM = 256, S2_AROWS = 288, S2_ABUF = S2_AROWS * 128, S2_WSLOT = 64 * 128, S2_NW = 6;      // S2_NW: weight ring slots
constexpr int S2_A_BASE = S2_NW * S2_WSLOT, S2_LDS = S2_A_BASE + 2 * S2_ABUF;
constexpr unsigned S2_OOB = 0x80000000u;
#define S2_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
#define S2_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

// the nine (class, dh, dw) pairs of a segment; class = 2 ph + pw; in-plane tap 3 a_h + a_w with a = 1 (p = 0), 2 (p = 1, d = 0), 0 (p = 1, d = 1)
__device__ __forceinline__ constexpr int s2_cls(int p) { return p == 0 ? 0 : (p <= 2 ? 1 : (p <= 4 ? 2 : 3)); }
__device__ __forceinline__ constexpr int s2_dh(int p) { return (p == 4 || p == 7 || p == 8) ? 1 : 0; }
__device__ __forceinline__ constexpr int s2_dw(int p) { return (p == 2 || p == 6 || p == 8) ? 1 : 0; }
__device__ __forceinline__ constexpr int s2_tap_hw(int p) {
  const int ph = s2_cls(p) >> 1, pw = s2_cls(p) & 1;
  const int ah = ph == 0 ? 1 : (s2_dh(p) == 0 ? 2 : 0), aw = pw == 0 ? 1 : (s2_dw(p) == 0 ? 2 : 0);
  return ah * 3 + aw;
}
__global__ __launch_bounds__(256, 1) void dgrad_s2_kernel(const S2Geom g, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wT,
                                                          bf16_t* __restrict__ dx, const bf16_t* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(s2_lds_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = bid - fdiv(bid, g.dNct) * g.nct; bid = fdiv(bid, g.dNct);
  const int plane = fdiv(bid, g.dTiles), tile = bid - plane * g.tiles;
  const int n = fdiv(plane, g.dTp), tp = plane - n * g.Tp;
  const int q0 = tile * S2_CM;
  const int co2 = g.Co * 2;
  const auto rs_dy = make_uniform_rsrc(dy, 0x7FFFFFFFu);
  const auto rs_w = make_uniform_rsrc(wT, 0x7FFFFFFFu);

  // ---- per-lane DMA offsets.  A 1-KiB piece = 8 rows of 128 bytes; lane -> row (lane >> 3), physical granule lane & 7, which holds
  // the row's LOGICAL granule (lane & 7) ^ (row & 7) (XOR swizzle on the source side, conflict-free ds_read_b128 under any row shift)
  const int prow = lane >> 3, rgl = (lane & 7) ^ prow;
  unsigned a_voff[9];                        // window piece kk of this wave: rows 8 (4 kk + wave) + prow
#pragma unroll
  for (int kk = 0; kk < 9; ++kk) {
    const int r = 8 * (4 * kk + wave) + prow, q = q0 + r;
    const int hq = fdiv(q, g.dWq), wq = q - hq * g.Wq;
    a_voff[kk] = (q < g.cells && wq < g.Wp) ? (unsigned)((hq * g.Wp + wq) * co2 + rgl * 16) : S2_OOB;
  }
  unsigned w_voff[2];                        // weight piece kk of this wave: rows (dx channels) 8 (4 kk + wave) + prow of the tile
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) w_voff[kk] = (unsigned)((64 * nt + 8 * (4 * kk + wave) + prow) * 27 * co2 + rgl * 16);
  const unsigned plane_bytes = (unsigned)(g.Hp * g.Wp) * (unsigned)co2;

  auto issue_window = [&](int buf, int kk, unsigned soff, bool ok) {       // ok = the plane exists (t' + dt < T'): else zeros
    unsigned char* a = smem + S2_A_BASE + buf * S2_ABUF + (4 * kk + wave) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (s2_lds_t)a, 16, ok ? a_voff[kk] : S2_OOB, soff, 0, 0);
  };
  auto issue_weights = [&](int slot, unsigned soff) {
    unsigned char* b = smem + slot * S2_WSLOT + wave * 1024;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (s2_lds_t)(b + kk * 4096), 16, w_voff[kk], soff, 0, 0);
  };

  // ---- fragment addressing ----
  const int fr = lane & 15, fq = lane >> 4;
  const int wm0 = wave * 64;
  unsigned b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) b_off[ks] = lds_base + (unsigned)(fr * 128 + (((ks * 4 + fq) ^ (fr & 7)) * 16));
  const int arow0 = wm0 + fr;

  // the two passes in either order: all blocks of the launch are resident together and would otherwise reach their store-heavy epilogues
  // (25 MB of dx per pass on the layer-2 entry map) at the same moments; odd blocks run the long pass first
  for (int pi = 0; pi < 2; ++pi) {
    const int pt = (blockIdx.x & 1) ? 1 - pi : pi;
    if (2 * tp + pt >= g.T) continue;         // (odd T: the last plane has no odd output)
    const int nseg = pt == 0 ? g.nchunk : 2 * g.nchunk;
    // segment -> chunk, dt, the plane's byte offset, the temporal tap; step (segment, pair) -> weight tile offset
    auto seg_chunk = [&](int sg) { return pt == 0 ? sg : (sg >> 1); };
    auto seg_dt = [&](int sg) { return pt == 0 ? 0 : (sg & 1); };
    auto seg_ok = [&](int sg) { return tp + seg_dt(sg) < g.Tp; };
    auto seg_soff = [&](int sg) {
      const int pl = (n * g.Tp + tp + (seg_ok(sg) ? seg_dt(sg) : 0));
      return __builtin_amdgcn_readfirstlane((unsigned)pl * plane_bytes + (unsigned)(seg_chunk(sg) * 128));
    };
    auto step_woff = [&](int sg, int tap_hw) {
      const int at = pt == 0 ? 1 : (seg_dt(sg) == 0 ? 2 : 0);
      return __builtin_amdgcn_readfirstlane((unsigned)((at * 9 + tap_hw) * co2 + seg_chunk(sg) * 128));
    };
    f32x4_t acc[4][4][4];                     // [class][channel tile j][cell tile i]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s2_u32x4 fa[2][4], fb[2][4];              // [k step][tile]: ONE set, the two k steps leapfrog (reads of one under the MFMAs of the other)

    // fragment reads of k step KS of step (segment parity sb, pair P)
    auto read_frags = [&](auto KS, auto P, int sb, int wslot, s2_u32x4 (&fa)[2][4], s2_u32x4 (&fb)[2][4]) {
      constexpr int ks = decltype(KS)::value, p = decltype(P)::value;
      const int row = arow0 + s2_dh(p) * g.Wq + s2_dw(p);
      const unsigned aa = lds_base + (unsigned)(S2_A_BASE + sb * S2_ABUF) + (unsigned)(row << 7) + (unsigned)((((ks << 2) | fq) ^ (row & 7)) << 4);
      const unsigned ba = b_off[ks] + (unsigned)(wslot * S2_WSLOT);
      S2_DSR(fa[ks][0], aa, 0); S2_DSR(fa[ks][1], aa, 2048); S2_DSR(fa[ks][2], aa, 4096); S2_DSR(fa[ks][3], aa, 6144);
      S2_DSR(fb[ks][0], ba, 0); S2_DSR(fb[ks][1], ba, 2048); S2_DSR(fb[ks][2], ba, 4096); S2_DSR(fb[ks][3], ba, 6144);
    };

    // ---- prologue of the pass: window of segment 0, weight tiles of steps 0, 1, 2; fragments of step 0 ----
    __syncthreads();                          // (pass 1: every wave is past the epilogue's use of nothing in LDS -- and past pass 0's reads)
    {
      const unsigned so = seg_soff(0); const bool ok = seg_ok(0);
#pragma unroll
      for (int kk = 0; kk < 9; ++kk) issue_window(0, kk, so, ok);
      issue_weights(0, step_woff(0, s2_tap_hw(0)));
      issue_weights(1, step_woff(0, s2_tap_hw(1)));
      issue_weights(2, step_woff(0, s2_tap_hw(2)));
      issue_weights(3, step_woff(0, s2_tap_hw(3)));
      issue_weights(4, step_woff(0, s2_tap_hw(4)));
    }
    S2_VMCNT(0);
    __syncthreads();
    read_frags(S2C<0>{}, S2C<0>{}, 0, 0, fa, fb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    for (int sg0 = 0; sg0 < nseg; sg0 += 2) {
      const bool last_iter = sg0 + 2 >= nseg;
      auto step = [&](auto SG, auto P) {
        constexpr int sgi = decltype(SG)::value, p = decltype(P)::value;
        constexpr int lin = sgi * 9 + p;                       // step index inside the iteration (0..17); register set = lin & 1
        const int sg = sg0 + sgi;
        // -- counted wait.  Weight tiles travel FIVE steps ahead (ring of six slots; 18 steps per iteration: the slot of a step is
        // lin % 6); what this step reads -- tile s now, tile s + 1 half a step on -- was issued at step s - 4 at the latest, so the
        // issues of steps s - 3 .. s - 1 (three tiles = 6 instructions) may still fly.  The next segment's window goes out in ONE
        // burst behind the tile of pair 0 (9 instructions per wave, dy rows that may come from HBM): vmcnt retires in order, so the
        // burst is only waited for with the first tile issued behind it, five steps later (pair 5), well before pair 8 reads it --
        const bool fed = sg0 + sgi + 1 < nseg;                  // this segment issued a window burst at its pair 0
        const bool tail = last_iter && lin >= 14;               // the pass's last steps: fewer tiles behind them than the count assumes
        if (tail) S2_VMCNT(0);
        else if (p >= 1 && p <= 4 && fed) S2_VMCNT(15);
        else S2_VMCNT(6);
        __builtin_amdgcn_s_barrier();
        // -- DMA issue: the weight tile of step s + 5, then (pair 0) the next segment's window into the other buffer --
        {
          constexpr int p5 = (p + 5) % 9, o5 = (p + 5) / 9;
          const int sg5 = sg + o5;
          if (sg5 < nseg) issue_weights((lin + 5) % 6, step_woff(sg5, s2_tap_hw(p5)));
        }
        if constexpr (p == 0) {
          if (fed) {
            const unsigned so = seg_soff(sg + 1); const bool ok = seg_ok(sg + 1);
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) issue_window((sgi + 1) & 1, kk, so, ok);
          }
        }
        // -- k step 1 of this step is read under the MFMAs of k step 0, k step 0 of step s + 1 under those of k step 1 --
        constexpr int c = s2_cls(p);
        read_frags(S2C<1>{}, P, sgi & 1, lin % 6, fa, fb);
        __builtin_amdgcn_sched_barrier(0);           // (the eight reads go out AHEAD of the sixteen MFMAs that cover them: left alone, hipcc sinks them to the group's end)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[c][j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[0][j]), __builtin_bit_cast(bf16x8_t, fa[0][i]),
                                                                   acc[c][j][i], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
          constexpr int p1 = (p + 1) % 9, o1 = (p + 1) / 9;
          if (sg + o1 < nseg) read_frags(S2C<0>{}, S2C<p1>{}, (sgi + o1) & 1, (lin + 1) % 6, fa, fb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[c][j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[1][j]), __builtin_bit_cast(bf16x8_t, fa[1][i]),
                                                                   acc[c][j][i], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
#define S2_SEG(SGI) step(S2C<SGI>{}, S2C<0>{}); step(S2C<SGI>{}, S2C<1>{}); step(S2C<SGI>{}, S2C<2>{}); step(S2C<SGI>{}, S2C<3>{}); \
      step(S2C<SGI>{}, S2C<4>{}); step(S2C<SGI>{}, S2C<5>{}); step(S2C<SGI>{}, S2C<6>{}); step(S2C<SGI>{}, S2C<7>{}); step(S2C<SGI>{}, S2C<8>{});
      S2_SEG(0) S2_SEG(1)
#undef S2_SEG
    }

    // ---- epilogue of the pass: four classes, each a strided set of dx rows.  The accumulators of a class go through LDS (the tiles
    // are dead: [256 cells][64 channels] fp32, row pitch 272 bytes) and leave as whole 128-byte rows, eight lanes per cell, sixteen
    // bytes per lane -- few registers (the shared implicit-GEMM epilogue, inlined four times beside 256 live accumulators, spilled
    // to scratch) and full-line stores ----
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int SP = 68;                      // floats per staged cell row (64 + 4: the 16-byte writes of a lane row spread over the banks)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ph = c >> 1, pw = c & 1;
      // the eight rows this thread stores for the class, and (ahead of everything else: they come from HBM) their addend pieces
      int orow8[8]; uint4 add8[8];          // (element offsets below 2^31: the launcher checks)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int item = it * 256 + tid, cell = item >> 3, g8 = item & 7;
        const int q = q0 + cell;
        const int hq = fdiv(q, g.dWq), wq = q - hq * g.Wq;
        const int t = 2 * tp + pt, h = 2 * hq + ph, w = 2 * wq + pw;
        const bool ok = q < g.cells && wq < g.Wp && h < g.H && w < g.W;
        orow8[it] = ok ? (((n * g.T + t) * g.H + h) * g.W + w) * g.Ci + 64 * nt + 8 * g8 : -1;
        add8[it] = make_uint4(0, 0, 0, 0);
        if (addend != nullptr && ok) add8[it] = *reinterpret_cast<const uint4*>(addend + orow8[it]);
      }
      __syncthreads();                          // the tiles (or the previous class's rows) are dead
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4_t*>(stg + (wm0 + 16 * i + fr) * SP + 16 * j + 4 * fq) = acc[c][j][i];
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int item = it * 256 + tid, cell = item >> 3, g8 = item & 7;
        if (orow8[it] >= 0) {
          const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(stg + cell * SP + 8 * g8), v1 = *reinterpret_cast<const f32x4_t*>(stg + cell * SP + 8 * g8 + 4);
          float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          float a8[8]; unpack8(add8[it], a8);              // (zeros without an addend)
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] += a8[k];
          *reinterpret_cast<uint4*>(dx + orow8[it]) = pack8(f);
        }
      }
      __builtin_amdgcn_sched_barrier(0);        // (one class at a time: hoisting the next class's addresses and addend loads spills)
    }
  }
}

static long g_dgrad_s2_launches = 0;
extern "C" int64_t mscl_debug_dgrad_s2_launches(void) { return g_dgrad_s2_launches; }     // tests: which kernel took a launch

// returns 1 if launched, 0 if the shape is not this kernel's, > 0 on a launch error
int mscl_launch_dgrad_s2(const mscl_conv_desc* d, const bf16_t* dy, const bf16_t* wT, bf16_t* dx, const bf16_t* addend, hipStream_t st) {
  if (d->kT != 3 || d->kH != 3 || d->kW != 3 || d->sT != 2 || d->sH != 2 || d->sW != 2 || d->pT != 1 || d->pH != 1 || d->pW != 1) return 0;
  if ((d->K % 128) || (d->C % 64)) return 0;                    // dy channels in pairs of 64-channel chunks; dx channels in tiles of 64
  const int Wq = d->Wo + 1;
  if (S2_CM + Wq + 1 > S2_AROWS) return 0;                      // the window of a tile: 256 + W' + 2 rows
  if ((long)d->N * d->To * d->Ho * d->Wo * d->K * 2 >= (1L << 31) || (long)d->C * 27 * d->K * 2 >= (1L << 31)) return 0;
  if ((long)d->N * d->T * d->H * d->W * d->C >= (1L << 31)) return 0;          // 32-bit element offsets into dx
  static MsclTune t_on("MSCL_DGRAD_S2");                        // A/B aid: 0 = the implicit-GEMM parity classes; 2 = also on small maps (tests)
  const int level = t_on.get(1);
  if (level == 0) return 0;
  S2Geom g{};
  g.N = d->N; g.Tp = d->To; g.Hp = d->Ho; g.Wp = d->Wo; g.T = d->T; g.H = d->H; g.W = d->W; g.Co = d->K; g.Ci = d->C;
  g.Wq = Wq; g.cells = d->Ho * Wq; g.tiles = (g.cells + S2_CM - 1) / S2_CM; g.nct = d->C / 64; g.nchunk = d->K / 64;
  const long blocks = (long)d->N * d->To * g.tiles * g.nct;
  // one 256-cell tile is 13 us of MFMA work at the very best: a map that does not give (nearly) every CU a block is faster on the
  // split-K implicit-GEMM kernel (layers 3 and 4 of R3D-18: 64 and 32 blocks)
  if ((blocks < 192 && level < 2) || blocks >= (1L << 30)) return 0;
  g.dWq = make_fastdiv(Wq); g.dTiles = make_fastdiv(g.tiles); g.dTp = make_fastdiv(d->To); g.dNct = make_fastdiv(g.nct);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(dgrad_s2_kernel, dim3((unsigned)blocks), dim3(256), S2_LDS, st, g, dy, wT, dx, addend);
  hipError_t e = hipGetLastError(); if (e != hipSuccess) return (int)e;
  ++g_dgrad_s2_launches;
  return 1;
}
