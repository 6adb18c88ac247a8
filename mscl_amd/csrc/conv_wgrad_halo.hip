// Weight gradient of the 3x3x3 / stride 1 / pad 1 convolutions with the input window resident in LDS (layer 1: 64 -> 64 on
// 56 x 56 planes; since round 4 also the 128 -> 128 layers on 28 x 28 planes and the 256 -> 256 layers on 14 x 14 planes, as
// 64 x 64 channel slices):  dW[co][kt][a][b][ci] = sum over positions of dy[pos][co] * x[plane t+kt-1][pos + (a-1, b-1)][ci].
// Reference op: the weight gradient autograd computes for Conv3DSimple / BasicBlock (mmaction/models/backbones/r3d.py:16-34,
// 95-127) and the SEPC PConv3D convolutions (necks/sepc.py:57-135).
//
// Why: the general kernel (conv_wgrad.hip) re-stages the x rows once per tap; at 64 x 64 channels a step stages 16 KB
// for 32 MFMAs = 128 B per MFMA-clock of a CU, twice what the global -> LDS path delivers: 254 us = 350 TFLOP/s on layer 1.
// Here a block walks (plane tile, kt) items of ONE kt and ONE (64 co, 64 ci) channel slice: per item it stages the 376-row
// window of the source plane (padded-linear order, see conv_halo.hip) and the 256-position dy tile ONCE (79 KB) and runs all 9
// in-plane taps against them, 1152 MFMAs: 69 B per MFMA.  The 9 x 64 x 64 partial products stay in registers (144 accumulators
// per lane) over all items of the block; at the end every block stores its slab with plain stores and a small second kernel
// adds the slabs into dW (float atomics of 147 KB per block would cost more than the GEMM, MI355X_MICROARCH.md).
//  * both operands are position-major, so fragments are column reads: ds_read_b64_tr_b16 (as conv_wgrad.hip);
//  * pad columns / rows outside the plane are zero in BOTH tiles (buffer range check), so they add nothing;
//  * items are double-buffered in LDS (2 x 79 KB): the next item's DMA pieces are issued between the k steps.
// Round 4: (1) the round-3 form ran ONE wave per SIMD whose single instruction stream also issued the 20 DMA pieces of the next
// item (60-180 issue cycles each) and waited on its own transposing reads: 6.3 us per item against 1.9 us of MFMA work, 12 GB/s
// per CU of fill -- neither roof.  Now two waves per SIMD split the k steps of an item (waves 4-7 take positions 128-255 of the
// tile with a full set of 144 accumulators of their own; the pairs are added through LDS once, at the end of the block), so one
// wave's DMA issue and read latencies are covered by its partner's MFMAs, and each wave issues half the pieces: layer 1
// 128.3 -> 97.9 us (A/B in one process).  (2) Items are walked tile-major (consecutive planes of one plane tile) and the blocks
// of a slot -- its three kt, and its channel slices -- get consecutive logical ids on one XCD: a dy tile is fetched by all of
// them at the same time and an x plane at three consecutive items, so beyond the XCD's L2 both maps are read about once instead
// of three times.  (3) The reduction rows of a k step are dealt to the lanes so that the second read of a fragment and the k step
// are immediate offsets (whswz): 10 address registers fewer, which is what lets 144 accumulators fit two waves per SIMD.
// (4) Channel slices: a layer with C / K multiples of 64 runs as (K/64) x (C/64) independent 64 x 64 problems on the same
// positions (row pitch = the map's channel count): the 128-channel layers of the step (four 256-position tiles per 28 x 30
// padded plane: 82 % of the tile rows live) and the 256-channel layers (one tile per 14 x 16 plane: 87 %).
#include "common.h"
#include <cstdlib>

struct WHGeom {
  int N, T, H, W, HW, Wp, tiles;   // tiles per plane (256 padded-linear positions each)
  int total;                        // plane tiles = N * T * tiles
  int gk;                           // slots: blocks per (kt, channel slice)
  int planes;                       // N * T
  int C, K, ncs, nsub;              // channels of x / dy; ci slices (C / 64); blocks per slot = KT * (K/64) * (C/64)
  int KT;                           // temporal taps: 3 (pad 1) or 1 (pad 0: the 1x3x3 conv2 of the Bottleneck trunks, round 6)
  FastDiv dWp, dTiles, dT, dPlanes;
};

constexpr int WH_XROWS = 376, WH_XBYTES = WH_XROWS * 128, WH_DYBYTES = 256 * 128, WH_STAGE = WH_XBYTES + WH_DYBYTES;
constexpr unsigned WH_OOB = 0x80000000u;
constexpr int WH_SLAB = 9 * 64 * 64;

// XOR key on the 16-byte granule index of a 128-byte row.  The 32 reduction rows of a k step are dealt to the lanes as
// row = 4 * (lane >> 4) + ((lane >> 2) & 3) + 16 h (h = the first / second transposing read of a fragment) -- any order of the
// reduction index serves, as long as both operands use the same -- so that h and the k step are IMMEDIATE offsets of one address
// register per fragment (+2048 / +4096 bytes; bits 1, 2 of the row, which the key is made of, do not move) and a 32-lane group of
// a read touches 8 consecutive rows: 8 distinct 32-byte slots of the bank row under this key, whatever the tap's row shift.
__device__ __forceinline__ int whswz(int row) { return row & 6; }
__device__ __forceinline__ auto wh_rsrc(const void* p) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7FFFFFFF, 0x00020000);
}
typedef __attribute__((address_space(3))) void* wh_lds_t;
typedef __attribute__((ext_vector_type(4))) short wh_s16x4;
typedef __attribute__((ext_vector_type(8))) short wh_s16x8;

// NW = waves per block: two per SIMD that split the k steps of every item (see the header; the one-wave-per-SIMD form of round 3
// measured 128 vs 98 us on layer 1 and was deleted in round 5 with its switch MSCL_WGRAD_HALO_WAVES)
constexpr int WH_NW = 8;
__global__ __launch_bounds__(64 * WH_NW, 1) void wgrad_halo64_kernel(const WHGeom g, const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ dy, float* __restrict__ slabs) {
  constexpr int NW = WH_NW;
  constexpr int NG = NW / 4;                               // wave groups splitting the 8 k steps of an item
  constexpr int RPP = 8 * NW;                              // rows per DMA pass (64 * NW threads x 16 B)
  constexpr int XP = (WH_XROWS + RPP - 1) / RPP, DP = 256 / RPP, NP = XP + DP;      // passes: 12 + 8 / 6 + 4
  constexpr int TRIPS = 4 / NG;                            // trips of two k steps per item and group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, grp = wave >> 2;                // ci tile of the wave; k-step group
  // the three kt blocks of a slot: consecutive logical ids = one XCD (its L2 then serves two of the three reads of every tile)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int sub = lin % g.nsub, slot = lin / g.nsub;       // sub = (co slice * ncs + ci slice) * KT + kt
  const int KT = __builtin_amdgcn_readfirstlane(g.KT), ktc = KT >> 1;      // ktc = the temporal pad
  const int kt = sub % KT, cs = (sub / KT) % g.ncs, kslice = sub / (KT * g.ncs);
  const int c2 = g.C * 2, k2 = g.K * 2;                    // row pitch of x / dy in bytes
  const auto rs_x = wh_rsrc(x);
  const auto rs_dy = wh_rsrc(dy);
  // wave tile: all 64 co x ci [16*wq, +16): one B fragment per step feeds 4 MFMAs (a 32 x 32 tile needs twice the
  // transposing reads per MFMA, and those, not the MFMAs, then set the pace)
  const int fgp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;

  // accumulators: [tap 0..8][co tile 0..3]
  f32x4_t acc[9][4];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t9][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // items of this block: a contiguous range of the TILE-MAJOR order (item i = tile i / planes of plane i % planes), the same range
  // for the three kt blocks of the slot
  const int i_end = (int)((long)(slot + 1) * g.total / g.gk);
  constexpr int i_step = 1;
  auto item_plane = [&](int it) { return it - fdiv(it, g.dPlanes) * g.planes; };
  auto next_valid = [&](int it) {
    while (it < i_end) {
      const int plane = item_plane(it);
      const int t = plane - fdiv(plane, g.dT) * g.T;
      if ((unsigned)(t + kt - ktc) < (unsigned)g.T) break;
      it += i_step;
    }
    return it;
  };
  // Per-item DMA state.  Piece k < XP stages window rows RPP k + (tid >> 3), piece XP + k' dy rows RPP k' + (tid >> 3).
  // Consecutive pieces advance a row's padded-linear position by RPP (no per-piece offset arrays: they would need dynamic
  // register indexing); the swizzle key of a row is unchanged by +32 / +64 (bits 1 and 2), so the lane's source granule is fixed.
  const int prow = tid >> 3, pg = tid & 7;
  const unsigned xg = (unsigned)((pg ^ whswz(prow)) * 16 + cs * 128), dg = (unsigned)((pg ^ whswz(prow)) * 16 + kslice * 128);
  int x_q = 0, d_q = 0;                                    // padded-linear position of the row of the NEXT piece of each kind
  unsigned xs = 0, ds = 0;
  auto prepare = [&](int it) {
    const int tile = fdiv(it, g.dPlanes), plane = it - tile * g.planes;
    const int q0 = g.Wp + tile * 256;
    x_q = q0 - g.Wp - 1 + prow;                            // >= -1
    d_q = q0 + prow;
    xs = __builtin_amdgcn_readfirstlane((unsigned)((plane + kt - ktc) * g.HW) * (unsigned)c2);
    ds = __builtin_amdgcn_readfirstlane((unsigned)(plane * g.HW) * (unsigned)k2);
  };
  auto issue_piece = [&](int k, int stage) {               // k (wave-uniform, runtime) in [0, NP), issued in order
    unsigned char* base = smem + stage * WH_STAGE;
    if (k < XP) {
      const int x_hp = fdiv(x_q < 0 ? 0 : x_q, g.dWp), x_wp = x_q - x_hp * g.Wp;
      const bool ok = x_q >= 0 && x_hp >= 1 && x_hp <= g.H && x_wp >= 1 && x_wp <= g.W;
      const unsigned vo = ok ? (unsigned)(((x_hp - 1) * g.W + (x_wp - 1)) * c2) + xg : WH_OOB;
      x_q += RPP;
      if (!(k == XP - 1 && wave == NW - 1))                // rows 376..383 do not exist
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (wh_lds_t)(base + (k * 64 * NW + wave * 64) * 16), 16, vo, xs, 0, 0);
    } else {
      const int d_hp = fdiv(d_q, g.dWp), d_wp = d_q - d_hp * g.Wp;
      const bool ok = d_hp <= g.H && d_wp >= 1 && d_wp <= g.W;
      const unsigned vo = ok ? (unsigned)(((d_hp - 1) * g.W + (d_wp - 1)) * k2) + dg : WH_OOB;
      d_q += RPP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (wh_lds_t)(base + WH_XBYTES + ((k - XP) * 64 * NW + wave * 64) * 16), 16, vo, ds, 0, 0);
    }
  };

  int pt = next_valid((int)((long)slot * g.total / g.gk));
  int cur = 0;
  if (pt < i_end) {
    prepare(pt);
    for (int k = 0; k < NP; ++k) issue_piece(k, 0);
  }
  // Per-lane LDS addresses of the transposing reads.  Row r = r_l + 16h + 32*ks + shift(tap); the XOR swizzle key uses
  // bits 1 and 2 of r, which 16h and 32*ks do not touch: ONE address per tap, h and the k step go into the instruction's
  // immediate offset, and co tile i is tile 0 XOR 32*i (granule bits 1, 2 of the address are otherwise only keyed).
  const int r_l = 4 * fgp + qq;
  const int sub8 = (pp & 1) * 8;
  const int a_addr = WH_XBYTES + r_l * 128 + (((pp >> 1) ^ whswz(r_l)) * 16) + sub8;     // dy tile, co tile 0
  int b_addr[9];                           // window, this wave's ci tile, per tap (relative to the stage base)
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int r = r_l + (t9 / 3) * g.Wp + (t9 % 3);
    b_addr[t9] = r * 128 + (((2 * wq + (pp >> 1)) ^ whswz(r)) * 16) + sub8;
  }
  // Transposing reads are inline asm with hand-counted lgkmcnt waits: hipcc guards every LDS read it can see with
  // s_waitcnt vmcnt(0) while an LDS-DMA is in flight (it cannot know the DMA fills the OTHER stage), which would put
  // each of the next item's DMA pieces' full latency into this item's MFMA stream.
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)(smem);
  const unsigned kofs = (unsigned)(grp * TRIPS * 2 * 4096);       // first k step of this wave's group
#define WH_TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")

  while (pt < i_end) {
    const int nxt = next_valid(pt + i_step);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this item's tiles landed
    __syncthreads();                                       // ... for every wave; the other stage is no longer being read
    const bool has_next = nxt < i_end;
    if (has_next) prepare(nxt);
    // 72 steps (8 k steps x 9 taps) as trips of 18 (2 per wave): static register slots (A double buffer
    // by k-step parity, B ring of 3), the trip's k base lives in the address registers, the k step inside a trip in the
    // immediate offset.  Operands are fetched two steps ahead; the last trip's look-ahead reads fall beyond the wave's k steps
    // (harmless, never used) so that the hand-counted waits stay the same on every trip.
    unsigned pa, pb[9];
    const unsigned st = lds0 + (unsigned)(cur * WH_STAGE) + kofs;
    pa = st + (unsigned)a_addr;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) pb[t9] = st + (unsigned)b_addr[t9];
    wh_s16x4 va[2][4][2], vb[3][2];                        // [slot][co tile][h], [slot][h]
#define WH_READ_A(KSL, SLOT) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { \
      WH_TR_READ(va[SLOT][i_][0], pa ^ (unsigned)(i_ * 32), (KSL) * 4096); \
      WH_TR_READ(va[SLOT][i_][1], pa ^ (unsigned)(i_ * 32), (KSL) * 4096 + 2048); } } while (0)
#define WH_READ_B(LSTEP, SLOT) do { WH_TR_READ(vb[SLOT][0], pb[(LSTEP) % 9], ((LSTEP) / 9) * 4096); \
                                    WH_TR_READ(vb[SLOT][1], pb[(LSTEP) % 9], ((LSTEP) / 9) * 4096 + 2048); } while (0)
    WH_READ_A(0, 0);
    WH_READ_B(0, 0);
    WH_READ_B(1, 1);
    int piece = 0;
    for (int trip = 0; trip < TRIPS; ++trip) {
#pragma unroll
      for (int ls = 0; ls < 18; ++ls) {
        const int ksl = ls / 9, t9 = ls % 9;
        // look-ahead (A first: it must be older than the B fragments of its k step); local steps 18, 19 = next trip's 0, 1
        if ((ls + 2) % 9 == 0) {
          if ((ls + 2) / 9 == 1) WH_READ_A(1, 1); else WH_READ_A(2, 0);
        }
        switch (ls + 2) {                  // (the immediate offset must be a literal)
#define WH_CASE(L) case L: WH_READ_B(L, (L) % 3); break;
          WH_CASE(2) WH_CASE(3) WH_CASE(4) WH_CASE(5) WH_CASE(6) WH_CASE(7) WH_CASE(8) WH_CASE(9) WH_CASE(10) WH_CASE(11)
          WH_CASE(12) WH_CASE(13) WH_CASE(14) WH_CASE(15) WH_CASE(16) WH_CASE(17) WH_CASE(18) WH_CASE(19)
#undef WH_CASE
          default: break;
        }
        // next item's tiles: 10 pieces per wave, all in the first of its two trips (odd steps and step 16): a whole trip
        // (>= 1150 cycles of MFMA work) lies between the last issue and the next item's wait.
        const bool islot = trip == 0 && ((ls & 1) != 0 || ls == 16);
        if (has_next && islot && piece < NP) { issue_piece(piece, cur ^ 1); ++piece; }
        // this step's operands were issued two steps ago: everything younger may stay in flight
        const int y1 = ((ls + 1) % 9 == 0) ? 10 : 2;       // reads issued by the previous step
        const int y0 = ((ls + 2) % 9 == 0) ? 10 : 2;       // ... and by this one
        wh_s16x4 &b0 = vb[ls % 3][0], &b1 = vb[ls % 3][1];
        if (y0 + y1 == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b0), "+v"(b1));
        else asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(b0), "+v"(b1));
        if (t9 == 0) {   // first use of this k step's A fragments: older than the B fragment just waited for
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(va[ksl][i][0]), "+v"(va[ksl][i][1]));
        }
        wh_s16x8 wb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, wb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const wh_s16x4 v0 = va[ksl][i][0], v1 = va[ksl][i][1];
          wh_s16x8 w8 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          acc[t9][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w8), fb, acc[t9][i], 0, 0, 0);
        }
      }
      // next trip: two k steps further
      pa += 8192u;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) pb[t9] += 8192u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // drain the unused look-ahead reads before the stage is reused
    pt = nxt;
    cur ^= 1;
  }
#undef WH_READ_A
#undef WH_READ_B
#undef WH_TR_READ

  // Registers leave in REGISTER order: a lane's accumulator (tap t9, co tile i) is one float4 (co = 16 i + 4 (lane >> 4) + r, ci =
  // 16 wq + (lane & 15)), and both the hand-over between the SIMD partners and the slab are laid out [wq][t9 * 4 + i][lane] float4 --
  // 36 ds_write_b128 / ds_read_b128 and 36 global_store_dwordx4 per lane, every wave instruction 1 KB contiguous (the [tap][co][ci]
  // order of rounds 3-4 took 144 dword instructions for each, the stores in 64-byte segments).  The reduce kernel undoes the order.
  {
    // ---- the k-step halves of a SIMD pair: waves 4-7 hand their 144 partial sums per lane over through LDS (4 x 36 KB, the
    // stages are dead), waves 0-3 add them in a fixed order ----
    __syncthreads();
    float4* red4 = reinterpret_cast<float4*>(smem);
    if (grp == 1) {
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          red4[((wq * 36 + t9 * 4 + i) << 6) + lane] = make_float4(acc[t9][i][0], acc[t9][i][1], acc[t9][i][2], acc[t9][i][3]);
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = red4[((wq * 36 + t9 * 4 + i) << 6) + lane];
        acc[t9][i][0] += v.x; acc[t9][i][1] += v.y; acc[t9][i][2] += v.z; acc[t9][i][3] += v.w;
      }
  }
  float4* slab4 = reinterpret_cast<float4*>(slabs + (long)lin * WH_SLAB);
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      slab4[((wq * 36 + t9 * 4 + i) << 6) + lane] = make_float4(acc[t9][i][0], acc[t9][i][1], acc[t9][i][2], acc[t9][i][3]);
}

// dw[co][kt*9 + t9][ci] += sum over the gk slabs of (kt, channel slice), in a fixed order.  One thread per float4 of a slab (the four
// co of one accumulator): 16-byte loads that a wave reads 1 KB at a time, four owned elements of dw updated by plain adds.
__global__ __launch_bounds__(256) void wgrad_halo64_reduce_kernel(const float4* __restrict__ slabs4, float* __restrict__ dw, int gk, int nsub,
                                                                  int ncs, int C, long total4, int KT) {
  constexpr int S4 = WH_SLAB / 4;                          // float4 per slab
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int sub = (int)(t / S4), q = (int)(t - (long)sub * S4);      // q = (wq * 36 + t9 * 4 + i) * 64 + lane
  const float4* p = slabs4 + (long)sub * S4 + q;
  const long bstride = (long)nsub * S4;
  float4 s4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // four independent load chains
  int b = 0;
  for (; b + 4 <= gk; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float4 v = p[(long)(b + u) * bstride]; s4[u].x += v.x; s4[u].y += v.y; s4[u].z += v.z; s4[u].w += v.w; }
  }
  for (; b < gk; ++b) { const float4 v = p[(long)b * bstride]; s4[0].x += v.x; s4[0].y += v.y; s4[0].z += v.z; s4[0].w += v.w; }   // (<= 3 slabs)
  const float r4[4] = {(s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y),
                       (s4[0].z + s4[1].z) + (s4[2].z + s4[3].z), (s4[0].w + s4[1].w) + (s4[2].w + s4[3].w)};
  const int lane = q & 63, u36 = q >> 6, wq = u36 / 36, v36 = u36 - wq * 36, t9 = v36 >> 2, i = v36 & 3;
  const int kt = sub % KT, cs = sub / KT, cis = cs % ncs, cos = cs / ncs;
  const int co0 = cos * 64 + i * 16 + (lane >> 4) * 4, ci = cis * 64 + wq * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) dw[((long)(co0 + r) * (9 * KT) + kt * 9 + t9) * C + ci] += r4[r];     // one owner per element: plain adds, the same bits every run
}

static long g_wgrad_halo_launches = 0;
extern "C" int64_t mscl_debug_wgrad_halo_launches(void) { return g_wgrad_halo_launches; }

static int wh_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  return cus;
}
// which layers take this kernel: 3x3x3 / 1 / 1 (and, round 6, 1x3x3 / 1 / (0,1,1): SlowOnly-50 layers 1-3 78.6 / 77.0 / 63.8 -> 62.4 / 54.0 / 53.0 us,
// the 28 x 28 FPN level of the R3D-18 step 45.2 -> 37.5 us alone, the step unchanged), channel counts multiples of 64, planes whose padded rows fit the 376-row window;
// planes of at least MSCL_WGRAD_HALO_MIN padded positions (default 200: the 14 x 16 planes of layer 3 fill 87 % of one
// 256-position tile; the 7 x 9 planes of layer 4 would fill 25 % and stay with the general kernel)
static bool wh_shape(const mscl_conv_desc* d) {
  if (d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pH != 1 || d->pW != 1) return false;
  if (!((d->kT == 3 && d->pT == 1) || (d->kT == 1 && d->pT == 0))) return false;
  if ((d->C % 64) || (d->K % 64) || d->C > 512 || d->K > 512) return false;
  const int Wp = d->W + 2;
  if (256 + 2 * Wp + 2 > WH_XROWS) return false;
  if ((long)d->N * d->T * d->H * d->W * d->C * 2 >= (1L << 31) || (long)d->N * d->T * d->H * d->W * d->K * 2 >= (1L << 31)) return false;
  static MsclTune t_min("MSCL_WGRAD_HALO_MIN");
  if ((long)d->H * Wp < t_min.get(200)) return false;
  const int nsub = d->kT * (d->C / 64) * (d->K / 64);
  if (nsub > wh_cus()) return false;
  // enough plane tiles for every block to walk a few: a block that stages one or two items pays its prologue, its pair reduction
  // and its 147-KB slab for nothing (128 -> 128 on the 4 x 14 x 14 pyramid level, 32 items over 21 x 12 blocks: 27.8 vs 23.0 us
  // for the general kernel)
  const long items = (long)d->N * d->T * ((d->H * Wp + 255) / 256);
  static MsclTune t_items("MSCL_WGRAD_HALO_ITEMS");
  return items * nsub >= (long)t_items.get(4) * wh_cus();
}
static int wh_slots(const mscl_conv_desc* d) {
  const int nsub = d->kT * (d->C / 64) * (d->K / 64);
  const int tiles = (d->H * (d->W + 2) + 255) / 256;
  int gk = wh_cus() / nsub;
  if (gk > d->N * d->T * tiles) gk = d->N * d->T * tiles;
  return gk < 1 ? 1 : gk;
}
// floats of workspace mscl_wgrad_halo64 wants (0: the layer is not covered)
extern "C" int64_t mscl_wgrad_halo_ws(const mscl_conv_desc* d) {
  if (!d || !wh_shape(d)) return 0;
  return (int64_t)wh_slots(d) * d->kT * (d->C / 64) * (d->K / 64) * WH_SLAB;
}

// returns 1 if launched, 0 if the shape / workspace is not covered, <0 / >0 on error
int mscl_wgrad_halo64(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats,
                      hipStream_t st) {
  if (!wh_shape(d)) return 0;
  WHGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.HW = d->H * d->W; g.Wp = d->W + 2;
  g.C = d->C; g.K = d->K; g.ncs = d->C / 64; g.KT = d->kT; g.nsub = g.KT * g.ncs * (d->K / 64);
  g.tiles = (d->H * g.Wp + 255) / 256;
  g.total = d->N * d->T * g.tiles;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_halo64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  int gk = wh_slots(d);
  if (ws == nullptr || (long)gk * g.nsub * WH_SLAB > ws_floats) gk = (int)(ws ? ws_floats / ((long)g.nsub * WH_SLAB) : 0);
  if (gk < 1) return 0;
  g.gk = gk;
  g.planes = d->N * d->T;
  g.dWp = make_fastdiv(g.Wp); g.dTiles = make_fastdiv(g.tiles); g.dT = make_fastdiv(d->T); g.dPlanes = make_fastdiv(g.planes);
  const unsigned blocks = (unsigned)(g.nsub * gk);
  hipLaunchKernelGGL(wgrad_halo64_kernel, dim3(blocks), dim3(64 * WH_NW), (size_t)2 * WH_STAGE, st, g, x, dy, ws);
  MSCL_LAUNCH_CHECK();
  const long total4 = (long)g.nsub * (WH_SLAB / 4);
  hipLaunchKernelGGL(wgrad_halo64_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, (const float4*)ws, dw, gk, g.nsub,
                     g.ncs, d->C, total4, g.KT);
  MSCL_LAUNCH_CHECK();
  ++g_wgrad_halo_launches;
  return 1;
}
