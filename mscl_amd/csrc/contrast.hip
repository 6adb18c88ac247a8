// MoCo InfoNCE against the aged negative queue (streaming, never materialises the R x 65537 logits),
// bit-exact queue bookkeeping, and the frame-level LMCL loss (gfx950).
// Roofline: HBM.  One forward pass reads the queue once (dim*K*4 B, 33.5 MB at K=65536) for ALL query
// rows that share the snapshot; lanes run along K so every wave reads whole contiguous rows of the
// (dim, K) buffer; query rows are wave-uniform and come through the scalar cache.
#include "common.h"

#define NCE_BCOLS 128         // queue columns per block, TWO per lane: a wave-instruction reads 512 B of one queue row
#define NCE_WAVES 4           // the 4 waves split the feature dimension

// "Virtual enqueue": the snapshot the reference takes AFTER _dequeue_and_enqueue(keys) (moco.py:423-440) -- every age +1, the
// n_new columns from *ptr on replaced by the new keys at age 1 -- read straight from the queue as it stands BEFORE that write
// plus the keys (keys: n_new x dim fp32 row-major, ptr: the device queue_ptr).  Same arithmetic on the same values as after the
// real write, so the pass that needs the later snapshot (App. E-3: the rotated-flow and rf terms) no longer waits for the pass
// on the earlier one.  keys == NULL: the queue as it stands.
struct NceVirt { const float* keys; const int64_t* ptr; int n_new; };
__device__ __forceinline__ bool nce_in_new(const NceVirt& v, int k, int& j) {
  if (v.keys == nullptr) return false;
  j = k - (int)*v.ptr;
  return j >= 0 && j < v.n_new;
}

// Round 5: wider pieces, two blocks per CU.  Rounds 1-4 gave a block 64 columns, one per lane: a wave-instruction read 256 bytes of
// a queue row, the next row's piece lay 256 KB further on, and a wave held 32 such loads: 28.6 / 38.2 us per pass over the 33.5-MB
// queue (1.1 TB/s, 0.14 of the HBM peak) on the step's serial loss phase.  Measured on the way here: 256 columns per block, four per
// lane (1-KB pieces, one 110-KB block per CU) 17.0 / 34.9 us -- with ONE wave per SIMD the 3072 multiply-adds of a lane (24 rows),
// the load round trips and the LDS exchange follow one another instead of overlapping.  Kept: 128 columns per block, two per lane
// (512-byte pieces), every load of the lane's 32 channels in flight before the first use, and LDS small enough for TWO blocks per CU
// (two waves per SIMD: one block's arithmetic runs under the other's loads).
// Wave w takes channels [w dim/4, (w + 1) dim/4) of the dot products; the four partial sums of a (row, column) meet through LDS.
// Query rows are staged once per block as qs[c][RT] (channel-major): the rows of one channel are then a few broadcast
// ds_read_b128; fetching them through the scalar cache (RT s_loads per channel, each waited for) kept the pass at 0.5 TB/s.
// (qs always holds 128 channel rows: the waves take fixed 32-channel shares of a 128-channel space, channels >= dim are zeros --
// loops over a runtime channel count sent the load buffer to scratch and put a wait behind every load)
template <int RT>
__device__ __forceinline__ void nce_q_load(const float* __restrict__ q, float (&qr)[RT / 2], int R, int dim) {
#pragma unroll
  for (int n = 0; n < RT / 2; ++n) {          // 128 RT elements over 256 threads; issued AHEAD of the queue loads, stored behind them
    const int i = threadIdx.x + 256 * n, c = i / RT, r = i - c * RT;
    const bool ok = r < R && c < dim;
    const float v = q[ok ? r * dim + c : 0];
    qr[n] = ok ? v : 0.f;
  }
}
template <int RT>
__device__ __forceinline__ void nce_q_store(const float (&qr)[RT / 2], float* __restrict__ qs) {
#pragma unroll
  for (int n = 0; n < RT / 2; ++n) qs[threadIdx.x + 256 * n] = qr[n];
}

// the lane's columns: first column (clamped for a lane past the end: K is even, so a lane is live or dead as a whole), which of
// them the virtual enqueue replaces, their age decay.  In two steps: the positions need no memory and the queue loads go out on
// them; the ages (two dependent round trips: the queue pointer, then the counts) are fetched BEHIND those loads and turned into
// decays after the arithmetic -- at the head of the kernel they held the queue loads back by their full latency.
struct NceCols { int k0; bool live; int newmask; int jn[2]; float decay[2]; };
__device__ __forceinline__ NceCols nce_cols(int blk, int lane, int K, const NceVirt& vt) {
  NceCols c;
  const int k = blk * NCE_BCOLS + lane * 2;
  c.live = k < K;
  c.k0 = c.live ? k : K - 2;
  c.newmask = 0;
  c.jn[0] = c.jn[1] = 0; c.decay[0] = c.decay[1] = 0.f;
  if (vt.keys != nullptr) {                    // (wave-uniform; the plain passes skip the pointer load)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (nce_in_new(vt, c.k0 + j, c.jn[j])) c.newmask |= 1 << j;
  }
  return c;
}
__device__ __forceinline__ void nce_decay(NceCols& c, const int64_t* __restrict__ count, const NceVirt& vt) {
  const long a0 = count[c.k0], a1 = count[c.k0 + 1];
  const float age0 = vt.keys == nullptr ? (float)a0 : ((c.newmask & 1) ? 1.f : (float)(a0 + 1));
  const float age1 = vt.keys == nullptr ? (float)a1 : ((c.newmask & 2) ? 1.f : (float)(a1 + 1));
  c.decay[0] = powf(0.99999f, age0); c.decay[1] = powf(0.99999f, age1);      // recognizers/moco.py:484
}

// wv[i] = the lane's two columns of queue row c0 + i in the snapshot `vt` describes, every load issued before the first use; the rare
// lanes whose columns the virtual enqueue replaces patch the components in
__device__ __forceinline__ void nce_load_cols(const float* __restrict__ queue, float2 (&wv)[32], int c0, int dim, int K,
                                              const NceCols& cl, const NceVirt& vt) {
  // unconditional loads from clamped rows, zeroed afterwards by a select: a load under a condition becomes a branch with a wait
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = c0 + i < dim ? c0 + i : dim - 1;
    wv[i] = *reinterpret_cast<const float2*>(queue + (long)c * K + cl.k0);
  }
  if (cl.newmask) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = c0 + i < dim ? c0 + i : dim - 1;
      if (cl.newmask & 1) wv[i].x = vt.keys[(long)cl.jn[0] * dim + c];
      if (cl.newmask & 2) wv[i].y = vt.keys[(long)cl.jn[1] * dim + c];
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i)
    if (c0 + i >= dim) wv[i] = make_float2(0.f, 0.f);
}

// acc[r] (a float2: the lane's two columns) = sum over this wave's channels of q[r][c] * W[c][k0 .. k0 + 1]
template <int RT>
__device__ __forceinline__ void nce_partial(const float* __restrict__ qs, const float2 (&wv)[32], int c0, float2 (&acc)[RT]) {
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = make_float2(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float2 w = wv[i];
    const float4* qv = reinterpret_cast<const float4*>(qs + (c0 + i) * RT);
#pragma unroll
    for (int r4 = 0; r4 < RT / 4; ++r4) {
      const float4 v = qv[r4];
      const float qr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float2& a = acc[4 * r4 + u];
        a.x = fmaf(qr[u], w.x, a.x); a.y = fmaf(qr[u], w.y, a.y);
      }
    }
    // left alone, the scheduler hoists the (independent) query-row reads of ALL channels above the arithmetic: hundreds of registers, spills
    if (i & 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// part[(blk*R + r)*3 + {0: max, 1: sum exp(l - max), 2: #(l > pos)}], one block per 128 columns
template <int RT>
__global__ __launch_bounds__(256, 2) void nce_fwd_kernel(const float* __restrict__ queue, const int64_t* __restrict__ count,
                                                         const float* __restrict__ q, const float* __restrict__ pos,
                                                         float* __restrict__ part, int R, int dim, int K, float inv_T, const NceVirt vt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // qs[dim][RT] | red[4][RT][64 lanes] float2
  float* qs = sm; float2* red = reinterpret_cast<float2*>(qs + 128 * RT);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: row bases in SGPRs)
  const int c0 = wave * 32;
  NceCols cl = nce_cols(blockIdx.x, lane, K, vt);
  float qr[RT / 2];
  nce_q_load<RT>(q, qr, R, dim);
  float2 wv[32];
  nce_load_cols(queue, wv, c0, dim, K, cl, vt);              // (the loads travel while the query rows are staged)
  nce_decay(cl, count, vt);
  nce_q_store<RT>(qr, qs);
  __syncthreads();
  {
    float2 acc[RT];
    nce_partial<RT>(qs, wv, c0, acc);
#pragma unroll
    for (int r = 0; r < RT; ++r) red[(wave * RT + r) * 64 + lane] = acc[r];
  }
  // (the tile stays "in use" to here, as it does in the backward kernel, which compiles to 138 registers: without a later use hipcc
  // turns this kernel's channel loop inside out -- query-row reads of nine channels ahead of the first multiply -- and spills 209)
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("" :: "v"(wv[i].x), "v"(wv[i].y));
  __syncthreads();
  // wave w finishes rows r = w, w+4, ...
#pragma unroll 1
  for (int r = wave; r < R; r += NCE_WAVES) {
    const float2 d0 = red[(0 * RT + r) * 64 + lane], d1 = red[(1 * RT + r) * 64 + lane], d2 = red[(2 * RT + r) * 64 + lane],
                 d3 = red[(3 * RT + r) * 64 + lane];
    const float dot[2] = {d0.x + d1.x + d2.x + d3.x, d0.y + d1.y + d2.y + d3.y};
    const float pr = pos[r] * inv_T;
    float l[2], cnt = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      l[j] = cl.live ? dot[j] * cl.decay[j] * inv_T : -INFINITY;
      cnt += (cl.live && l[j] > pr) ? 1.f : 0.f;
    }
    const float m = wave_max(fmaxf(l[0], l[1]));
    const float s = wave_sum(cl.live ? __expf(l[0] - m) + __expf(l[1] - m) : 0.f);
    const float c = wave_sum(cnt);
    if (lane == 0) { float* o = part + ((long)blockIdx.x * R + r) * 3; o[0] = m; o[1] = s; o[2] = c; }
  }
}

// ---- Round 6: the feature x queue product on the matrix cores (exact fp32: v_mfma_f32_16x16x4_f32 == an fmaf chain) ----------------
// The vector form above splits the 128 channels over four waves: every lane does R x 32 multiply-adds per column, fetches the query
// rows from LDS for each channel, and the four partial sums of every (row, column) meet through LDS -- loads, 3072 multiply-adds and
// the exchange follow one another inside a block (17 / 25 us per pass at 24 rows = 1.9 / 1.3 TB/s of queue reads).  Here a WAVE owns
// 64 queue columns over ALL channels, so a logit is born complete in one accumulator:
//   D[queue column m][query row n] += A[m][c] * B[c][n],  A = the queue tile as it arrives from memory, B = the query rows.
// * A operand = the load itself.  Load g of a lane is 16 bytes: channel 32 (lane >> 4) + g, columns k0 + 4 (lane & 15) .. + 3 -- a wave
//   instruction reads 256 contiguous bytes of each of four queue rows -- and register j of that load is the A operand (row m =
//   lane & 15, k = lane >> 4) of the MFMA whose row m stands for column k0 + 4 m + j: the MFMA does not care which column a row is,
//   so no data moves between the load and the matrix core.  All 32 loads of a wave (32 KB) are issued before the first use.
// * B operand = q[16 h + (lane & 15)][32 (lane >> 4) + g], 32 registers per 16-row half = eight 16-byte loads per wave; no LDS.
// * The accumulator of MFMA j holds, in register v, column k0 + 16 (lane >> 4) + 4 v + j for query row lane & 15: a lane ends up
//   with SIXTEEN CONSECUTIVE columns of ONE query row, so max / sum-exp / rank count are in-lane loops; the four lane groups of a
//   row and the two waves of a 128-column chunk meet through 6 KB of LDS in a fixed order (the `part` layout of the ABI is kept).
// * Ages: lane l turns count[k0 + l] into its decay once; the sixteen a lane needs come back from LDS as four 16-byte reads.
// One block = 4 waves = 256 columns: 256 blocks at K = 65536, one per CU, 128 KB of loads in flight per CU.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define NCE_MCOLS 64                      // queue columns per wave of the MFMA kernels (two waves = one 128-column chunk of `part`)

struct NceTile { int k0, kc; bool live4; };            // kc: the lane's first column (clamped), live4: its four columns exist
// (no zero-fill here: rows >= R are clamped to row R - 1 and never stored; channels >= dim meet zeroed queue registers)
template <int RH>
__device__ __forceinline__ void nce_mfma_load_q(const float* __restrict__ q, float (&bq)[RH][32], int R, int dim, int lane) {
  const int lm = lane & 15, lg = lane >> 4;
  const int cq = dim >> 2, imax = (cq >> 2) - 1;          // lane group lg contracts channels [lg cq, (lg + 1) cq): contiguous in a query row
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    const int r = 16 * h + lm < R ? 16 * h + lm : R - 1;
    const float* qp = q + r * dim + lg * cq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(qp + 4 * (i < imax ? i : imax));
      bq[h][4 * i] = v[0]; bq[h][4 * i + 1] = v[1]; bq[h][4 * i + 2] = v[2]; bq[h][4 * i + 3] = v[3];
    }
  }
}
__device__ __forceinline__ void nce_mfma_load_tile(const float* __restrict__ queue, f32x4_t (&wv)[32], int dim, int K, int lane,
                                                   const NceTile& t, const NceVirt& vt, int p) {
  // one buffer descriptor over the queue; a lane's byte offset (its channel of the group, its first column) in a VGPR, the group's
  // row offset 4 g K in the instruction's SGPR operand: no 64-bit address arithmetic per load (the launcher checks dim K 4 < 2^31)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int lg = lane >> 4;
  const uint64_t qa = reinterpret_cast<uint64_t>(queue);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
  const int cq = dim >> 2, gmax = cq - 1;  // MFMA step g contracts channels lg cq + g (lg = 0 .. 3): step g exists for every lane group or for none
  const int voff = (lg * cq * K + t.kc) * 4;
#pragma unroll
  for (int g = 0; g < 32; ++g) {           // unconditional loads from clamped rows / columns
    const int gs = g < gmax ? g : gmax;
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, gs * 4 * K, 0);
    wv[g] = __builtin_bit_cast(f32x4_t, v);
  }
  if (vt.keys != nullptr) {                // the virtual enqueue's columns (wave-uniform test first: one tile in K / 64 has any)
    if (p < t.k0 + NCE_MCOLS && p + vt.n_new > t.k0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int jn = t.kc + j - p;
        if (jn >= 0 && jn < vt.n_new) {
#pragma unroll
          for (int g = 0; g < 32; ++g) wv[g][j] = vt.keys[(long)jn * dim + lg * cq + (g < gmax ? g : gmax)];
        }
      }
    }
  }
  if (gmax < 31) {                         // dim < 128 (wave-uniform); dead columns need no zeros: the epilogue masks them by column
#pragma unroll
    for (int g = 0; g < 32; ++g)
      if (g > gmax) wv[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
}
// decay of a column at age `a` (recognizers/moco.py:484) in the snapshot `vt` describes, as nce_decay takes it (p = *vt.ptr)
__device__ __forceinline__ float nce_mfma_decay(long a, int k, int p, const NceVirt& vt) {
  const int jn = k - p;
  const float age = vt.keys == nullptr ? (float)a : ((jn >= 0 && jn < vt.n_new) ? 1.f : (float)(a + 1));
  return powf(0.99999f, age);
}
template <int RH>
__device__ __forceinline__ void nce_mfma_logits(const f32x4_t (&wv)[32], const float (&bq)[RH][32], f32x4_t (&acc)[RH][4]) {
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[h][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 32; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < RH; ++h) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[g][j], bq[h][g], acc[h][j], 0, 0, 0);
}

template <int RH>
__global__ __launch_bounds__(256, 1) void nce_fwd_mfma_kernel(const float* __restrict__ queue, const int64_t* __restrict__ count,
                                                              const float* __restrict__ q, const float* __restrict__ pos,
                                                              float* __restrict__ part, int R, int dim, int K, float inv_T,
                                                              const NceVirt vt) {
  __shared__ __attribute__((aligned(16))) float dec_s[4][NCE_MCOLS];
  __shared__ float pm[4][4][16 * RH][3];                         // [wave][lane group][row]{max, sum, count}
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lm = lane & 15, lg = lane >> 4;
  NceTile t; t.k0 = (blockIdx.x * 4 + wave) * NCE_MCOLS;
  t.live4 = t.k0 + 4 * lm < K; t.kc = t.live4 ? t.k0 + 4 * lm : K - 4;
  // the small loads first (queue pointer, the lane's age, the query rows: ~20 instructions, nothing waits for them yet), then the
  // tile; the pointer is needed only behind the tile's loads, by the test for the virtual enqueue's columns
  const int pq = vt.keys != nullptr ? (int)*vt.ptr : 0;
  const int kd = t.k0 + lane < K ? t.k0 + lane : K - 1;
  const long age = count[kd];
  float bq[RH][32];
  nce_mfma_load_q<RH>(q, bq, R, dim, lane);
  __builtin_amdgcn_sched_barrier(0);        // (left alone, hipcc issues the query-row loads in the MIDDLE of the tile's: the first MFMA
  f32x4_t wv[32];                           //  then waits for 37 of the 48 loads and the chain starts when the tile is nearly complete)
  nce_mfma_load_tile(queue, wv, dim, K, lane, t, vt, pq);
  dec_s[wave][lane] = nce_mfma_decay(age, kd, pq, vt);
  f32x4_t acc[RH][4];
  nce_mfma_logits<RH>(wv, bq, acc);
  float dc[16];                                                  // the lane's sixteen columns: k0 + 16 lg + 4 v + j  ->  dc[4 v + j]
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(&dec_s[wave][16 * lg + 4 * v]);
    dc[4 * v] = d4[0]; dc[4 * v + 1] = d4[1]; dc[4 * v + 2] = d4[2]; dc[4 * v + 3] = d4[3];
  }
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    const int r = 16 * h + lm;
    const float pr = pos[r < R ? r : 0] * inv_T;
    float l[16], m = -INFINITY, cnt = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = t.k0 + 16 * lg + 4 * v + j < K;
        const float x = live ? acc[h][j][v] * dc[4 * v + j] * inv_T : -INFINITY;
        l[4 * v + j] = x; m = fmaxf(m, x); cnt += (live && x > pr) ? 1.f : 0.f;
      }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (m == -INFINITY) ? 0.f : __expf(l[i] - m);
    float* o = pm[wave][lg][r];
    o[0] = m; o[1] = sum; o[2] = cnt;
  }
  __syncthreads();
  // one partial per 128-column chunk (= two waves x four lane groups), merged in a fixed order
  if (threadIdx.x < 2 * 16 * RH) {
    const int ch = threadIdx.x / (16 * RH), r = threadIdx.x - ch * 16 * RH;
    const int blk = blockIdx.x * 2 + ch;
    if (r < R && blk * NCE_BCOLS < K) {
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) M = fmaxf(M, pm[2 * ch + w][g][r][0]);
      float S = 0.f, Cn = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* o = pm[2 * ch + w][g][r];
          S += (o[0] == -INFINITY) ? 0.f : o[1] * __expf(o[0] - M);
          Cn += o[2];
        }
      float* o = part + ((long)blk * R + r) * 3; o[0] = M; o[1] = S; o[2] = Cn;
    }
  }
}

// Backward on the matrix cores.  Phase 1 = the forward's logits, turned in place into the coefficients cf[r][column] = softmax * decay *
// scale.  Phase 2 = dq[r][c] += sum over the tile's columns of cf[r][column] * queue[c][column]: the contraction index is now the COLUMN,
// which phase 1 had along the lanes -- the one transposition this pass needs.  The tile goes through LDS once (each load is written as
// it arrives, under phase 1's MFMAs) in the image Wt[column group of 16][channel][16 columns], 16-byte units XOR-swizzled by
// (channel >> 2) & 3, and comes back as the B operand of D2[r][c] += A2[r][kk] B2[kk][c] with K-step t <-> columns 16 kk + t:
//   A2 (row r = lane & 15, kk = lane >> 4) = cf[r][16 kk + t] -- exactly the accumulator register (j, v) with 4 v + j = t the lane
//   already holds: no movement;  B2 (kk, channel 16 cb + (lane & 15)) = one 16-byte LDS read per four K-steps, conflict-free.
// The four waves' [rows x dim] results are added in wave order through the dead tile image and leave as ONE slab per block
// ([block][rt][dim], summed by nce_bwd_reduce_kernel as before; half as many slabs as the vector form wrote).
#define NCE_WT_FLOATS (4 * 128 * 16)       // one wave's tile image: [4 column groups][128 channels][16 columns] = 32 KB
template <int RH>
__global__ __launch_bounds__(256, 1) void nce_bwd_mfma_kernel(const float* __restrict__ queue, const int64_t* __restrict__ count,
                                                              const float* __restrict__ q, const float* __restrict__ lse,
                                                              const float* __restrict__ row_scale, float* __restrict__ slab,
                                                              int R, int rt, int dim, int K, float inv_T, const NceVirt vt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // Wt[4 waves][NCE_WT_FLOATS] | dec_s[4][64]
  float* dec_s = sm + 4 * NCE_WT_FLOATS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lm = lane & 15, lg = lane >> 4;
  float* Wt = sm + wave * NCE_WT_FLOATS;
  NceTile t; t.k0 = (blockIdx.x * 4 + wave) * NCE_MCOLS;
  t.live4 = t.k0 + 4 * lm < K; t.kc = t.live4 ? t.k0 + 4 * lm : K - 4;
  const int pq = vt.keys != nullptr ? (int)*vt.ptr : 0;
  const int kd = t.k0 + lane < K ? t.k0 + lane : K - 1;
  const long age = count[kd];
  float ls[RH], sc[RH]; bool rok[RH];
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    const int r = 16 * h + lm;
    rok[h] = r < R;
    ls[h] = lse[rok[h] ? r : 0]; sc[h] = inv_T * row_scale[rok[h] ? r : 0];
  }
  float bq[RH][32];
  nce_mfma_load_q<RH>(q, bq, R, dim, lane);
  __builtin_amdgcn_sched_barrier(0);
  f32x4_t wv[32];
  nce_mfma_load_tile(queue, wv, dim, K, lane, t, vt, pq);
  dec_s[wave * NCE_MCOLS + lane] = nce_mfma_decay(age, kd, pq, vt);
  // phase 1, each load also stored into the tile image: the lane's four columns are unit (lm & 3) of column group lm >> 2
  const int cq = dim >> 2;
  f32x4_t acc[RH][4];
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[h][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 32; ++g) {
    const int ch = lg * cq + g;                             // (g >= cq: zeros, stored over rows that phase 2 never reads as real channels)
    if (g < cq)
      *reinterpret_cast<f32x4_t*>(Wt + (((lm >> 2) * 128 + (ch & 127)) * 16 + 4 * ((lm & 3) ^ ((ch >> 2) & 3)))) = wv[g];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < RH; ++h) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[g][j], bq[h][g], acc[h][j], 0, 0, 0);
  }
  float dc[16];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(&dec_s[wave * NCE_MCOLS + 16 * lg + 4 * v]);
    dc[4 * v] = d4[0]; dc[4 * v + 1] = d4[1]; dc[4 * v + 2] = d4[2]; dc[4 * v + 3] = d4[3];
  }
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = t.k0 + 16 * lg + 4 * v + j < K;
        const float d = dc[4 * v + j];
        acc[h][j][v] = (live && rok[h]) ? __expf(acc[h][j][v] * d * inv_T - ls[h]) * d * sc[h] : 0.f;
      }
  // phase 2 (a wave reads only the image it wrote: LDS operations of one wave complete in order)
  f32x4_t d2[RH][8];
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) d2[h][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float* wr = Wt + (lg * 128 + lm) * 16;
  const int sw = (lm >> 2) & 3;
#pragma unroll
  for (int cp = 0; cp < 4; ++cp) {                          // channel blocks in pairs: two independent accumulator chains per row half
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(wr + (2 * cp) * 256 + 4 * (i ^ sw));
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(wr + (2 * cp + 1) * 256 + 4 * (i ^ sw));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int h = 0; h < RH; ++h) {
          d2[h][2 * cp] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[h][jj][i], b0[jj], d2[h][2 * cp], 0, 0, 0);
          d2[h][2 * cp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[h][jj][i], b1[jj], d2[h][2 * cp + 1], 0, 0, 0);
        }
    }
  }
  __syncthreads();                                          // every wave is done with its image: the region changes hands
  float* red = sm;                                          // [wave][16 RH rows][128 channels]
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[(wave * 16 * RH + 16 * h + 4 * lg + v) * 128 + 16 * cb + lm] = d2[h][cb][v];
  __syncthreads();
  float* o = slab + (long)blockIdx.x * rt * dim;
  for (int e = threadIdx.x; e < rt * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    if (c < dim && r < 16 * RH)
      o[r * dim + c] = ((red[r * 128 + c] + red[(16 * RH + r) * 128 + c]) + red[(32 * RH + r) * 128 + c]) + red[(48 * RH + r) * 128 + c];
  }
}

// one block per row: merge the per-block partials.  256 threads take nblk / 256 partials each with every load in flight, then
// meet through LDS (one wave looping over 1024 partials twice was 32 dependent round trips: 8 us, three times in the loss phase)
__global__ __launch_bounds__(256) void nce_finish_kernel(const float* __restrict__ part, const float* __restrict__ pos,
                                                         float* __restrict__ lse, float* __restrict__ loss_rows,
                                                         int32_t* __restrict__ rank, int R, int nblk, float inv_T) {
  __shared__ float red[3][4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float p = pos[r] * inv_T;
  constexpr int U = 8;
  float m = -INFINITY, s = 0.f, c = 0.f;
  for (int b0 = tid; b0 < nblk; b0 += 256 * U) {
    float pm[U], ps[U], pc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u * 256;
      const float* o = part + ((long)(b < nblk ? b : b0) * R + r) * 3;
      pm[u] = b < nblk ? o[0] : -INFINITY; ps[u] = b < nblk ? o[1] : 0.f; pc[u] = b < nblk ? o[2] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float mn = fmaxf(m, pm[u]);
      s = (mn == -INFINITY) ? 0.f : s * expf(m - mn) + ps[u] * expf(pm[u] - mn);
      m = mn; c += pc[u];
    }
  }
  // block merge: the maximum first, then the sums rescaled to it
  const float wm = wave_max(m);
  if (lane == 0) red[0][wave] = wm;
  __syncthreads();
  const float M = fmaxf(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])), p);
  const float sw = wave_sum(m == -INFINITY ? 0.f : s * expf(m - M)), cw = wave_sum(c);
  if (lane == 0) { red[1][wave] = sw; red[2][wave] = cw; }
  __syncthreads();
  if (tid == 0) {
    const float S = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + expf(p - M);
    const float C = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const float l = M + logf(S);
    lse[r] = l; loss_rows[r] = l - p; rank[r] = (int32_t)(C + 0.5f);
  }
}

// dq[r][c] += inv_T * row_scale[r] * sum_k softmax_k * decay_k * queue[c][k]
// A block owns 128 columns.  Phase 1 = the forward's logits (partial sums through LDS) -> the softmax coefficients gc[r][column].
// Phase 2 = the block's share of dq as a [RT x 128] x [128 x dim] product on the vector units: the queue tile is still in the
// registers that loaded it and is laid out by channel, Wt[channel][128 columns], over the dead reduction image; a thread owns FOUR
// channels x RT / 8 rows, so a 4-column step costs 4 + RT / 8 sixteen-byte LDS reads for 2 RT multiply-adds (one channel x RT / 2
// rows, the round-1 form, cost 1 + RT / 2: the phase was LDS-bound).
// The per-block results go to a slab ([block][RT][dim], plain stores) summed by nce_bwd_reduce_kernel.  (One atomicAdd per
// element per block meant millions of float atomics on the same 12 KB: contention-bound at 67 us for a 33.5-MB read.)
#define NCE_WPAD 132         // Wt row pitch in floats: 16-byte aligned rows, conflict-free b128 reads down 16 consecutive channels
template <int RT>
__global__ __launch_bounds__(256, 2) void nce_bwd_kernel(const float* __restrict__ queue, const int64_t* __restrict__ count,
                                                         const float* __restrict__ q, const float* __restrict__ lse,
                                                         const float* __restrict__ row_scale, float* __restrict__ slab,
                                                         int R, int dim, int K, float inv_T, const NceVirt vt) {
  // LDS: [ qs[dim][RT], later gc[RT][128] ] [ red[4][RT][64] float2, later Wt[dim][NCE_WPAD] ] -- the coefficients are written when
  // the last reader of the query rows has passed the barrier behind phase 1
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm; float* gc = sm;
  float2* red = reinterpret_cast<float2*>(sm + RT * NCE_BCOLS);
  float* Wt = reinterpret_cast<float*>(red);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: row bases in SGPRs)
  const int c0 = wave * 32;
  NceCols cl = nce_cols(blockIdx.x, lane, K, vt);
  float qr[RT / 2];
  nce_q_load<RT>(q, qr, R, dim);
  float2 wv[32];
  nce_load_cols(queue, wv, c0, dim, K, cl, vt);
  nce_decay(cl, count, vt);
  nce_q_store<RT>(qr, qs);
  __syncthreads();
  {
    float2 acc[RT];
    nce_partial<RT>(qs, wv, c0, acc);
#pragma unroll
    for (int r = 0; r < RT; ++r) red[(wave * RT + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  float2 cf[RT / NCE_WAVES];
#pragma unroll
  for (int n = 0; n < RT / NCE_WAVES; ++n) {
    const int r = wave + n * NCE_WAVES;
    const float2 d0 = red[(0 * RT + r) * 64 + lane], d1 = red[(1 * RT + r) * 64 + lane], d2 = red[(2 * RT + r) * 64 + lane],
                 d3 = red[(3 * RT + r) * 64 + lane];
    const float dot[2] = {d0.x + d1.x + d2.x + d3.x, d0.y + d1.y + d2.y + d3.y};
    const bool rl = cl.live && r < R;
    const float ls = lse[r < R ? r : 0], sc = inv_T * row_scale[r < R ? r : 0];
    cf[n].x = rl ? __expf(dot[0] * cl.decay[0] * inv_T - ls) * cl.decay[0] * sc : 0.f;
    cf[n].y = rl ? __expf(dot[1] * cl.decay[1] * inv_T - ls) * cl.decay[1] * sc : 0.f;
  }
  __syncthreads();                                // every wave has read red (and qs long before): both regions change hands
#pragma unroll
  for (int n = 0; n < RT / NCE_WAVES; ++n)
    *reinterpret_cast<float2*>(gc + (wave + n * NCE_WAVES) * NCE_BCOLS + lane * 2) = cf[n];
#pragma unroll
  for (int i = 0; i < 32; ++i)                    // (all 128 rows: channels >= dim hold zeros)
    *reinterpret_cast<float2*>(Wt + (c0 + i) * NCE_WPAD + lane * 2) = cl.live ? wv[i] : make_float2(0.f, 0.f);
  __syncthreads();
  // phase 2: thread -> channels cg + 32 u (u < 4), rows RG rg .. RG rg + RG - 1
  constexpr int RG = RT / 8;
  const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
  float a2[RG][4];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int u = 0; u < 4; ++u) a2[r][u] = 0.f;
#pragma unroll 2
  for (int j = 0; j < NCE_BCOLS; j += 4) {
    float4 w[4], gv[RG];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(Wt + (cg + 32 * u) * NCE_WPAD + j);
#pragma unroll
    for (int r = 0; r < RG; ++r) gv[r] = *reinterpret_cast<const float4*>(gc + (rg * RG + r) * NCE_BCOLS + j);
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        a2[r][u] = fmaf(gv[r].x, w[u].x, fmaf(gv[r].y, w[u].y, fmaf(gv[r].z, w[u].z, fmaf(gv[r].w, w[u].w, a2[r][u]))));
  }
  float* o = slab + (long)blockIdx.x * RT * dim;
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (cg + 32 * u < dim) o[(rg * RG + r) * dim + cg + 32 * u] = a2[r][u];
}

// dq[r][c] += sum over blocks of slab[b][r][c]  (rows r < R); blockIdx.y takes every gridDim.y-th slab.
// kpos != NULL: the positive pair's term rides along (one launch less per pass on the step's serial loss phase) --
// dq[r][:] += row_scale[r] * inv_T * (softmax_pos[r] - 1) * kpos[r][:], added by the blocks of slab column 0 as one more addend
// BEHIND the slab sum: the arithmetic of nce_pos_bwd_kernel, and in deterministic mode still one add per element.
__global__ __launch_bounds__(256) void nce_bwd_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dq, int nslab,
                                                             int RT, int R, int dim, const float* __restrict__ kpos,
                                                             const float* __restrict__ pos, const float* __restrict__ lse,
                                                             const float* __restrict__ row_scale, float inv_T) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= R * dim) return;
  const int step = gridDim.y;
  int b = blockIdx.y;
  float pterm = 0.f;
  if (kpos != nullptr && b == 0) {
    const int r = e / dim;
    pterm = row_scale[r] * inv_T * (__expf(pos[r] * inv_T - lse[r]) - 1.f) * kpos[e];
  }
  if (step == 1) {            // deterministic mode: ONE add per element of a sum taken in slab order, 16 loads in flight per trip
    float s = 0.f;
    for (; b + 16 <= nslab; b += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = slab[(long)(b + u) * RT * dim + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; b < nslab; ++b) s += slab[(long)b * RT * dim + e];
    atomicAdd(&dq[e], s + pterm);
    return;
  }
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  for (; b + 3 * step < nslab; b += 4 * step) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += slab[(long)(b + u * step) * RT * dim + e];
  }
  for (; b < nslab; b += step) s4[0] += slab[(long)b * RT * dim + e];
  atomicAdd(&dq[e], ((s4[0] + s4[1]) + (s4[2] + s4[3])) + pterm);
}

#define NCE_DISPATCH(RV, CALL8, CALL16, CALL24, CALL32) \
  if ((RV) <= 8) { CALL8; } else if ((RV) <= 16) { CALL16; } else if ((RV) <= 24) { CALL24; } else { CALL32; }

// Row tiles: the kernels hold up to 32 query rows per block in registers; more rows (3 * B stacked row groups with a per-GPU
// batch above 10, e.g. the shipped config's videos_per_gpu = 32) run as consecutive 32-row tiles over the same snapshot, each
// with its own slice of `part` ([tile][blk][rows of the tile][3], tiles before the last are full) -- no limit at the ABI.
#define NCE_ROW_TILE 32
extern "C" int mscl_nce_fwd(const float* queue, const int64_t* count, const float* q, const float* pos_logit, float* part,
                            int R, int dim, int K, float inv_T, void* stream) {
  return mscl_nce_fwd_virt(queue, count, q, pos_logit, part, R, dim, K, inv_T, nullptr, 0, nullptr, stream);
}
extern "C" int mscl_nce_fwd_virt(const float* queue, const int64_t* count, const float* q, const float* pos_logit, float* part,
                                 int R, int dim, int K, float inv_T, const float* new_keys, int n_new, const int64_t* queue_ptr,
                                 void* stream) {
  if (!queue || !count || !q || !pos_logit || !part || R <= 0 || dim <= 0 || K <= 0) return MSCL_E_ARG;
  if (new_keys && (!queue_ptr || n_new <= 0 || n_new > K)) return MSCL_E_ARG;
  const NceVirt vt{new_keys, queue_ptr, new_keys ? n_new : 0};
  if (dim > 128 || dim % NCE_WAVES || K % 2) return MSCL_E_SHAPE;       // (K % 2: a lane reads two columns as one 8-byte piece)
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (K + NCE_BCOLS - 1) / NCE_BCOLS;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_fwd_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  // (K % 4, dim % 16: a lane reads four columns of a queue row and four channels of a query row as 16-byte pieces; 32-bit byte
  // offsets; every other shape stays with the vector kernels.  The round-6 A/B of the two inside the step is profiles/r06_ab_step.txt)
  const bool mfma = K % 4 == 0 && dim % 16 == 0 && (long)dim * K * 4 < (1L << 31);
  for (int r0 = 0; r0 < R; r0 += NCE_ROW_TILE) {
    const int Rt = R - r0 < NCE_ROW_TILE ? R - r0 : NCE_ROW_TILE;
    const int rt = Rt <= 8 ? 8 : (Rt <= 16 ? 16 : (Rt <= 24 ? 24 : 32));
    const size_t lds = ((size_t)128 * rt + (size_t)NCE_WAVES * rt * NCE_BCOLS) * sizeof(float);        // 61 KB at 24 rows: two blocks per CU
    const float* qt = q + (size_t)r0 * dim; const float* pt = pos_logit + r0;
    float* part_t = part + (size_t)nblk * r0 * 3;
    if (mfma) {
      const int nb = (K + 4 * NCE_MCOLS - 1) / (4 * NCE_MCOLS);
      if (Rt <= 16) hipLaunchKernelGGL(nce_fwd_mfma_kernel<1>, dim3(nb), dim3(256), 0, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt);
      else hipLaunchKernelGGL(nce_fwd_mfma_kernel<2>, dim3(nb), dim3(256), 0, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt);
      MSCL_LAUNCH_CHECK();
      continue;
    }
    NCE_DISPATCH(Rt,
      hipLaunchKernelGGL(nce_fwd_kernel<8>, dim3(nblk), dim3(256), lds, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_fwd_kernel<16>, dim3(nblk), dim3(256), lds, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_fwd_kernel<24>, dim3(nblk), dim3(256), lds, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_fwd_kernel<32>, dim3(nblk), dim3(256), lds, st, queue, count, qt, pt, part_t, Rt, dim, K, inv_T, vt))
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mscl_nce_finish(const float* part, const float* pos_logit, float* lse, float* loss_rows, int32_t* rank, int R,
                               int nblk, float inv_T, void* stream) {
  if (!part || !pos_logit || !lse || !loss_rows || !rank || R <= 0 || nblk <= 0) return MSCL_E_ARG;
  for (int r0 = 0; r0 < R; r0 += NCE_ROW_TILE) {
    const int Rt = R - r0 < NCE_ROW_TILE ? R - r0 : NCE_ROW_TILE;
    hipLaunchKernelGGL(nce_finish_kernel, dim3(Rt), dim3(256), 0, (hipStream_t)stream, part + (size_t)nblk * r0 * 3, pos_logit + r0,
                       lse + r0, loss_rows + r0, rank + r0, Rt, nblk, inv_T);
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mscl_nce_bwd(const float* queue, const int64_t* count, const float* q, const float* lse, const float* row_scale,
                            float* dq, float* ws, int64_t ws_floats, int R, int dim, int K, float inv_T, void* stream) {
  return mscl_nce_bwd_virt(queue, count, q, lse, row_scale, dq, ws, ws_floats, R, dim, K, inv_T, nullptr, 0, nullptr, nullptr, nullptr, stream);
}
extern "C" int mscl_nce_bwd_virt(const float* queue, const int64_t* count, const float* q, const float* lse, const float* row_scale,
                                 float* dq, float* ws, int64_t ws_floats, int R, int dim, int K, float inv_T,
                                 const float* new_keys, int n_new, const int64_t* queue_ptr, const float* kpos, const float* pos_logit,
                                 void* stream) {
  if (!queue || !count || !q || !lse || !row_scale || !dq || !ws || R <= 0 || dim <= 0 || K <= 0) return MSCL_E_ARG;
  if ((kpos == nullptr) != (pos_logit == nullptr)) return MSCL_E_ARG;
  if (new_keys && (!queue_ptr || n_new <= 0 || n_new > K)) return MSCL_E_ARG;
  const NceVirt vt{new_keys, queue_ptr, new_keys ? n_new : 0};
  if (dim > 128 || dim % NCE_WAVES || K % 2) return MSCL_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (K + NCE_BCOLS - 1) / NCE_BCOLS;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const bool mfma = K % 4 == 0 && dim % 16 == 0 && (long)dim * K * 4 < (1L << 31);
  static bool attr_m = false;
  if (mfma && !attr_m) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_bwd_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_m = true;
  }
  for (int r0 = 0; r0 < R; r0 += NCE_ROW_TILE) {      // the tiles share the slab workspace: they run one after the other on `st`
    const int Rt = R - r0 < NCE_ROW_TILE ? R - r0 : NCE_ROW_TILE;
    const int rt = Rt <= 8 ? 8 : (Rt <= 16 ? 16 : (Rt <= 24 ? 24 : 32));
    if ((int64_t)nblk * rt * dim > ws_floats) return MSCL_E_ARG;
    if (mfma) {
      const int nb = (K + 4 * NCE_MCOLS - 1) / (4 * NCE_MCOLS);            // one slab per block of 256 columns: nb <= nblk
      const size_t ldsm = ((size_t)4 * NCE_WT_FLOATS + 4 * NCE_MCOLS) * sizeof(float);
      const float* qt = q + (size_t)r0 * dim; const float* lt = lse + r0; const float* st_ = row_scale + r0;
      if (Rt <= 16) hipLaunchKernelGGL(nce_bwd_mfma_kernel<1>, dim3(nb), dim3(256), ldsm, st, queue, count, qt, lt, st_, ws, Rt, rt, dim, K, inv_T, vt);
      else hipLaunchKernelGGL(nce_bwd_mfma_kernel<2>, dim3(nb), dim3(256), ldsm, st, queue, count, qt, lt, st_, ws, Rt, rt, dim, K, inv_T, vt);
      MSCL_LAUNCH_CHECK();
      // (32 slab columns: swept again for the 256 slabs of this form -- 4 / 8 / 16 / 32 / 64: 26.7 / 24.3 / 23.0 / 22.8 / 23.6 us per call pair at 24 rows)
      hipLaunchKernelGGL(nce_bwd_reduce_kernel, dim3((Rt * dim + 255) / 256, mscl_det() ? 1 : 32), dim3(256), 0, st, (const float*)ws,
                         dq + (size_t)r0 * dim, nb, rt, Rt, dim, kpos ? kpos + (size_t)r0 * dim : nullptr, kpos ? pos_logit + r0 : nullptr, lt, st_,
                         inv_T);
      MSCL_LAUNCH_CHECK();
      continue;
    }
    const size_t red_f = (size_t)NCE_WAVES * rt * NCE_BCOLS, wt_f = (size_t)128 * NCE_WPAD;      // Wt lies over the dead reduction image
    const size_t lds = ((size_t)rt * NCE_BCOLS + (red_f > wt_f ? red_f : wt_f)) * sizeof(float);          // (dim <= 128 = NCE_BCOLS: gc covers qs); 79 KB at 24 rows
    const float* qt = q + (size_t)r0 * dim; const float* lt = lse + r0; const float* st_ = row_scale + r0;
    NCE_DISPATCH(Rt,
      hipLaunchKernelGGL(nce_bwd_kernel<8>, dim3(nblk), dim3(256), lds, st, queue, count, qt, lt, st_, ws, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_bwd_kernel<16>, dim3(nblk), dim3(256), lds, st, queue, count, qt, lt, st_, ws, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_bwd_kernel<24>, dim3(nblk), dim3(256), lds, st, queue, count, qt, lt, st_, ws, Rt, dim, K, inv_T, vt),
      hipLaunchKernelGGL(nce_bwd_kernel<32>, dim3(nblk), dim3(256), lds, st, queue, count, qt, lt, st_, ws, Rt, dim, K, inv_T, vt))
    MSCL_LAUNCH_CHECK();
    // (deterministic mode: one block column, so every element receives ONE add of a sum taken in slab order)
    hipLaunchKernelGGL(nce_bwd_reduce_kernel, dim3((Rt * dim + 255) / 256, mscl_det() ? 1 : 32), dim3(256), 0, st, (const float*)ws,
                       dq + (size_t)r0 * dim, nblk, rt, Rt, dim, kpos ? kpos + (size_t)r0 * dim : nullptr, kpos ? pos_logit + r0 : nullptr, lt, st_,
                       inv_T);
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------- the step's log vector in one launch
// mean loss / top-1 / top-5 per InfoNCE group (heads/moco_head.py:60-77: accuracy = share of rows whose positive is
// ranked < k), the LMCL entries, and their sum (recognizers/base.py:297-298), in MSCLWithAug's key order.  Replaces ~50
// compare / cast / mean / stack micro-kernels per step.
// groups: A = RGB queue pass (rows [q_rgb | q_flow | q_flow_rot]), B = flow queue pass, C = flow queue pass after the
// base-flow enqueue (rows [q_flow_rot | q_rgb | q_rgb]); nA, nC in {2, 3} (3 = cross-modal terms of the rotated flow).
__global__ __launch_bounds__(64) void mscl_logs_kernel(const int32_t* __restrict__ rankA, const float* __restrict__ lossA,
                                                       const int32_t* __restrict__ rankB, const float* __restrict__ lossB,
                                                       const int32_t* __restrict__ rankC, const float* __restrict__ lossC,
                                                       const float* __restrict__ lmcl_sum, const int32_t* __restrict__ lmcl_hits,
                                                       int B, int nA, int nC, float w_intra, float n_rows, float* __restrict__ logs) {
  __shared__ float g3[7][3];                       // [group][top1, top5, loss]: A0 A1 A2 B C0 C1 C2
  const int lane = threadIdx.x;
  for (int grp = 0; grp < 7; ++grp) {
    const int32_t* rk; const float* ls; int idx;
    if (grp < 3) { rk = rankA; ls = lossA; idx = grp; if (idx >= nA) continue; }
    else if (grp == 3) { rk = rankB; ls = lossB; idx = 0; }
    else { rk = rankC; ls = lossC; idx = grp - 4; if (idx >= nC) continue; }
    float t1 = 0.f, t5 = 0.f, l = 0.f;
    for (int r = lane; r < B; r += 64) {
      const int rr = rk[idx * B + r];
      t1 += rr < 1 ? 1.f : 0.f; t5 += rr < 5 ? 1.f : 0.f; l += ls[idx * B + r];
    }
    t1 = wave_sum(t1); t5 = wave_sum(t5); l = wave_sum(l);
    if (lane == 0) { g3[grp][0] = t1 / (float)B; g3[grp][1] = t5 / (float)B; g3[grp][2] = l / (float)B; }
  }
  __syncthreads();
  if (lane == 0) {
    int o = 0;
    float total = 0.f;
    auto put3 = [&](int grp) { logs[o++] = g3[grp][0]; logs[o++] = g3[grp][1]; logs[o++] = g3[grp][2]; total += g3[grp][2]; };
    put3(0);                                              // top1_acc, top5_acc, loss_cls
    put3(3);                                              // *_flow
    { const float v = g3[4][2] * w_intra; logs[o++] = v; total += v; }      // loss_cls_flow_aug
    put3(5);                                              // *_mx        (q_rgb vs flow queue)
    put3(1);                                              // *_mx_r      (q_flow vs rgb queue)
    if (nA == 3 && nC == 3) { put3(6); put3(2); }         // *_mx_aug, *_mx_r_aug
    const float lp = lmcl_sum[0] / n_rows;
    logs[o++] = lp; total += lp;                          // loss_pos
    logs[o++] = (float)lmcl_hits[0] / n_rows;             // top1_acc_pos
    logs[o++] = (float)lmcl_hits[1] / n_rows;             // top5_acc_pos
    logs[o++] = total;                                    // loss
  }
}
extern "C" int mscl_step_logs(const int32_t* rankA, const float* lossA, const int32_t* rankB, const float* lossB,
                              const int32_t* rankC, const float* lossC, const float* lmcl_sum, const int32_t* lmcl_hits, int B,
                              int nA, int nC, float w_intra, float n_rows, float* logs, void* stream) {
  if (!rankA || !lossA || !rankB || !lossB || !rankC || !lossC || !lmcl_sum || !lmcl_hits || !logs || B <= 0) return MSCL_E_ARG;
  if (nA < 2 || nA > 3 || nC < 2 || nC > 3 || nA != nC) return MSCL_E_SHAPE;
  hipLaunchKernelGGL(mscl_logs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rankA, lossA, rankB, lossB, rankC, lossC, lmcl_sum,
                     lmcl_hits, B, nA, nC, w_intra, n_rows, logs);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- queue bookkeeping (int64, bit-exact)
// One launch: every block reads the pointer before it counts itself done, so the block that counts LAST may move it (the second,
// one-thread launch that did this sat on the step's serial loss phase twice).  Library-owned ticket: two enqueues must not run side
// by side (they are ordered on one stream in the step, as the reference orders them).
__device__ unsigned g_enq_ticket;
__global__ __launch_bounds__(256) void enqueue_kernel(float* __restrict__ queue, int64_t* __restrict__ count,
                                                      int64_t* __restrict__ ptr, const float* __restrict__ keys, int n,
                                                      int dim, int K) {
  const int64_t p = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long total = (long)K + (long)n * dim;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (e < K) {
      count[e] = (e >= p && e < p + n) ? (int64_t)1 : count[e] + 1;
    } else {
      const long i = e - K; const int c = (int)(i / n), j = (int)(i % n);
      queue[(long)c * K + p + j] = keys[(long)j * dim + c];
    }
  }
  __syncthreads();                               // (every thread of the block has used p)
  if (threadIdx.x == 0) {
    if (atomicAdd(&g_enq_ticket, 1u) == gridDim.x - 1) {
      __hip_atomic_store(ptr, (p + n) % K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      g_enq_ticket = 0;                          // ready for the next launch (stream order)
    }
  }
}
extern "C" int mscl_queue_enqueue(float* queue, int64_t* count, int64_t* ptr, const float* keys, int n, int dim, int K, void* stream) {
  if (!queue || !count || !ptr || !keys || n <= 0 || dim <= 0 || K <= 0) return MSCL_E_ARG;
  if (K % n) return MSCL_E_SHAPE;                 // recognizers/moco.py:432 `assert self.K % batch_size == 0`
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)K + (long)n * dim;
  hipLaunchKernelGGL(enqueue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, queue, count, ptr, keys, n, dim, K);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- LMCL: one block per clip
#define LMCL_MAX_T 32
// deterministic mode: the clips' losses meet in this table and the block that arrives last adds them in clip order (one launch
// for all clips; round 2 launched the clips one after the other, 8 x 22 us on the step's critical path).  Library-owned words: two
// mscl_lmcl launches must not run side by side in deterministic mode (the step has one).
#define LMCL_MAX_B 4096
__device__ float g_lmcl_part[LMCL_MAX_B];
__device__ unsigned g_lmcl_ticket;
// (round 6: 1024 threads per clip instead of 256 -- the kernel is a string of short phases on ONE block per clip, eight CUs in all, on the
// step's serial loss phase: 25 us under the tracer; sixteen waves share the 24 normalisations, the 128 frame-pair products and the
// gradient rows four times as wide)
__global__ __launch_bounds__(1024) void lmcl_kernel(const float* __restrict__ rgb, const float* __restrict__ flow,
                                                   float* __restrict__ loss_sum, int32_t* __restrict__ hits,
                                                   float* __restrict__ drgb, float* __restrict__ dflow, int B, int t, int C,
                                                   float inv_T, int det) {
  extern __shared__ float sm[];
  const int t2 = 2 * t;
  float* xr = sm;                    // [t][C] normalised
  float* xf = xr + t * C;            // [2t][C] normalised
  float* nr = xf + t2 * C;           // [t]
  float* nf = nr + t;                // [2t]
  float* sim = nf + t2;              // [t][2t] -> dlogits
  float* gr = sim + t * t2;          // [t][C]  grad wrt normalised rgb
  float* gf = gr + t * C;            // [2t][C]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = (int)blockDim.x >> 6, NT = (int)blockDim.x;
  float* lrow = gf + t2 * C;          // [t] per-frame loss terms of this clip, added up in frame order by one thread
  const float* rb = rgb + (long)b * t * C; const float* fb = flow + (long)b * t2 * C;
  for (int row = wave; row < t + t2; row += NW) {
    const float* src = row < t ? rb + row * C : fb + (row - t) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += src[c] * src[c];
    s = wave_sum(s);
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    float* dst = row < t ? xr + row * C : xf + (row - t) * C;
    for (int c = lane; c < C; c += 64) dst[c] = src[c] / nrm;
    if (lane == 0) { if (row < t) nr[row] = nrm; else nf[row - t] = nrm; }
  }
  __syncthreads();
  for (int e = wave; e < t * t2; e += NW) {
    const int i = e / t2, j = e % t2;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[i * C + c] * xf[j * C + c];
    s = wave_sum(s);
    if (lane == 0) sim[e] = s * inv_T;
  }
  __syncthreads();
  const float scale = 1.f / (float)(B * t);
  if (tid < t) {
    const int i = tid;
    float m = -INFINITY;
    for (int j = 0; j < t2; ++j) m = fmaxf(m, sim[i * t2 + j]);
    float s = 0.f;
    for (int j = 0; j < t2; ++j) s += expf(sim[i * t2 + j] - m);
    const float lse = m + logf(s), pos = sim[i * t2 + i];
    int rank = 0;
    for (int j = 0; j < t2; ++j) rank += (sim[i * t2 + j] > pos) ? 1 : 0;
    lrow[i] = lse - pos;
    if (rank == 0) atomicAdd(&hits[0], 1);
    if (rank < 5) atomicAdd(&hits[1], 1);
    for (int j = 0; j < t2; ++j) sim[i * t2 + j] = (expf(sim[i * t2 + j] - lse) - (j == i ? 1.f : 0.f)) * scale * inv_T;
  }
  __syncthreads();
  if (tid == 0) {                     // one float add per clip; deterministic mode: one add of the sum taken in clip order
    float s = 0.f;
    for (int i = 0; i < t; ++i) s += lrow[i];
    if (!det) atomicAdd(loss_sum, s);
    else {
      __hip_atomic_store(&g_lmcl_part[b], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      if (atomicAdd(&g_lmcl_ticket, 1u) == (unsigned)(B - 1)) {
        __threadfence();
        float tot = 0.f;
        for (int k = 0; k < B; ++k) tot += __hip_atomic_load(&g_lmcl_part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(loss_sum, tot);
        g_lmcl_ticket = 0;            // ready for the next launch (stream order)
      }
    }
  }
  for (int e = tid; e < t * C; e += NT) {
    const int i = e / C, c = e % C;
    float s = 0.f;
    for (int j = 0; j < t2; ++j) s += sim[i * t2 + j] * xf[j * C + c];
    gr[e] = s;
  }
  for (int e = tid; e < t2 * C; e += NT) {
    const int j = e / C, c = e % C;
    float s = 0.f;
    for (int i = 0; i < t; ++i) s += sim[i * t2 + j] * xr[i * C + c];
    gf[e] = s;
  }
  __syncthreads();
  for (int row = wave; row < t + t2; row += NW) {
    const bool isr = row < t;
    const float* n8 = isr ? xr + row * C : xf + (row - t) * C;
    const float* g8 = isr ? gr + row * C : gf + (row - t) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += n8[c] * g8[c];
    s = wave_sum(s);
    const float inv = 1.f / (isr ? nr[row] : nf[row - t]);
    float* dst = isr ? drgb + ((long)b * t + row) * C : dflow + ((long)b * t2 + (row - t)) * C;
    for (int c = lane; c < C; c += 64) dst[c] = (g8[c] - n8[c] * s) * inv;
  }
}
extern "C" int mscl_lmcl(const float* rgb, const float* flow, float* loss_sum, int32_t* hits, float* drgb, float* dflow, int B,
                         int t, int C, float inv_T, void* stream) {
  if (!rgb || !flow || !loss_sum || !hits || !drgb || !dflow || B <= 0 || t <= 0 || C <= 0) return MSCL_E_ARG;
  if (t > LMCL_MAX_T) return MSCL_E_SHAPE;
  const size_t lds = ((size_t)6 * t * C + 4 * t + 2 * t * t) * sizeof(float);
  if (lds > 150 * 1024) return MSCL_E_SHAPE;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lmcl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  const int det = (mscl_det() && B <= LMCL_MAX_B) ? 1 : 0;
  if (mscl_det() && !det) return MSCL_E_SHAPE;
  hipLaunchKernelGGL(lmcl_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, rgb, flow, loss_sum, hits, drgb, dflow, B, t, C, inv_T, det);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- positive-pair pieces of InfoNCE
// pos[r] = <a[r], b[r]>   (l_pos = einsum('nc,nc->n'), recognizers/moco.py:481)
__global__ __launch_bounds__(64) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, int dim) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += a[(long)r * dim + i] * b[(long)r * dim + i];
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}
extern "C" int mscl_rowdot(const float* a, const float* b, float* out, int rows, int dim, void* stream) {
  if (!a || !b || !out || rows <= 0 || dim <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(rowdot_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, a, b, out, dim);
  MSCL_LAUNCH_CHECK();
  return 0;
}
// dq[r][:] += row_scale[r] * inv_T * (softmax_pos[r] - 1) * kpos[r][:]
__global__ __launch_bounds__(64) void nce_pos_bwd_kernel(const float* __restrict__ kpos, const float* __restrict__ pos,
                                                         const float* __restrict__ lse, const float* __restrict__ row_scale,
                                                         float* __restrict__ dq, int dim, float inv_T) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float coef = row_scale[r] * inv_T * (__expf(pos[r] * inv_T - lse[r]) - 1.f);
  for (int i = lane; i < dim; i += 64) dq[(long)r * dim + i] += coef * kpos[(long)r * dim + i];
}
extern "C" int mscl_nce_pos_bwd(const float* kpos, const float* pos, const float* lse, const float* row_scale, float* dq,
                                int R, int dim, float inv_T, void* stream) {
  if (!kpos || !pos || !lse || !row_scale || !dq || R <= 0 || dim <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(nce_pos_bwd_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, kpos, pos, lse, row_scale, dq, dim, inv_T);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Loss-phase glue in two launches (was ~20 torch.cat / repeat / add / slice kernels between the InfoNCE passes):
// mscl_loss_pack lays the query / key rows of the three queue passes, their row scales and the LMCL flow frames out in ONE
// workspace; mscl_loss_unpack sums the passes' query gradients per input and splits the LMCL flow gradient back.
// Row groups (B rows each), in the order the reference builds its logits (recognizers/mscl.py:239-261, heads/moco_head_v2.py:38-53):
//   pass A (RGB queue, pre-enqueue):      queries [q_rgb | q_fb | q_fa*], keys k_rgb for every group            (* with aug_mx only)
//   pass C (flow queue, post-enqueue):    queries [q_fa | q_rgb | q_rgb*], keys [k_fa | k_fb | k_fa*], scales [w_intra/B | 1/B | 1/B*]
// Workspace (floats): QA[nA B D] KA[nA B D] QC[nC B D] KC[nC B D] sA[nA B] sC[nC B] ones[B] flow[B 2t Cf]
struct LossPackArgs {
  const float *q_rgb, *q_fb, *q_fa, *k_rgb, *k_fb, *k_fa, *p_fb, *p_fa;
  float* ws;
  float* pos;                // [(2 n + 1) B] positive logits of the passes' rows, A | B | C (NULL: not wanted)
  int B, D, t, Cf, n;        // n = 3 with the aug cross-modal terms, else 2
  float w_intra;
};
__global__ __launch_bounds__(256) void loss_pack_kernel(const LossPackArgs a) {
  const int BD = a.B * a.D, n = a.n;
  const long oQA = 0, oKA = (long)n * BD, oQC = 2L * n * BD, oKC = 3L * n * BD, osA = 4L * n * BD, osC = osA + (long)n * a.B,
             oone = osC + (long)n * a.B, oflow = oone + a.B;
  const long total = oflow + (long)a.B * 2 * a.t * a.Cf;
  const float invB = 1.f / (float)a.B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v;
    if (i < oKA) { const int gq = (int)(i / BD); const long e = i - (long)gq * BD; v = (gq == 0 ? a.q_rgb : gq == 1 ? a.q_fb : a.q_fa)[e]; }
    else if (i < oQC) { v = a.k_rgb[(i - oKA) % BD]; }
    else if (i < oKC) { const long r = i - oQC; const int gq = (int)(r / BD); v = (gq == 0 ? a.q_fa : a.q_rgb)[r - (long)gq * BD]; }
    else if (i < osA) { const long r = i - oKC; const int gq = (int)(r / BD); v = (gq == 1 ? a.k_fb : a.k_fa)[r - (long)gq * BD]; }
    else if (i < osC) v = invB;
    else if (i < oone) v = (i - osC < a.B) ? a.w_intra * invB : invB;
    else if (i < oflow) v = invB;
    else {                      // flow[b][f][c]: frames 0..t-1 of the base pass, t..2t-1 of the rotated pass (local_cl_head.py:59)
      const long r = i - oflow; const int c = (int)(r % a.Cf); const long bf = r / a.Cf; const int f = (int)(bf % (2 * a.t)), b = (int)(bf / (2 * a.t));
      v = f < a.t ? a.p_fb[((long)b * a.t + f) * a.Cf + c] : a.p_fa[((long)b * a.t + f - a.t) * a.Cf + c];
    }
    a.ws[i] = v;
  }
  // pos[r] = <query row, its key row> of every row of the three passes (l_pos = einsum('nc,nc->n'), recognizers/moco.py:481), from
  // the inputs themselves -- a wave per row, the arithmetic of rowdot_kernel: three launches less on the serial loss phase
  if (a.pos != nullptr) {
    const int lane = threadIdx.x & 63, nw = gridDim.x * 4, rows = (2 * n + 1) * a.B;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
      const int grp = r / a.B, b = r - grp * a.B;
      const float *qa, *ka;
      if (grp < n) { qa = grp == 0 ? a.q_rgb : grp == 1 ? a.q_fb : a.q_fa; ka = a.k_rgb; }            // pass A
      else if (grp == n) { qa = a.q_fb; ka = a.k_fb; }                                                  // pass B
      else { const int gq = grp - n - 1; qa = gq == 0 ? a.q_fa : a.q_rgb; ka = gq == 1 ? a.k_fb : a.k_fa; }      // pass C
      float sdot = 0.f;
      for (int i = lane; i < a.D; i += 64) sdot += qa[(long)b * a.D + i] * ka[(long)b * a.D + i];
      sdot = wave_sum(sdot);
      if (lane == 0) a.pos[r] = sdot;
    }
  }
}
extern "C" int mscl_loss_pack(const float* q_rgb, const float* q_fb, const float* q_fa, const float* k_rgb, const float* k_fb,
                              const float* k_fa, const float* p_fb, const float* p_fa, float* ws, float* pos, int B, int D, int t,
                              int Cf, int use_aug, float w_intra, void* stream) {
  if (!q_rgb || !q_fb || !q_fa || !k_rgb || !k_fb || !k_fa || !p_fb || !p_fa || !ws || B <= 0 || D <= 0 || t <= 0 || Cf <= 0) return MSCL_E_ARG;
  LossPackArgs a{q_rgb, q_fb, q_fa, k_rgb, k_fb, k_fa, p_fb, p_fa, ws, pos, B, D, t, Cf, use_aug ? 3 : 2, w_intra};
  hipLaunchKernelGGL(loss_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, a);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// out (floats): dq_rgb[B D] dq_fb[B D] dq_fa[B D] dp_rgb[B t C] dp_fb[B t Cf] dp_fa[B t Cf]
//   dq_rgb = dA[0] + dC[1] (+ dC[2]),  dq_fb = dA[1] + dB,  dq_fa = dC[0] (+ dA[2])      (dA, dC: [n B D], dB: [B D])
//   dp_fb / dp_fa = frames 0..t-1 / t..2t-1 of dpf [B 2t Cf];  dp_rgb = dpr [B t C]
__global__ __launch_bounds__(256) void loss_unpack_kernel(const float* __restrict__ dA, const float* __restrict__ dB,
                                                          const float* __restrict__ dC, const float* __restrict__ dpr,
                                                          const float* __restrict__ dpf, float* __restrict__ out, int B, int D, int t,
                                                          int C, int Cf, int n) {
  const long BD = (long)B * D, o1 = BD, o2 = 2 * BD, o3 = 3 * BD, o4 = o3 + (long)B * t * C, o5 = o4 + (long)B * t * Cf, total = o5 + (long)B * t * Cf;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v;
    if (i < o1) v = dA[i] + dC[BD + i] + (n == 3 ? dC[2 * BD + i] : 0.f);
    else if (i < o2) v = dA[BD + (i - o1)] + dB[i - o1];
    else if (i < o3) v = dC[i - o2] + (n == 3 ? dA[2 * BD + (i - o2)] : 0.f);
    else if (i < o4) v = dpr[i - o3];
    else {
      const bool aug = i >= o5; const long r = i - (aug ? o5 : o4);
      const int c = (int)(r % Cf); const long bf = r / Cf; const int f = (int)(bf % t), b = (int)(bf / t);
      v = dpf[((long)b * 2 * t + f + (aug ? t : 0)) * Cf + c];
    }
    out[i] = v;
  }
}
extern "C" int mscl_loss_unpack(const float* dA, const float* dB, const float* dC, const float* dpr, const float* dpf, float* out,
                                int B, int D, int t, int C, int Cf, int use_aug, void* stream) {
  if (!dA || !dB || !dC || !dpr || !dpf || !out || B <= 0 || D <= 0 || t <= 0 || C <= 0 || Cf <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(loss_unpack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, dA, dB, dC, dpr, dpf, out, B, D, t, C, Cf, use_aug ? 3 : 2);
  MSCL_LAUNCH_CHECK();
  return 0;
}
