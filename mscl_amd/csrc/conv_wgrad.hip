// Conv3d weight gradient on bf16 MFMA (gfx950):  dW[co][tap][ci] += sum_m dy[m][co] * x[src(m,tap)][ci]
//
// GEMM view: D[co][col] with col = tap*Cin + ci (the memory order of dW), reduction over positions m.
// Both operands are position-major in HBM (NDHWC), i.e. the reduction index is the ROW of both LDS
// tiles, so the MFMA fragments (8 consecutive reduction indices per lane) are column reads:
// ds_read_b64_tr_b16, CDNA4's transposing LDS read, delivers them with no shuffles.
//
// One block = CO output channels x NCOL columns (NCOL = 192 = three 64-channel taps that share ONE dy
// tile, or 64) x a slice of the positions, 64 positions per step:
//  * both tiles are staged by LDS-DMA (buffer_load ... lds), double-buffered; padded taps / tails use an
//    out-of-range offset -> hardware zero fill; the XOR swizzle that spreads the transposing reads over
//    the banks is applied on the source side (the LDS image of a wave-instruction is linear);
//  * the per-position decode (n,t,h,w -> base offset + separable tap-validity mask) is done ONCE per row
//    by 64 threads two steps ahead and shared through LDS -- it used to be redone per 16-byte granule;
//  * the 4 waves split the columns (no cross-wave reduction); each adds its fp32 tile to dW with one
//    atomic per element (dW is caller-zeroed; repeated trunk traversals accumulate).
#include "common.h"
#include <cstdlib>

struct WGeom {
  int N, T, H, W, C;       // x
  int To, Ho, Wo, K;       // dy
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int M, ntaps, cgs, ncols;                    // ncols = ntaps*C
  int co_tiles, col_tiles, splits, per_split;  // per_split: positions per split (multiple of 64)
  FastDiv dWo, dHo, dTo;
};

// XOR on the 16-byte granule index.  G = granules per tile row: a 32-lane group of a transposing read touches 8 rows x 32 B;
// with 128-byte rows odd / even rows already sit in opposite halves of the 256-byte bank row, with 256-byte rows (G = 16) they
// alias, so row bit 0 goes into granule bit 3 as well.
template <int G>
__device__ __forceinline__ int wswz(int row) {
  return (row & 2) | ((row >> 1) & 4) | (G == 16 ? ((row & 1) << 3) : 0);
}

// ds_read_b64_tr_b16 as inline asm (see the main loop) + the matching hand-placed wait; the "+v" operands tie the
// consumers of the two registers to the wait
__device__ __forceinline__ s16x4_t lds_tr_read(const unsigned char* p) {
  s16x4_t v;
  const unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)(p);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait2(s16x4_t& x, s16x4_t& y) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y)); }

__device__ __forceinline__ auto wg_rsrc(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// WCO = waves along the output-channel axis (1: the 4 waves split the columns and every wave reads the whole dy tile;
// 2: a 2 x 2 arrangement, used with the 128 x 128 tile where each wave then owns 64 x 64 and one LDS byte feeds 2.5x the MFMAs)
// `g` may live in the kernarg segment of a single launch or in the item table of a grouped one (conv_wgrad_group_kernel); `bidx` /
// `nblk`: this block's index among the layer's blocks and their number; `atomic`: dw is shared with another launch item (a conv
// module applied to several pyramid levels) or with other position splits -> float atomics, else the block owns its elements
template <int CO, int NCOL, int WCO>
__device__ __forceinline__ void conv_wgrad_body(const WGeom& g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                float* __restrict__ dw, float* __restrict__ slab, const int bidx, const int nblk,
                                                const bool atomic) {
  constexpr int PB = 64;                        // positions per step
  constexpr int GA = CO / 8, GB = NCOL / 8;     // granules per tile row
  constexpr int NA = (PB * GA + 255) / 256;     // dy DMA passes (one 1-KiB chunk per wave per pass)
  constexpr int NB = PB * GB / 256;             // x DMA passes
  constexpr int WCOL = 4 / WCO;
  constexpr int WM = CO / WCO, IA = WM / 16;    // output channels per wave
  constexpr int WN = NCOL / WCOL, JB = WN / 16; // columns per wave
  static_assert(WCO == 1 || WCO == 2, "wave arrangement");
  static_assert(WM % 16 == 0 && (GA < 16 || GA == 16) && (GB < 16 || GB == 16 || GB == 24), "swizzle covers rows of up to 16 granules (24: unswizzled bit 3)");
  static_assert((PB * GB) % 256 == 0 && WN % 16 == 0 && (PB * GA) % 64 == 0, "tile config");
  constexpr bool SWA = GA >= 8, SWB = GB >= 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* At = smem;                               // [2][PB][CO] bf16
  unsigned char* Bt = At + 2 * PB * CO * 2;               // [2][PB][NCOL] bf16
  int2* rinfo = reinterpret_cast<int2*>(Bt + 2 * PB * NCOL * 2);   // [8][PB] ring of {base position, validity mask}

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block order: all column tiles (taps) of ONE position slice are consecutive logical ids and the ids are
  // remapped so that an XCD gets a contiguous run: the slice's x / dy rows are then served 27x from that
  // XCD's L2 instead of the Infinity Cache (layer 1 re-reads 2.8 GB per launch otherwise)
  int bid = xcd_remap(bidx, nblk);
  const int colt = bid % g.col_tiles; bid /= g.col_tiles;
  const int split = bid % g.splits; const int cot = bid / g.splits;
  const int co0 = cot * CO, n0 = colt * NCOL;
  const int mbeg = split * g.per_split;
  const int mend = min(g.M, mbeg + g.per_split);
  const int nsteps = (mend > mbeg) ? (mend - mbeg + PB - 1) / PB : 0;
  if (nsteps == 0 && slab == nullptr) return;      // (deterministic mode: an empty split still stores its zeros)

  const unsigned x_bytes = (unsigned)((long)g.N * g.T * g.H * g.W * g.C * 2);
  const unsigned dy_bytes = (unsigned)((long)g.M * g.K * 2);
  const auto rs_x = wg_rsrc(x, x_bytes);
  const auto rs_dy = wg_rsrc(dy, dy_bytes);
  const int c2 = g.C * 2, k2 = g.K * 2;
  const int cmask = (1 << g.cgs) - 1;

  // ---- per-thread constants of the x-tile DMA: pass p moves granule G = (p*4 + wave)*64 + lane of the tile
  int xb_row[NB], xb_delta[NB], xb_bits[NB];    // tile row, byte delta of (tap, channel granule), tap validity bits (0 = dead column)
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    const int G = (p * 4 + wave) * 64 + lane;
    const int row = G / GB, pg = G % GB;
    const int lg = SWB ? (pg ^ wswz<GB>(row)) : pg;         // logical granule fetched into physical slot pg
    const int jg = (n0 >> 3) + lg;                          // global column granule
    int delta = 0, bits = 0;
    if (jg * 8 < g.ncols) {
      const int tap = jg >> g.cgs, cig = jg & cmask;
      const int kw = tap % g.kW, kh = (tap / g.kW) % g.kH, kt = tap / (g.kW * g.kH);
      delta = ((kt * g.H + kh) * g.W + kw) * c2 + cig * 16;
      bits = (1 << kt) | (1 << (8 + kh)) | (1 << (16 + kw));
    }
    xb_row[p] = row; xb_delta[p] = delta; xb_bits[p] = bits;
  }
  int ya_row[NA], ya_off[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int G = (p * 4 + wave) * 64 + lane;
    const int row = G / GA, pg = G % GA;
    const int lg = SWA ? (pg ^ wswz<GA>(row)) : pg;
    ya_row[p] = row; ya_off[p] = (co0 + lg * 8 < g.K) ? (co0 + lg * 8) * 2 : -1;
  }

  // rows of 4 consecutive steps (256 positions) are decoded at once by all 256 threads, every 4th step,
  // into an 8-slot ring: slot of step s is s & 7
  auto decode_rows = [&](int step0) {
    {
      const int m = mbeg + step0 * PB + tid;
      int base = 0, mask = 0;
      if (m < mend) {
        const int q1 = fdiv(m, g.dWo), wo = m - q1 * g.Wo;
        const int q2 = fdiv(q1, g.dHo), ho = q1 - q2 * g.Ho;
        const int n = fdiv(q2, g.dTo), to = q2 - n * g.To;
        const int t0 = to * g.sT - g.pT, h0 = ho * g.sH - g.pH, w0 = wo * g.sW - g.pW;
        for (int k = 0; k < g.kT; ++k) mask |= ((unsigned)(t0 + k) < (unsigned)g.T) ? (1 << k) : 0;
        for (int k = 0; k < g.kH; ++k) mask |= ((unsigned)(h0 + k) < (unsigned)g.H) ? (1 << (8 + k)) : 0;
        for (int k = 0; k < g.kW; ++k) mask |= ((unsigned)(w0 + k) < (unsigned)g.W) ? (1 << (16 + k)) : 0;
        base = ((n * g.T + t0) * g.H + h0) * g.W + w0;
      }
      rinfo[((step0 & 7) * PB + tid) & (8 * PB - 1)] = make_int2(base, mask);
    }
  };
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned xoff[NB];
  auto issue_tiles = [&](int step, int buf) {
    const int mb = mbeg + step * PB;
    unsigned char* a = At + buf * PB * CO * 2 + wave * 1024;
    unsigned char* b = Bt + buf * PB * NCOL * 2 + wave * 1024;
    const int2* ri = rinfo + (step & 7) * PB;
    // all LDS reads of the row table come BEFORE the first DMA of this tile: with an LDS-DMA in flight hipcc
    // guards every ds_read with s_waitcnt vmcnt(0), which would serialise the DMA instructions one by one
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int2 r = ri[xb_row[p]];
      const bool ok = xb_bits[p] != 0 && (r.y & xb_bits[p]) == xb_bits[p];
      xoff[p] = ok ? (unsigned)(r.x * c2 + xb_delta[p]) : x_bytes;
    }

#pragma unroll
    for (int p = 0; p < NA; ++p) {
      if ((p * 4 + wave) * 64 < PB * GA) {                 // wave-uniform (tiles narrower than 4 KiB)
        const int m = mb + ya_row[p];
        const unsigned off = (m < mend && ya_off[p] >= 0) ? (unsigned)(m * k2 + ya_off[p]) : dy_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(a + p * 4096), 16, off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const unsigned off = xoff[p];       // (a captured array element passed directly makes hipcc drop the host stub)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(b + p * 4096), 16, off, 0, 0, 0);
    }
  };

  f32x4_t acc[IA][JB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int wco = wave / WCOL, wcol = wave % WCOL;
  decode_rows(0);
  decode_rows(4);
  __syncthreads();
  issue_tiles(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) issue_tiles(s + 1, cur ^ 1);     // row info of step s+1 was published >= one barrier ago
    const unsigned char* a = At + cur * PB * CO * 2;
    const unsigned char* b = Bt + cur * PB * NCOL * 2;
#pragma unroll
    for (int ks = 0; ks < PB / 32; ++ks) {
      // The transposing reads are inline asm: hipcc guards every LDS read it knows about with s_waitcnt vmcnt(0) while an
      // LDS-DMA is in flight (it cannot see that the DMA fills the OTHER stage), which used to serialise "stage tile s+1"
      // and "compute tile s" completely.  The asm reads are invisible to that pass; lgkmcnt is waited for by hand.
      s16x4_t va[IA][2], vb[JB][2];
#pragma unroll
      for (int i = 0; i < IA; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + 8 * grp + 4 * h + qq;
          const int gq = (wco * WM) / 8 + i * 2 + (pp >> 1);
          const unsigned char* ad = a + r * (CO * 2) + ((SWA ? (gq ^ wswz<GA>(r)) : gq) * 16) + (pp & 1) * 8;
          va[i][h] = lds_tr_read(ad);
        }
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + 8 * grp + 4 * h + qq;
          const int gq = (wcol * WN) / 8 + j * 2 + (pp >> 1);
          const unsigned char* bd = b + r * (NCOL * 2) + ((SWB ? (gq ^ wswz<GB>(r)) : gq) * 16) + (pp & 1) * 8;
          vb[j][h] = lds_tr_read(bd);
        }
      }
      bf16x8_t fa[IA], fb[JB];
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
      for (int i = 0; i < IA; ++i) {
        lds_wait2(va[i][0], va[i][1]);
        s16x8_t w8 = {va[i][0][0], va[i][0][1], va[i][0][2], va[i][0][3], va[i][1][0], va[i][1][1], va[i][1][2], va[i][1][3]};
        fa[i] = __builtin_bit_cast(bf16x8_t, w8);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        lds_wait2(vb[j][0], vb[j][1]);
        s16x8_t w8 = {vb[j][0][0], vb[j][0][1], vb[j][0][2], vb[j][0][3], vb[j][1][0], vb[j][1][1], vb[j][1][2], vb[j][1][3]};
        fb[j] = __builtin_bit_cast(bf16x8_t, w8);
      }
#pragma unroll
      for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if ((s & 3) == 3 && s + 5 < nsteps) decode_rows(s + 5);           // steps s+5..s+8 reuse the slots of s-3..s (all consumed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- each wave adds its CO x WN tile: D row = co, col = column within the wave's slice ----
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco * WM + i * 16 + (lane >> 4) * 4 + r;
        const int col = n0 + wcol * WN + j * 16 + (lane & 15);
        if (co < g.K && col < g.ncols) {
          // deterministic mode: this split's tile goes to its own slab with plain stores; wgrad_slab_reduce_kernel adds the
          // slabs to dw in split order
          if (slab != nullptr) slab[((long)split * g.K + co) * g.ncols + col] = acc[i][j][r];
          else if (!atomic) dw[(long)co * g.ncols + col] += acc[i][j][r];             // one owner per element: no atomic needed
          else atomicAdd(&dw[(long)co * g.ncols + col], acc[i][j][r]);
        }
      }
}

template <int CO, int NCOL, int WCO>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WGeom g, const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ dy, float* __restrict__ dw,
                                                         float* __restrict__ slab) {
  conv_wgrad_body<CO, NCOL, WCO>(g, x, dy, dw, slab, (int)blockIdx.x, (int)gridDim.x, g.splits > 1);
}

// ---- grouped launch (round 5): the weight gradients (and bias gradients) of up to MSCL_WGRAD_GROUP_MAX layers in ONE launch --------
// A weight gradient is a leaf of the backward chain, and on the small maps (layers 3-4, their entries and shortcuts, the pyramid
// levels) conv_wgrad_kernel<64,64,1> is latency-bound: 8-41 us per launch at 6 % MFMA busy, 30 launches and 0.77 ms of kernel time
// per step, most of it on the RGB query chain with the chip nearly idle.  Streams do not help -- every fork edge inside the
// captured step costs more than it hides (profiles/r05_ab_sweeps.md) -- so the concurrency comes from the grid: the host defers
// these launches, keeps (x, dy) alive, and hands a bucket's worth of them over at once; block b finds its item by a scan of
// first[] (wave-uniform) and runs the unchanged body on that item's geometry.  Items that share a dw (the SEPC convs are applied to
// several pyramid levels) add with float atomics.
#define MSCL_WGRAD_GROUP_MAX 16
struct WGroupItem { WGeom g; const bf16_t* x; const bf16_t* dy; float* dw; int atomic; int pad_; };
struct WGroup { int n; int first[MSCL_WGRAD_GROUP_MAX + 1]; WGroupItem it[MSCL_WGRAD_GROUP_MAX]; };

__global__ __launch_bounds__(256) void conv_wgrad_group_kernel(const WGroup grp) {
  int i = 0;
#pragma unroll 1
  for (int k = 1; k < grp.n; ++k) if ((int)blockIdx.x >= grp.first[k]) i = k;
  const WGroupItem& it = grp.it[i];
  conv_wgrad_body<64, 64, 1>(it.g, it.x, it.dy, it.dw, nullptr, (int)blockIdx.x - grp.first[i], grp.first[i + 1] - grp.first[i],
                             it.atomic != 0);
}

// bias gradients of the same group: column sums of every item's dy in one launch (colsum_kernel's loop; blockIdx.y = item)
// (matrices wider than 512 columns: 512-column chunks along blockIdx.z, as colsum_kernel takes them along blockIdx.y; ldc = row pitch)
struct CGroupItem { const bf16_t* dy; float* dbias; long rows; int C; int blocks; int ldc; int chunks; };
struct CGroup { int n; CGroupItem it[MSCL_WGRAD_GROUP_MAX]; };
__global__ __launch_bounds__(256) void colsum_group_kernel(const CGroup grp) {
  CGroupItem it = grp.it[blockIdx.y];
  if ((int)blockIdx.x >= it.blocks || (int)blockIdx.z >= it.chunks) return;
  it.dy += blockIdx.z * it.C; it.dbias += blockIdx.z * it.C;
  const int C = it.C, G = C / 8;
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 256 / G;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int UNR = 4;
  const long stride = (long)it.blocks * RP;
  for (long r0 = (long)blockIdx.x * RP + tr; r0 < it.rows; r0 += stride * UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = r0 + u * stride;
      v[u] = *reinterpret_cast<const uint4*>(it.dy + (r < it.rows ? r : r0) * it.ldc + tg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + u * stride >= it.rows) break;
      float f[8]; unpack8(v[u], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += f[k];
    }
  }
  __shared__ float red[4 * 512];
  block_channel_sum(s, red, G, C, 1, 0);
  __syncthreads();
  for (int k = threadIdx.x; k < C; k += 256) atomicAdd(&it.dbias[k], red[k] + red[C + k] + red[2 * C + k] + red[3 * C + k]);
}

// dw[e] += slab[0][e] + slab[1][e] + ... in split order (deterministic mode)
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long n, int nslab) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nslab; k += 4) {          // four slabs in flight per trip (one per trip = one memory round trip per slab); order kept
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = slab[(long)(k + u < nslab ? k + u : k) * n + e];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (k + u < nslab) s += v[u];
    }
    dw[e] += s;
  }
}

// column sums, deterministic form: block x plain-stores the sums of its row share into part[x][C]; colsum_finish adds them in order
__global__ __launch_bounds__(256) void colsum_det_kernel(const bf16_t* __restrict__ xx, float* __restrict__ part, long rows, int C, int ldc) {
  xx += blockIdx.y * C; part += blockIdx.y * C;
  const int G = C / 8;
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 256 / G;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long rbeg = (long)blockIdx.x * per, rend = rbeg + per < rows ? rbeg + per : rows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int UNR = 4;            // loads of a trip issued together: the loop is latency-bound
  for (long r0 = rbeg + tr; r0 < rend; r0 += (long)RP * UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = r0 + (long)u * RP;
      v[u] = *reinterpret_cast<const uint4*>(xx + (r < rend ? r : r0) * ldc + tg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + (long)u * RP >= rend) break;
      float f[8]; unpack8(v[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
  __shared__ float red[4 * 512];
  block_channel_sum(s, red, G, C, 1, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) part[(long)blockIdx.x * ldc + i] = (red[i] + red[C + i]) + (red[2 * C + i] + red[3 * C + i]);
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C) return;
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= nblk; k += 8) {       // eight loads in flight, added in index order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u) * C + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nblk; ++k) s += part[(long)k * C + i];
  out[i] += s;
}

// column sums of a bf16 (rows, C) matrix into fp32 out[C] (+=): conv bias gradient
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ xx, float* __restrict__ out, long rows, int C, int ldc) {
  xx += blockIdx.y * C; out += blockIdx.y * C;      // matrices wider than 512 columns: 512-column chunks along blockIdx.y (ldc = row pitch)
  const int G = C / 8;
  const int tg = threadIdx.x % G;                 // requires 256 % G == 0 (C/8 power of two <= 256)
  const int tr = threadIdx.x / G, RP = 256 / G;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // few blocks, four rows in flight per thread: every block ends with one atomic per channel on the SAME C addresses, and
  // same-address atomics serialise in L2 at ~25 ns each (1568 blocks on the 50176 x 128 map: 40 us, all of it that queue)
  constexpr int UNR = 4;
  const long stride = (long)gridDim.x * RP;
  for (long r0 = (long)blockIdx.x * RP + tr; r0 < rows; r0 += stride * UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = r0 + u * stride;
      v[u] = *reinterpret_cast<const uint4*>(xx + (r < rows ? r : r0) * ldc + tg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + u * stride >= rows) break;
      float f[8]; unpack8(v[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
  __shared__ float red[4 * 512];
  block_channel_sum(s, red, G, C, 1, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&out[i], red[i] + red[C + i] + red[2 * C + i] + red[3 * C + i]);
}

// tiles and position splits of one layer; returns the number of blocks
template <int CO, int NCOL>
static long plan_w(WGeom& g) {
  g.co_tiles = (g.K + CO - 1) / CO;
  g.col_tiles = (g.ncols + NCOL - 1) / NCOL;
  const long tiles = (long)g.co_tiles * g.col_tiles;
  long target = (CO + NCOL) > 128 ? 512 : 768;            // blocks overall: 2 per CU with the 68-KB 128 x 128 tile, ~3 otherwise
  if (g.C <= 8) target = 2048;                            // stems: 16-byte gathers per position, the DMA latency wants more waves (95 -> 81 us)
  long want = (CO + NCOL) > 128 ? (target / tiles > 0 ? target / tiles : 1) : (target + tiles - 1) / tiles;
  long maxs = (g.M + 255) / 256;                          // at least 4 steps per block
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  long per = ((g.M + want - 1) / want + 63) / 64 * 64;
  g.per_split = (int)per;
  g.splits = (int)((g.M + per - 1) / per);
  return tiles * g.splits;
}

template <int CO, int NCOL, int WCO = 1>
static int launch_w(WGeom g, const bf16_t* x, const bf16_t* dy, float* dw, hipStream_t st, float* det_ws = nullptr, long det_floats = 0) {
  plan_w<CO, NCOL>(g);
  const long tiles = (long)g.co_tiles * g.col_tiles;
  long per;
  float* slab = nullptr;
  const long dwn = (long)g.K * g.ncols;
  if (mscl_det() && g.splits > 1) {                       // one fp32 slab per split, added to dw in split order afterwards
    if (det_ws == nullptr || det_floats < dwn) return MSCL_E_ARG;
    long fit = det_floats / dwn;
    if (fit < g.splits) {                                 // fewer, longer splits when the workspace is short
      per = ((g.M + fit - 1) / fit + 63) / 64 * 64;
      g.per_split = (int)per;
      g.splits = (int)((g.M + per - 1) / per);
    }
    slab = det_ws;
  }
  const size_t lds = (size_t)2 * 64 * (CO + NCOL) * 2 + (size_t)8 * 64 * sizeof(int2);
  auto kern = conv_wgrad_kernel<CO, NCOL, WCO>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * g.splits)), dim3(256), lds, st, g, x, dy, dw, slab);
  MSCL_LAUNCH_CHECK();
  if (slab != nullptr) {
    long rb = (dwn + 255) / 256; if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, (const float*)slab, dw, dwn, g.splits);
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}

static long g_wgrad_group_launches = 0;
extern "C" int64_t mscl_debug_wgrad_group_launches(void) { return g_wgrad_group_launches; }      // tests: the grouped launch ran

int mscl_wgrad_halo64(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats,
                      hipStream_t st);           // conv_wgrad_halo.hip
int mscl_wgrad_stem(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats,
                    hipStream_t st);             // conv_wgrad_stem.hip
int mscl_wgrad_thin(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw, float* ws, int64_t ws_floats,
                    hipStream_t st);             // conv_thin.hip
// 128 x 128 tile, 2 x 2 waves of 64 x 64: per 64-position step a wave makes 16 transposing reads for 16 MFMAs (the 64 x 64
// tile with the columns split four ways makes 10 for 4 and asks the LDS for 320 B/clk), and a block stages 32 KB for 128
// MFMAs instead of 16 KB for 32.
static bool big_tile(const mscl_conv_desc* d) {
  if (d->K < 128 || d->C < 64) return false;
  // measured (us, 64 -> 128 tile): 50176 positions x 27 taps 110 -> 86 (128 ch), 65 -> 57 (64 -> 128 ch, stride 2); 6272 positions
  // 58 -> 56; but 9-tap / 1-tap layers and maps of a few thousand positions lose (too few tiles to split over): 44 -> 47, 24 -> 32
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  // (round 4: the 128 x 128 tile on the 784-position maps of layer 4, 432 tiles in one split: 44.4 vs 41.5 us, 32.8 vs 25.9 on its
  // entry conv -- the small maps stay with the 64 x 64 tile)
  return M >= 16384 && (long)d->kT * d->kH * d->kW * d->C >= 1728;
}

static int wgeom_of(const mscl_conv_desc* d, WGeom& g) {
  if (d->C % 8 || d->K % 8) return MSCL_E_SHAPE;
  if (ilog2_exact(d->K / 8) < 0 || d->K / 8 > 256 || ilog2_exact(d->C / 8) < 0) return MSCL_E_SHAPE;
  if (d->kT > 8 || d->kH > 8 || d->kW > 8) return MSCL_E_SHAPE;
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.C = d->C;
  g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.K = d->K;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M * d->K >= (1L << 30) || (long)d->N * d->T * d->H * d->W * d->C >= (1L << 30)) return MSCL_E_SHAPE;   // 32-bit byte offsets
  g.M = (int)M; g.ntaps = d->kT * d->kH * d->kW; g.cgs = ilog2_exact(d->C / 8); g.ncols = g.ntaps * d->C;
  g.dWo = make_fastdiv(d->Wo); g.dHo = make_fastdiv(d->Ho); g.dTo = make_fastdiv(d->To);
  return 0;
}

extern "C" int64_t mscl_conv3d_wgrad_ws(const mscl_conv_desc* d, int with_bias);
// 1: mscl_conv3d_wgrad would run this layer on conv_wgrad_kernel<64,64,1> with no workspace -- the layers a grouped launch takes
extern "C" int mscl_conv3d_wgrad_groupable(const mscl_conv_desc* d) {
  if (!d || mscl_det()) return 0;                 // deterministic mode sums per-split slabs in a pass of its own: one launch per layer
  WGeom g{};
  if (wgeom_of(d, g) != 0) return 0;
  if (d->K < 64 || big_tile(d)) return 0;
  return mscl_conv3d_wgrad_ws(d, 0) == 0 ? 1 : 0;  // (> 0: a window-resident kernel takes the layer)
}

// The weight (and, where dbias[i] != NULL, bias) gradients of n <= MSCL_WGRAD_GROUP_MAX groupable layers in one launch each.
// descs: n descriptors; x / dy / dw / dbias: n device pointers each (host arrays).
extern "C" int mscl_conv3d_wgrad_group(int n, const mscl_conv_desc* descs, const uint16_t* const* x, const uint16_t* const* dy,
                                       float* const* dw, float* const* dbias, void* stream) {
  if (n <= 0 || n > MSCL_WGRAD_GROUP_MAX || !descs || !x || !dy || !dw) return MSCL_E_ARG;
  WGroup grp{};
  CGroup cg{};
  grp.n = n;
  long total = 0, cmax = 0;
  int zmax = 1;
  for (int i = 0; i < n; ++i) {
    if (!x[i] || !dy[i] || !dw[i]) return MSCL_E_ARG;
    if (!mscl_conv3d_wgrad_groupable(&descs[i])) return MSCL_E_SHAPE;
    WGroupItem& it = grp.it[i];
    const int ge = wgeom_of(&descs[i], it.g); if (ge) return ge;
    const long nb = plan_w<64, 64>(it.g);
    it.x = x[i]; it.dy = dy[i]; it.dw = dw[i];
    it.atomic = it.g.splits > 1;
    for (int k = 0; k < i; ++k) if (dw[k] == dw[i]) { it.atomic = 1; grp.it[k].atomic = 1; }     // one module, several applications
    grp.first[i] = (int)total;
    total += nb;
    if (total >= (1L << 30)) return MSCL_E_SHAPE;
    if (dbias && dbias[i]) {
      const mscl_conv_desc* d = &descs[i];
      const int Kc = d->K > 512 ? 512 : d->K;                // K / 8 is a power of two (wgeom_of): 512 divides a wider K
      CGroupItem& c = cg.it[cg.n++];
      c.dy = dy[i]; c.dbias = dbias[i]; c.rows = it.g.M; c.C = Kc; c.ldc = d->K; c.chunks = d->K / Kc;
      if (c.chunks > zmax) zmax = c.chunks;
      const int RPc = 256 / (Kc / 8);
      long b = (c.rows + RPc * 4 - 1) / (RPc * 4); if (b > 256) b = 256; if (b < 1) b = 1;
      c.blocks = (int)b;
      if (b > cmax) cmax = b;
    }
  }
  grp.first[n] = (int)total;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)2 * 64 * (64 + 64) * 2 + (size_t)8 * 64 * sizeof(int2);
  hipLaunchKernelGGL(conv_wgrad_group_kernel, dim3((unsigned)total), dim3(256), lds, st, grp);
  MSCL_LAUNCH_CHECK();
  if (cg.n > 0) {
    hipLaunchKernelGGL(colsum_group_kernel, dim3((unsigned)cmax, cg.n, zmax), dim3(256), 0, st, cg);
    MSCL_LAUNCH_CHECK();
  }
  ++g_wgrad_group_launches;
  return 0;
}

extern "C" int mscl_conv3d_wgrad(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw,
                                 float* dbias, float* ws, int64_t ws_floats, void* stream) {
  if (!d || !x || !dy || !dw) return MSCL_E_ARG;
  WGeom g{};
  { const int ge = wgeom_of(d, g); if (ge) return ge; }
  const long M = g.M;
  hipStream_t st = (hipStream_t)stream;
  int e;
  // 3x3x3 / 1 / 1 layers whose planes fill 256-position tiles: window-resident kernel on 64 x 64 channel slices (its slabs are
  // added in slot order: deterministic as it stands); the tail of `ws` stays free for deterministic mode's bias partials
  const long btail = (mscl_det() && dbias) ? (long)MSCL_DET_PARTS * d->K : 0;
  const int hres = (ws != nullptr && ws_floats > btail) ? mscl_wgrad_halo64(d, x, dy, dw, ws, ws_floats - btail, st) : 0;
  if (hres < 0 || hres > 1) return hres;
  // deterministic mode: `ws` doubles as the slab workspace of the general kernel ([splits][K][ncols] floats) and, behind it,
  // the partial column sums of the bias gradient ([MSCL_DET_PARTS][K]); mscl_conv3d_wgrad_ws() gives the size to pass
  float* dws = nullptr; long dfl = 0;
  if (mscl_det() && hres == 0 && ws != nullptr) {
    const long tail = dbias ? (long)MSCL_DET_PARTS * d->K : 0;
    dws = ws; dfl = ws_floats - tail;
    if (dfl < 0) return MSCL_E_ARG;
  }
  // (NCOL = 192, three taps sharing one dy tile, measured slower than 64 -- fewer blocks per CU -- and was dropped; so was the
  // shared-tap ping-pong kernel of round 3, conv_wgrad_pp.hip: a tie with the 128 x 128 tile at best, with two or three ring slots and
  // with the DMA pieces issued from either section -- 82.9-84.9 vs 84.7 us on 128 -> 128 -- because its L sections, 20 transposing
  // reads + 4 pieces + their row decode for 24 MFMAs, are three times as long as its M sections; the window-resident kernel on
  // channel slices, conv_wgrad_halo.hip, takes those layers at 62 us)
  const int pres = 0;
  int tres = 0;
  if (hres == 0 && pres == 0 && ws != nullptr) {          // 1x3x3 between 16- / 32-channel maps: window-resident kernel (conv_thin.hip)
    const long tail = (mscl_det() && dbias) ? (long)MSCL_DET_PARTS * d->K : 0;
    tres = mscl_wgrad_thin(d, x, dy, dw, ws, ws_floats - tail, st);
    if (tres < 0 || tres > 1) return tres;
  }
  int sres = 0;
  if (hres == 0 && tres == 0 && ws != nullptr) {          // W-paired RGB stem: window-resident kernel (conv_wgrad_stem.hip)
    const long tail = (mscl_det() && dbias) ? (long)MSCL_DET_PARTS * d->K : 0;
    sres = mscl_wgrad_stem(d, x, dy, dw, ws, ws_floats - tail, st);
    if (sres < 0 || sres > 1) return sres;
  }
  if (hres == 1 || pres == 1 || tres == 1 || sres == 1) e = 0;
  else if (big_tile(d)) e = launch_w<128, 128, 2>(g, x, dy, dw, st, dws, dfl);
  else if (d->K >= 64) e = launch_w<64, 64>(g, x, dy, dw, st, dws, dfl);
  else if (d->K == 32) e = launch_w<32, 64>(g, x, dy, dw, st, dws, dfl);
  // K == 8 (the 8-channel layers of r2d_50): the 16-row tile with its upper half zero-filled by the range check and never stored
  else if (d->K == 16 || d->K == 8) e = launch_w<16, 64>(g, x, dy, dw, st, dws, dfl);
  else return MSCL_E_SHAPE;
  if (e) return e;
  if (dbias) {
    const int Kc = d->K > 512 ? 512 : d->K, kchunks = d->K / Kc;        // K/8 is a power of two (checked above)
    if (mscl_det()) {
      if (ws == nullptr || ws_floats < (long)MSCL_DET_PARTS * d->K) return MSCL_E_ARG;
      float* part = ws + (ws_floats - (long)MSCL_DET_PARTS * d->K);
      const int RPd = 256 / (Kc / 8);
      long P = (M + RPd * 8 - 1) / (RPd * 8); if (P > MSCL_DET_PARTS) P = MSCL_DET_PARTS; if (P < 1) P = 1;      // a function of the shape alone
      hipLaunchKernelGGL(colsum_det_kernel, dim3((unsigned)P, kchunks), dim3(256), 0, st, dy, part, M, Kc, d->K);
      MSCL_LAUNCH_CHECK();
      hipLaunchKernelGGL(colsum_finish_kernel, dim3((d->K + 255) / 256), dim3(256), 0, st, (const float*)part, dbias, (int)P, d->K);
      MSCL_LAUNCH_CHECK();
      return 0;
    }
    const int RPc = 256 / (Kc / 8);
    long blocks = (M + RPc * 4 - 1) / (RPc * 4); if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks, kchunks), dim3(256), 0, st, dy, dbias, M, Kc, d->K);
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}

// floats of `ws` mscl_conv3d_wgrad wants for this layer in deterministic mode (at most 64 slabs of the weight gradient plus
// the bias partials); 0 outside deterministic mode for layers that do not use the window-resident kernel
extern "C" int64_t mscl_wgrad_thin_ws(const mscl_conv_desc* d);
extern "C" int64_t mscl_wgrad_halo_ws(const mscl_conv_desc* d);
extern "C" int64_t mscl_wgrad_stem_ws(const mscl_conv_desc* d);
extern "C" int64_t mscl_conv3d_wgrad_ws(const mscl_conv_desc* d, int with_bias) {
  if (!d) return 0;
  int64_t pp = mscl_wgrad_halo_ws(d);
  if (pp == 0) pp = mscl_wgrad_thin_ws(d);
  if (pp == 0) pp = mscl_wgrad_stem_ws(d);
  if (!mscl_det()) return pp;
  if (pp > 0) return pp + (with_bias ? (int64_t)MSCL_DET_PARTS * d->K : 0);
  const int64_t dwn = (int64_t)d->K * d->kT * d->kH * d->kW * d->C;
  int64_t slabs = ((int64_t)1 << 28) / (dwn > 0 ? dwn : 1);          // cap the slab workspace at 1 GiB
  if (slabs > 64) slabs = 64;
  if (slabs < 2) slabs = 2;
  return slabs * dwn + (with_bias ? (int64_t)MSCL_DET_PARTS * d->K : 0);
}
