// 1 x 3 x 3 stride-1 convolutions between 16- and 32-channel maps: the layer1 / layer2 convolutions of the flow trunk.
// Reference op: the conv2 / conv1 of BasicBlock in r2d_18 as the flow encoder runs it (mmaction/models/backbones/r3d.py:36-60
// Conv2DSimple -> nn.Conv3d(k = (1,3,3), pad = (0,1,1)); widths 16 / 32 in configs/.../mscl_r18_cosm_lr2e-2.py), forward and
// input gradient (the same gather with the transposed kernel and mirrored taps).
//
// These layers are byte-bound, not MFMA-bound: 0.9 GFLOP on a 12.8-MB map (16 clips x 8 frames x 56 x 56 x 16 channels), the map in
// and the map out are all the traffic there has to be.  The implicit-GEMM kernels stage the A tile once per tap: nine passes of the
// map through L2 -> LDS, and that fill rate (not HBM, not the MFMAs) set their time: 23.5 us per launch inside the step for
// 25.7 MB.  Here a block keeps its WINDOW of the map in LDS -- RB output rows of one (n, t) plane plus a halo row / column each
// side, zero-filled at the borders -- and takes all nine taps from it; the kernel (9 x C x K bf16 = 4.6 ... 18 KB) lives in
// registers as MFMA fragments for the whole block.
//  * operands swapped: the kernel is the MFMA's A operand (rows = output channels), the positions are the columns, so a lane
//    ends up with 4 consecutive output channels of one position: one 8-byte store, 16 lanes x 4 quads = 512 contiguous bytes;
//  * C = 16: a 32-deep MFMA step holds TWO taps x 16 channels (five steps, the tenth tap is zero weights); C = 32: a tap a step;
//  * the fragment of a step is one ds_read_b128 per lane at [window position + tap offset][channel group];
//  * BatchNorm statistics of the raw fp32 result as in igemm.h: lane sums over its tiles, row16_sum, LDS across the waves, one
//    float atomic per channel and block into the statistics slot of the plane's group.
#include "igemm.h"
#include <atomic>
#include <cstdlib>

struct ThinGeom {
  int planes, H, W, RB, bands;
  int grp_planes;       // planes per BatchNorm statistics group
  int flip;             // 1: taps mirrored (input gradient)
  FastDiv dW, dWp, dBands;
};

template <int C, int K>
__global__ __launch_bounds__(256) void conv_thin_kernel(const ThinGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        bf16_t* __restrict__ y, const float* __restrict__ bias,
                                                        const bf16_t* __restrict__ addend, float* __restrict__ ssum,
                                                        float* __restrict__ ssq, int relu) {
  constexpr int NS = C == 16 ? 5 : 9;           // MFMA steps of 32 reduction elements
  constexpr int NT = K / 16;                    // output-channel tiles
  constexpr int GP = C / 8;                     // 16-byte granules per position
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fj = lane & 15, fg = lane >> 4;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int plane = fdiv(bid, g.dBands), band = bid - plane * g.bands;
  const int r0 = band * g.RB;
  const int rows = min(g.RB, g.H - r0);
  const int Wp = g.W + 2;

  // ---- the kernel as A fragments: row = output channel fj of tile nt, 8 reduction elements of step s ----
  bf16x8_t wf[NS][NT];
  int toff[NS];                                 // byte offset of this lane's tap and channel group inside the window
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int tap = C == 16 ? 2 * s + (fg >> 1) : s;
    const int ch0 = C == 16 ? (fg & 1) * 8 : fg * 8;
    const bool live = tap < 9;
    const int tq = live ? tap : 0;
    const int th = fdiv(tq, FastDiv{0xAAAAAAABu, 33}), tw = tq - 3 * th;      // tap -> (kh, kw)
    const int dr = g.flip ? 1 - th : th - 1, dc = g.flip ? 1 - tw : tw - 1;
    toff[s] = ((dr * Wp + dc) * C + ch0) * 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (live) v = *reinterpret_cast<const uint4*>(w + ((long)(nt * 16 + fj) * 9 + tq) * C + ch0);
      wf[s][nt] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ---- the window: (rows + 2) x (W + 2) positions x C channels, position-major ----
  {
    const int total = (rows + 2) * Wp * GP;
    const bf16_t* xp = x + (long)plane * g.H * g.W * C;
    constexpr int UNR = 4;
    for (int i0 = tid; i0 < total; i0 += 256 * UNR) {
      uint4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = i0 + u * 256;
        const int pos = i / GP, gq = i - pos * GP;
        const int wr = fdiv(pos, g.dWp), wc = pos - wr * Wp;
        const int hr = r0 + wr - 1, cc = wc - 1;
        const bool ok = i < total && (unsigned)hr < (unsigned)g.H && (unsigned)cc < (unsigned)g.W;
        v[u] = ok ? *reinterpret_cast<const uint4*>(xp + ((long)hr * g.W + cc) * C + gq * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = i0 + u * 256;
        if (i < total) *reinterpret_cast<uint4*>(smem + (long)i * 16) = v[u];
      }
    }
  }
  __syncthreads();

  // ---- 16 positions x K channels per wave and trip ----
  const int npos = rows * g.W;
  const int ntile = (npos + 15) >> 4;
  const long obase = ((long)plane * g.H + r0) * g.W;          // first output position of the band
  float st_s[NT][4], st_q[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[nt][r] = 0.f; st_q[nt][r] = 0.f; }
  for (int t = wave; t < ntile; t += 4) {
    const int p = t * 16 + fj;
    const bool ok = p < npos;
    const int pr = ok ? fdiv(p, g.dW) : 0, pc = ok ? p - pr * g.W : 0;
    const int base = ((pr + 1) * Wp + pc + 1) * C * 2;
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + base + toff[s]));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][nt], b, acc[nt], 0, 0, 0);
    }
    if (!ok) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[4] = {acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) { st_s[nt][r] += v[r]; st_q[nt][r] += v[r] * v[r]; }
      const int n = nt * 16 + fg * 4;
      const long o = (obase + p) * K + n;
      if (bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (addend != nullptr) {
        const uint2 av = *reinterpret_cast<const uint2*>(addend + o);
        v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
        v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
      }
      if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(y + o) = pv;
    }
  }

  // ---- BatchNorm statistics ----
  if (ssum != nullptr) {
    __syncthreads();                              // the window is dead
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][K]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(st_s[nt][r]), q = row16_sum(st_q[nt][r]);
        if (fj == 0) { red[(wave * 2 + 0) * K + nt * 16 + fg * 4 + r] = s; red[(wave * 2 + 1) * K + nt * 16 + fg * 4 + r] = q; }
      }
    __syncthreads();
    if (tid < 2 * K) {
      const int v = tid / K, c = tid - v * K;
      const float t = (red[(0 * 2 + v) * K + c] + red[(1 * 2 + v) * K + c]) + (red[(2 * 2 + v) * K + c] + red[(3 * 2 + v) * K + c]);
      const int grp = plane / g.grp_planes;
      const long so = ((long)grp * MSCL_STAT_SLOTS + (int)(blockIdx.x % MSCL_STAT_ACTIVE)) * 2 * K + c;
      atomicAdd(v == 0 ? &ssum[so] : &ssq[so], t);
    }
  }
}

static std::atomic<long> g_thin_launches{0};
extern "C" int64_t mscl_debug_thin_launches(void) { return g_thin_launches.load(); }

// 1 = launched, 0 = shape not covered.  flip = 0: y = conv(x, w[K][9][C]); flip = 1: the input gradient, x = dy, w = the
// transposed kernel [Cin][9][Cout], C = the forward's Cout, K = its Cin.
int mscl_conv_thin(int planes, int H, int W, int C, int K, int flip, const bf16_t* x, const bf16_t* w, bf16_t* y, const float* bias,
                   const bf16_t* addend, int relu, float* ssum, float* ssq, int stat_groups, hipStream_t st) {
  if (!((C == 16 || C == 32) && (K == 16 || K == 32))) return 0;
  if (H < 4 || W < 4 || W > 254 || planes < 1 || stat_groups < 1 || planes % stat_groups) return 0;
  { static MsclTune t("MSCL_THIN"); if (t.read() && t.val == 0) return 0; }
  ThinGeom g;
  g.planes = planes; g.H = H; g.W = W;
  g.RB = (H % 8 == 0) ? 8 : (H % 7 == 0) ? 7 : 8;
  g.bands = (H + g.RB - 1) / g.RB;
  g.grp_planes = planes / stat_groups;
  g.flip = flip;
  g.dW = make_fastdiv(W); g.dWp = make_fastdiv(W + 2); g.dBands = make_fastdiv(g.bands);
  const long blocks = (long)planes * g.bands;
  if (blocks >= (1L << 31)) return 0;
  size_t lds = (size_t)(g.RB + 2) * (W + 2) * C * 2;
  if (lds < (size_t)8 * K * sizeof(float)) lds = (size_t)8 * K * sizeof(float);
  if (lds > 64 * 1024) return 0;
#define THIN_GO(CC, KK) hipLaunchKernelGGL((conv_thin_kernel<CC, KK>), dim3((unsigned)blocks), dim3(256), lds, st, g, x, w, y, bias, addend, ssum, ssq, relu)
  if (C == 16 && K == 16) THIN_GO(16, 16);
  else if (C == 16 && K == 32) THIN_GO(16, 32);
  else if (C == 32 && K == 16) THIN_GO(32, 16);
  else THIN_GO(32, 32);
#undef THIN_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -(int)e;
  g_thin_launches.fetch_add(1);
  return 1;
}

// ---------------------------------------------------------------- weight gradient of the same layers
//   dW[k][tap][c] += sum over positions of dy[pos][k] * x[pos + off(tap)][c]
// Reference op: the weight gradient autograd computes for those nn.Conv3d (r3d.py:36-60).  Byte-bound again (the two maps in, 9 x C
// x K floats out), and the output is so small that float atomics are the wrong tool: every block adding into the same 9-KB of dW
// runs 14x below the atomic rate (MI355X_MICROARCH.md, Global float atomics, row 'contention') -- the 16-row tile of
// conv_wgrad.hip took 31 us on the layer1 map for that reason.  Here:
//  * a block walks band units (RB rows of one plane) and keeps BOTH maps' windows in LDS in the padded-linear layout of
//    conv_wgrad_pp.hip: position q = row * (W + 2) + column with a zero column each side, dy rows [q], x rows [q + tap offset] -- the
//    reduction index is affine, no per-tap masks;
//  * nine waves, one tap each (no sum across waves): per 32 positions KT x CT MFMAs from 2 (KT + CT) transposing reads
//    (ds_read_b64_tr_b16: both operands are position-major); the reduction slot of lane group g, read h, row j is position
//    16 h + 4 g + j, so a 32-lane half reads 8 consecutive rows;
//  * every block plain-stores its 9 x K x C partial to a slab; wgrad_thin_reduce_kernel adds the slabs in block order into dW
//    (fixed order: the same bits every run, so deterministic mode takes this path as it is).
struct ThinWGeom {
  int planes, H, W, RB, bands, units, upb;     // upb: band units per block
  int QD, XR;                                   // dy rows (multiple of 32), x rows in LDS
  FastDiv dWp, dBands;
};

#define THIN_TR(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))

template <int C, int K>
__global__ __launch_bounds__(576) void wgrad_thin_kernel(const ThinWGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                         float* __restrict__ slab) {
  constexpr int KT = K / 16, CT = C / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tap = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fg = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int Wp = g.W + 2;
  const unsigned dyL = lds0, xL = lds0 + (unsigned)g.QD * K * 2;       // x rows start one guard position in (index -1 is row 0)
  const int th = tap / 3, tw = tap - 3 * th;
  const int xoff = Wp + (th - 1) * Wp + (tw - 1) + 1;                  // + 1: the guard row

  f32x4_t acc[KT][CT];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int u = blockIdx.x * g.upb; u < min(g.units, (int)(blockIdx.x + 1) * g.upb); ++u) {
    const int plane = fdiv(u, g.dBands), band = u - plane * g.bands;
    const int r0 = band * g.RB;
    const int rows = min(g.RB, g.H - r0);
    __syncthreads();                            // the previous unit's reads are done
    {   // dy: QD padded positions x K channels
      constexpr int GP = K / 8;
      const int total = g.QD * GP;
      const bf16_t* dp = dy + ((long)plane * g.H + r0) * g.W * K;
      for (int i = tid; i < total; i += 576) {
        const int q = i / GP, gq = i - q * GP;
        const int pr = fdiv(q, g.dWp), pc = q - pr * Wp - 1;
        const bool ok = pr < rows && (unsigned)pc < (unsigned)g.W;
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(dp + ((long)pr * g.W + pc) * K + gq * 8) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(smem + (long)i * 16) = v;
      }
    }
    {   // x: guard row, then the (rows + 2) x (W + 2) window, then zeros up to XR rows
      constexpr int GP = C / 8;
      const int total = g.XR * GP;
      const bf16_t* xp = x + (long)plane * g.H * g.W * C;
      unsigned char* xs = smem + (long)g.QD * K * 2;
      for (int i = tid; i < total; i += 576) {
        const int q = i / GP - 1, gq = i % GP;
        const int wr = q >= 0 ? fdiv(q, g.dWp) : 0, wc = q - wr * Wp;
        const int hr = r0 + wr - 1, cc = wc - 1;
        const bool ok = q >= 0 && wr < rows + 2 && (unsigned)hr < (unsigned)g.H && (unsigned)cc < (unsigned)g.W;
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(xp + ((long)hr * g.W + cc) * C + gq * 8) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(xs + (long)i * 16) = v;
      }
    }
    __syncthreads();
    const int nstep = (rows * Wp + 31) >> 5;
    for (int s = 0; s < nstep; ++s) {
      s16x4_t va[2][KT], vb[2][CT];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = s * 32 + 16 * h + 4 * fg + qq;
#pragma unroll
        for (int a = 0; a < KT; ++a) THIN_TR(va[h][a], dyL + (unsigned)(r * K * 2 + a * 32 + pp * 8));
#pragma unroll
        for (int b = 0; b < CT; ++b) THIN_TR(vb[h][b], xL + (unsigned)((r + xoff) * C * 2 + b * 32 + pp * 8));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      bf16x8_t fa[KT], fb[CT];
#pragma unroll
      for (int a = 0; a < KT; ++a) {
        const s16x8_t w8 = {va[0][a][0], va[0][a][1], va[0][a][2], va[0][a][3], va[1][a][0], va[1][a][1], va[1][a][2], va[1][a][3]};
        fa[a] = __builtin_bit_cast(bf16x8_t, w8);
      }
#pragma unroll
      for (int b = 0; b < CT; ++b) {
        const s16x8_t w8 = {vb[0][b][0], vb[0][b][1], vb[0][b][2], vb[0][b][3], vb[1][b][0], vb[1][b][1], vb[1][b][2], vb[1][b][3]};
        fb[b] = __builtin_bit_cast(bf16x8_t, w8);
      }
#pragma unroll
      for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  // the block's partial in dW layout [k][tap][c]: lane holds c = lane & 15, k = 4 * (lane >> 4) + r
  float* out = slab + (long)blockIdx.x * 9 * K * C;
  const int fj = lane & 15;
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((a * 16 + fg * 4 + r) * 9 + tap) * C + b * 16 + fj] = acc[a][b][r];
}

// dw[i] += sum over blocks (in block order) of slab[b][i]: eight lanes per element take an eighth of the blocks each
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nb, int n) {
  __shared__ float red[8][32];
  const int ix = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + ix;
  const int per = (nb + 7) >> 3;
  const int pe = min(nb, (py + 1) * per);
  float t = 0.f;
  if (i < n) {
    int p = py * per;
    for (; p + 8 <= pe; p += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slab[(long)(p + u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; p < pe; ++p) t += slab[(long)p * n + i];
  }
  red[py][ix] = t;
  __syncthreads();
  if (py != 0 || i >= n) return;
  t = red[0][ix];
#pragma unroll
  for (int k = 1; k < 8; ++k) t += red[k][ix];
  dw[i] += t;
}

static std::atomic<long> g_thin_wgrad_launches{0};
extern "C" int64_t mscl_debug_thin_wgrad_launches(void) { return g_thin_wgrad_launches.load(); }

static bool thin_wgrad_geom(const mscl_conv_desc* d, ThinWGeom& g) {
  if (!(d->kT == 1 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 && d->pH == 1 && d->pW == 1)) return false;
  if (!((d->C == 16 || d->C == 32) && (d->K == 16 || d->K == 32))) return false;
  if (d->H < 4 || d->W < 4 || d->W > 254) return false;
  { static MsclTune t("MSCL_THIN"); if (t.read() && t.val == 0) return false; }
  g.planes = d->N * d->T; g.H = d->H; g.W = d->W;
  g.RB = (d->H % 8 == 0) ? 8 : (d->H % 7 == 0) ? 7 : 8;
  g.bands = (d->H + g.RB - 1) / g.RB;
  const long units = (long)g.planes * g.bands;
  if (units >= (1L << 30)) return false;
  g.units = (int)units;
  const long pbytes = (long)9 * d->K * d->C * 4;
  long upb = (units + 511) / 512;                                  // at most 512 blocks ...
  const long byb = (units * pbytes + (12L << 20) - 1) / (12L << 20);      // ... and at most 12 MB of slabs
  if (byb > upb) upb = byb;
  g.upb = (int)upb;
  const int Wp = d->W + 2;
  g.QD = (g.RB * Wp + 31) / 32 * 32;
  g.XR = g.QD + 2 * Wp + 3;
  g.dWp = make_fastdiv(Wp); g.dBands = make_fastdiv(g.bands);
  return ((size_t)g.QD * d->K + (size_t)g.XR * d->C) * 2 <= 64 * 1024;
}

extern "C" int64_t mscl_wgrad_thin_ws(const mscl_conv_desc* d) {
  ThinWGeom g;
  if (!d || !thin_wgrad_geom(d, g)) return 0;
  return (int64_t)((g.units + g.upb - 1) / g.upb) * 9 * d->K * d->C;
}

// 1 = launched (dw += the gradient), 0 = shape not covered or no workspace
int mscl_wgrad_thin(const mscl_conv_desc* d, const bf16_t* x, const bf16_t* dy, float* dw, float* ws, int64_t ws_floats, hipStream_t st) {
  ThinWGeom g;
  if (!thin_wgrad_geom(d, g)) return 0;
  const int nb = (g.units + g.upb - 1) / g.upb;
  const int n = 9 * d->K * d->C;
  if (ws == nullptr || ws_floats < (int64_t)nb * n) return 0;
  const size_t lds = ((size_t)g.QD * d->K + (size_t)g.XR * d->C) * 2;
#define THIN_WGO(CC, KK) hipLaunchKernelGGL((wgrad_thin_kernel<CC, KK>), dim3(nb), dim3(576), lds, st, g, x, dy, ws)
  if (d->C == 16 && d->K == 16) THIN_WGO(16, 16);
  else if (d->C == 16 && d->K == 32) THIN_WGO(16, 32);
  else if (d->C == 32 && d->K == 16) THIN_WGO(32, 16);
  else THIN_WGO(32, 32);
#undef THIN_WGO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, (const float*)ws, dw, nb, n);
  e = hipGetLastError();
  if (e != hipSuccess) return -(int)e;
  g_thin_wgrad_launches.fetch_add(1);
  return 1;
}
