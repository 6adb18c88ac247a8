// Conv3d weight gradient on bf16 MFMA (gfx950):  dW[co][tap][ci] += sum_m dy[m][co] * x[src(m,tap)][ci]
//
// GEMM view per tap: D[co][ci] = sum over positions m.  Both operands are stored position-major in HBM
// (NDHWC), i.e. the reduction index is the *row* of both tiles, so the MFMA fragments (8 consecutive
// reduction indices per lane) are column reads of the LDS images: ds_read_b64_tr_b16, CDNA4's
// transposing LDS read, delivers them without any data shuffling.
//
// Work split: one block = (co tile <=64) x (ci tile <=64 of ONE tap) x (a slice of the positions).
// The four waves of a block each take 32 of the 128 positions staged per step and keep a full
// co x ci accumulator tile; they are summed through LDS at the end and added to dW with one
// fp32 atomic per element per block (dW is caller-zeroed; repeated trunk traversals accumulate).
#include "common.h"

struct WGeom {
  int N, T, H, W, C;       // x
  int To, Ho, Wo, K;       // dy
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int M, ntaps;
  int co_tiles, ci_tiles, splits, per_split;   // per_split: positions per split (multiple of 128)
  FastDiv dWo, dHo, dTo;
};

template <int CH> __device__ __forceinline__ int wswz(int row) {
  if constexpr (CH == 64) return (row & 2) | ((row >> 1) & 4);    // XOR on the 16-byte granule index
  else return 0;
}

// CO, CI: channel extents of the block tile (16, 32 or 64)
template <int CO, int CI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WGeom g, const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ dy, float* __restrict__ dw) {
  constexpr int PB = 128;                       // positions staged per step
  constexpr int GA = CO / 8, GB = CI / 8;       // granules per row
  constexpr int PA = (PB * GA + 255) / 256, PBs = (PB * GB + 255) / 256;
  constexpr int IA = CO / 16, JB = CI / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* At = smem;                               // [2][PB][CO] bf16
  unsigned char* Bt = smem + 2 * PB * CO * 2;             // [2][PB][CI] bf16

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int split = bid % g.splits; bid /= g.splits;
  const int cit = bid % g.ci_tiles; bid /= g.ci_tiles;
  const int tap = bid % g.ntaps; const int cot = bid / g.ntaps;
  const int kw = tap % g.kW, kh = (tap / g.kW) % g.kH, kt = tap / (g.kW * g.kH);
  const int co0 = cot * CO, ci0 = cit * CI;
  const int mbeg = split * g.per_split;
  const int mend = min(g.M, mbeg + g.per_split);

  uint4 ra[PA], rb[PBs];
  auto load_tiles = [&](int mb) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int e = p * 256 + tid; const int r = e / GA, gq = e % GA;
      const int m = mb + r;
      const bool ok = (e < PB * GA) && (m < mend);
      const long off = ok ? ((long)m * g.K + co0 + gq * 8) : 0;
      uint4 v = *reinterpret_cast<const uint4*>(dy + off);
      ra[p] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int p = 0; p < PBs; ++p) {
      const int e = p * 256 + tid; const int r = e / GB, gq = e % GB;
      const int m = mb + r;
      bool ok = (e < PB * GB) && (m < mend) && (ci0 + gq * 8 < g.C);
      int q1 = fdiv(m, g.dWo); const int wo = m - q1 * g.Wo;
      int q2 = fdiv(q1, g.dHo); const int ho = q1 - q2 * g.Ho;
      const int n = fdiv(q2, g.dTo); const int to = q2 - n * g.To;
      const int ti = to * g.sT - g.pT + kt, hi = ho * g.sH - g.pH + kh, wi = wo * g.sW - g.pW + kw;
      ok = ok && (unsigned)ti < (unsigned)g.T && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      const long off = ok ? ((((long)(n * g.T + ti) * g.H + hi) * g.W + wi) * g.C + ci0 + gq * 8) : 0;
      uint4 v = *reinterpret_cast<const uint4*>(x + off);
      rb[p] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tiles = [&](int buf) {
    unsigned char* a = At + buf * PB * CO * 2;
    unsigned char* b = Bt + buf * PB * CI * 2;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int e = p * 256 + tid; const int r = e / GA, gq = e % GA;
      if (e < PB * GA) *reinterpret_cast<uint4*>(a + r * (CO * 2) + ((gq ^ wswz<CO>(r)) * 16)) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < PBs; ++p) {
      const int e = p * 256 + tid; const int r = e / GB, gq = e % GB;
      if (e < PB * GB) *reinterpret_cast<uint4*>(b + r * (CI * 2) + ((gq ^ wswz<CI>(r)) * 16)) = rb[p];
    }
  };

  f32x4_t acc[IA][JB];
#pragma unroll
  for (int i = 0; i < IA; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int nsteps = (mend > mbeg) ? (mend - mbeg + PB - 1) / PB : 0;
  if (nsteps > 0) { load_tiles(mbeg); store_tiles(0); }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) load_tiles(mbeg + (s + 1) * PB);
    const unsigned char* a = At + cur * PB * CO * 2;
    const unsigned char* b = Bt + cur * PB * CI * 2;
    // this wave's 32 positions: rows wave*32 .. +31 ; lane (grp,qq,pp) addresses row 8*grp + 4*h + qq
    bf16x8_t fa[IA], fb[JB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      s16x4_t v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = wave * 32 + 8 * grp + 4 * h + qq;
        const int gq = i * 2 + (pp >> 1);
        const unsigned char* ad = a + r * (CO * 2) + ((gq ^ wswz<CO>(r)) * 16) + (pp & 1) * 8;
        v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(ad));
      }
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      s16x8_t w8 = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
      fa[i] = __builtin_bit_cast(bf16x8_t, w8);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      s16x4_t v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = wave * 32 + 8 * grp + 4 * h + qq;
        const int gq = j * 2 + (pp >> 1);
        const unsigned char* bd = b + r * (CI * 2) + ((gq ^ wswz<CI>(r)) * 16) + (pp & 1) * 8;
        v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(bd));
      }
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      s16x8_t w8 = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
      fb[j] = __builtin_bit_cast(bf16x8_t, w8);
    }
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if (s + 1 < nsteps) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- cross-wave sum through LDS, then one atomic per element ----
  float* red = reinterpret_cast<float*>(smem);            // [CO][CI]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = i * 16 + (lane >> 4) * 4 + r, ci = j * 16 + (lane & 15);
            if (w == 0) red[co * CI + ci] = acc[i][j][r]; else red[co * CI + ci] += acc[i][j][r];
          }
    }
    __syncthreads();
  }
  const int KC = g.ntaps * g.C;
  for (int e = tid; e < CO * CI; e += 256) {
    const int co = e / CI, ci = e % CI;
    if (co0 + co < g.K && ci0 + ci < g.C && nsteps > 0)
      atomicAdd(&dw[(long)(co0 + co) * KC + (long)tap * g.C + ci0 + ci], red[e]);
  }
}

// column sums of a bf16 (rows, C) matrix into fp32 out[C] (+=): conv bias gradient
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ xx, float* __restrict__ out, long rows, int C) {
  const int G = C / 8;
  const int tg = threadIdx.x % G;                 // requires 256 % G == 0 (C/8 power of two <= 256)
  const int tr = threadIdx.x / G, RP = 256 / G;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long r = (long)blockIdx.x * RP + tr; r < rows; r += (long)gridDim.x * RP) {
    const uint4 v = *reinterpret_cast<const uint4*>(xx + r * C + tg * 8);
    float f[8]; unpack8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += f[i];
  }
  __shared__ float red[2048];
  for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) atomicAdd(&red[tg * 8 + i], s[i]);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&out[i], red[i]);
}

template <int CO, int CI>
static int launch_w(WGeom g, const bf16_t* x, const bf16_t* dy, float* dw, hipStream_t st) {
  g.co_tiles = (g.K + CO - 1) / CO;
  g.ci_tiles = (g.C + CI - 1) / CI;
  const long tiles = (long)g.co_tiles * g.ci_tiles * g.ntaps;
  long want = (1536 + tiles - 1) / tiles;                 // ~6 blocks per CU overall
  long maxs = (g.M + 127) / 128;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  long per = ((g.M + want - 1) / want + 127) / 128 * 128;
  g.per_split = (int)per;
  g.splits = (int)((g.M + per - 1) / per);
  const size_t lds = (size_t)2 * 128 * (CO + CI) * 2 > (size_t)CO * CI * 4 ? (size_t)2 * 128 * (CO + CI) * 2 : (size_t)CO * CI * 4;
  hipLaunchKernelGGL((conv_wgrad_kernel<CO, CI>), dim3((unsigned)(tiles * g.splits)), dim3(256), lds, st, g, x, dy, dw);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_conv3d_wgrad(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw,
                                 float* dbias, void* stream) {
  if (!d || !x || !dy || !dw) return MSCL_E_ARG;
  if (d->C % 8 || d->K % 8) return MSCL_E_SHAPE;
  if (ilog2_exact(d->K / 8) < 0 || d->K / 8 > 256) return MSCL_E_SHAPE;
  WGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.C = d->C;
  g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.K = d->K;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M >= (1L << 31) - 256 || (long)d->N * d->T * d->H * d->W * d->C >= (1L << 40)) return MSCL_E_SHAPE;
  g.M = (int)M; g.ntaps = d->kT * d->kH * d->kW;
  g.dWo = make_fastdiv(d->Wo); g.dHo = make_fastdiv(d->Ho); g.dTo = make_fastdiv(d->To);
  hipStream_t st = (hipStream_t)stream;
  int e;
  const int co = d->K >= 64 ? 64 : d->K, ci = d->C >= 64 ? 64 : (d->C < 16 ? 16 : d->C);   // C=8 (padded stems) rides the 16-wide tile
#define W(CO, CI) if (co == CO && ci == CI) { e = launch_w<CO, CI>(g, x, dy, dw, st); goto done; }
  W(64, 64) W(64, 32) W(64, 16) W(32, 64) W(32, 32) W(32, 16) W(16, 64) W(16, 32) W(16, 16)
#undef W
  return MSCL_E_SHAPE;   // channel counts must be 8/16/32 or a multiple of 64
done:
  if (e) return e;
  if (dbias) {
    long blocks = (M + 2047) / 2048; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, dbias, M, d->K);
    MSCL_LAUNCH_CHECK();
  }
  return 0;
}
