// Conv3d forward and input-gradient as ONE implicit-GEMM kernel family on bf16 MFMA (gfx950).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//   rows    m = positions of the "row tensor"  (forward: y positions; dgrad: dx positions)
//   columns n = channels of the row tensor     (forward: Cout;        dgrad: Cin)
//   depth   k = (tap, source channel)          gathered on the fly from the NDHWC "source tensor"
//                                               (forward: x; dgrad: dy) -- no im2col buffer.
// B is the kernel laid out [n][tap][source channel] (K-contiguous, like A), so both operands are read
// from LDS as 16-byte k-contiguous fragments of v_mfma_f32_16x16x32_bf16.
//
// MI355X mapping: 256 threads = 4 waves; block tile BM x BN, depth step BK (64 when the source channel
// count allows a whole 128-byte row per tap, else 32); double-buffered LDS with one barrier per step,
// next tile's global loads issued before the current tile's MFMAs (register staging, because padded
// taps must be zero-filled per 16-byte granule); LDS images XOR-swizzled so every ds_read_b128 lane
// group hits 16 distinct 16-byte slots; the product is computed transposed (weights as the MFMA A
// operand) so each lane ends up with 4 consecutive output channels of one position -> 8-byte stores;
// BatchNorm sum / sum-of-squares are reduced in the epilogue (wave shuffles -> LDS -> one atomic per
// channel per block); the 1-D grid is remapped so that an XCD's L2 sees neighbouring position tiles.
#include "common.h"

struct IGemmGeom {
  int N, Ts, Hs, Ws, Cs;   // source (gathered) tensor
  int Tr, Hr, Wr, Cr;      // row tensor
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int M, KG, cgs, ntaps;
  int lsT, lsH, lsW;
  int mode;                // 0 forward gather, 1 dgrad stride-1 (linear), 2 dgrad strided (generic)
  int mtiles, ntiles;
};

template <int BK> __device__ __forceinline__ int swz(int row, int g) {
  if constexpr (BK == 64) return g ^ ((row >> 1) & 7);
  else { const int q = (row >> 2) & 3; return g ^ ((0x78 >> (q * 2)) & 3); }   // q -> {0,2,3,1}
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const IGemmGeom g, const bf16_t* __restrict__ src, const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
    const float* __restrict__ bias, const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
    float* __restrict__ stat_sq, const int relu) {
  constexpr int GPR = BK / 8;                 // 16-byte granules per tile row
  constexpr int RPP = 256 / GPR;              // rows covered per staging pass
  constexpr int AP = BM / RPP;                // A staging passes
  constexpr int BP = (BN + RPP - 1) / RPP;    // B staging passes
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int IM = WM / 16, JN = WN / 16;
  constexpr int KSUB = BK / 32;
  static_assert(WAVES_M * WAVES_N == 4 && IM >= 1 && JN >= 1 && AP >= 1, "tile config");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);                       // [2][BM*BK]
  bf16_t* Bs = As + 2 * BM * BK;                                      // [2][BN*BK]
  int* tap_delta = reinterpret_cast<int*>(Bs + 2 * BN * BK);          // [ntaps]
  int* tap_bits = tap_delta + g.ntaps;                                // [ntaps]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = bid % g.ntiles, mt = bid / g.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- tap tables ----
  for (int t = tid; t < g.ntaps; t += 256) {
    const int kw = t % g.kW, kh = (t / g.kW) % g.kH, kt = t / (g.kW * g.kH);
    const int lin = (kt * g.Hs + kh) * g.Ws + kw;
    tap_delta[t] = (g.mode == 0) ? lin : ((g.mode == 1) ? -lin : (kt | (kh << 8) | (kw << 16)));
    tap_bits[t] = (1 << kt) | (1 << (8 + kh)) | (1 << (16 + kw));
  }

  // ---- per-row gather state (one row per staging pass) ----
  const int rg = tid % GPR;                   // this thread's granule column within a tile row
  const int rr = tid / GPR;                   // row within a pass
  int row_base[AP], row_mask[AP], row_aux[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int m = m0 + p * RPP + rr;
    int mask = 0, base = 0, aux = 0;
    if (m < g.M) {
      const int wr = m % g.Wr; int r = m / g.Wr;
      const int hr = r % g.Hr; r /= g.Hr;
      const int tr = r % g.Tr; const int n = r / g.Tr;
      int t0, h0, w0;
      if (g.mode == 0) { t0 = tr * g.sT - g.pT; h0 = hr * g.sH - g.pH; w0 = wr * g.sW - g.pW; }
      else { t0 = tr + g.pT; h0 = hr + g.pH; w0 = wr + g.pW; }
      for (int k = 0; k < g.kT; ++k) {
        const int d = (g.mode == 0) ? t0 + k : t0 - k;
        const bool ok = (g.mode == 2) ? (d >= 0 && (d & (g.sT - 1)) == 0 && (d >> g.lsT) < g.Ts) : ((unsigned)d < (unsigned)g.Ts);
        mask |= ok ? (1 << k) : 0;
      }
      for (int k = 0; k < g.kH; ++k) {
        const int d = (g.mode == 0) ? h0 + k : h0 - k;
        const bool ok = (g.mode == 2) ? (d >= 0 && (d & (g.sH - 1)) == 0 && (d >> g.lsH) < g.Hs) : ((unsigned)d < (unsigned)g.Hs);
        mask |= ok ? (1 << (8 + k)) : 0;
      }
      for (int k = 0; k < g.kW; ++k) {
        const int d = (g.mode == 0) ? w0 + k : w0 - k;
        const bool ok = (g.mode == 2) ? (d >= 0 && (d & (g.sW - 1)) == 0 && (d >> g.lsW) < g.Ws) : ((unsigned)d < (unsigned)g.Ws);
        mask |= ok ? (1 << (16 + k)) : 0;
      }
      if (g.mode == 2) { base = n * g.Ts; aux = t0 | (h0 << 10) | (w0 << 20); }
      else base = ((n * g.Ts + t0) * g.Hs + h0) * g.Ws + w0;
    }
    row_base[p] = base; row_mask[p] = mask; row_aux[p] = aux;
  }
  __syncthreads();   // tap tables visible

  uint4 ra[AP], rb[BP];
  const int cmask = (1 << g.cgs) - 1;

  auto load_tiles = [&](int kt) {
    const int kg = kt * GPR + rg;
    const bool kin = kg < g.KG;
    const int tap = kin ? (kg >> g.cgs) : 0;
    const int c0 = (kg & cmask) << 3;
    const int td = tap_delta[tap], tb = tap_bits[tap];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const bool ok = kin && ((row_mask[p] & tb) == tb);
      int pos;
      if (g.mode == 2) {
        const int a = row_aux[p];
        const int dt = ((a & 1023) - (td & 255)) >> g.lsT;
        const int dh = (((a >> 10) & 1023) - ((td >> 8) & 255)) >> g.lsH;
        const int dw = (((a >> 20) & 1023) - ((td >> 16) & 255)) >> g.lsW;
        pos = ((row_base[p] + dt) * g.Hs + dh) * g.Ws + dw;
      } else {
        pos = row_base[p] + td;
      }
      const int off = ok ? (pos * g.Cs + c0) : 0;
      uint4 v = *reinterpret_cast<const uint4*>(src + off);
      ra[p] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int r = p * RPP + rr;
      const bool ok = kin && (r < BN) && (n0 + r < g.Cr);
      const long off = ok ? ((long)(n0 + r) * g.KG + kg) * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(wgt + off);
      rb[p] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tiles = [&](int buf) {
    unsigned char* a = reinterpret_cast<unsigned char*>(As + buf * BM * BK);
    unsigned char* b = reinterpret_cast<unsigned char*>(Bs + buf * BN * BK);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int r = p * RPP + rr;
      *reinterpret_cast<uint4*>(a + r * (BK * 2) + swz<BK>(r, rg) * 16) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int r = p * RPP + rr;
      if (r < BN) *reinterpret_cast<uint4*>(b + r * (BK * 2) + swz<BK>(r, rg) * 16) = rb[p];
    }
  };

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4_t acc[JN][IM];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < IM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.KG + GPR - 1) / GPR;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const unsigned char* a = reinterpret_cast<const unsigned char*>(As + cur * BM * BK);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(Bs + cur * BN * BK);
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      bf16x8_t fa[IM], fb[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        const int r = wm0 + i * 16 + fr;
        fa[i] = *reinterpret_cast<const bf16x8_t*>(a + r * (BK * 2) + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int r = wn0 + j * 16 + fr;
        fb[j] = *reinterpret_cast<const bf16x8_t*>(b + r * (BK * 2) + swz<BK>(r, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < IM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: BatchNorm statistics of the raw fp32 result ----
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [2][BN], tiles are dead after the last barrier
    for (int i = tid; i < 2 * BN; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[j][i][r]; s[r] += v; q[r] += v * v; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s[r] += __shfl_xor(s[r], o, 64); q[r] += __shfl_xor(q[r], o, 64); }
      }
      if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = wn0 + j * 16 + fq * 4 + r;
          atomicAdd(&red[nl], s[r]);
          atomicAdd(&red[BN + nl], q[r]);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < BN; i += 256) {
      if (n0 + i < g.Cr) { atomicAdd(&stat_sum[n0 + i], red[i]); atomicAdd(&stat_sq[n0 + i], red[BN + i]); }
    }
  }

  // ---- epilogue: (+bias) (+addend) (relu) -> bf16, 4 consecutive channels per lane ----
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int m = m0 + wm0 + i * 16 + fr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int n = n0 + wn0 + j * 16 + fq * 4;
      if (n >= g.Cr) continue;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      const long o = (long)m * g.Cr + n;
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
      *reinterpret_cast<uint2*>(out + o) = pv;
    }
  }
}

// ---------------------------------------------------------------------------------------- host side
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_cfg(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias,
                      const bf16_t* addend, float* ssum, float* ssq, int relu, hipStream_t st) {
  g.mtiles = (g.M + BM - 1) / BM;
  g.ntiles = (g.Cr + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * BK * 2 + (size_t)g.ntaps * 8;
  auto kern = conv_igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.mtiles * g.ntiles), dim3(256), lds, st, g, src, wgt, out, bias, addend, ssum, ssq, relu);
  MSCL_LAUNCH_CHECK();
  return 0;
}

static int launch_igemm(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias,
                        const bf16_t* addend, float* ssum, float* ssq, int relu, hipStream_t st) {
  if ((long)g.N * g.Ts * g.Hs * g.Ws * g.Cs >= (1L << 31) || (long)g.M * g.Cr >= (1L << 31)) return MSCL_E_SHAPE;
  const bool bk64 = (g.Cs % 64) == 0;
  const int Cr = g.Cr;
  auto blocks = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((Cr + bn - 1) / bn); };
#define GO(BM, BN, BK, WMv, WNv) return launch_cfg<BM, BN, BK, WMv, WNv>(g, src, wgt, out, bias, addend, ssum, ssq, relu, st)
  if (bk64) {
    if (Cr >= 128) {
      if (blocks(128, 128) >= 384) GO(128, 128, 64, 2, 2);
      GO(64, 128, 64, 2, 2);
    }
    if (Cr > 32) {
      if (blocks(256, 64) >= 512) GO(256, 64, 64, 4, 1);
      if (blocks(128, 64) >= 384) GO(128, 64, 64, 2, 2);
      GO(64, 64, 64, 2, 2);
    }
    if (Cr > 16) GO(256, 32, 64, 4, 1);
    GO(256, 16, 64, 4, 1);
  }
  if (Cr >= 128) GO(128, 128, 32, 2, 2);
  if (Cr > 32) GO(256, 64, 32, 4, 1);
  if (Cr > 16) GO(256, 32, 32, 4, 1);
  GO(256, 16, 32, 4, 1);
#undef GO
}

static int check_desc(const mscl_conv_desc* d) {
  if (!d) return MSCL_E_ARG;
  if (d->N <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return MSCL_E_ARG;
  if (d->C % 8 || d->K % 8) return MSCL_E_SHAPE;
  if (ilog2_exact(d->C / 8) < 0 || ilog2_exact(d->K / 8) < 0) return MSCL_E_SHAPE;
  if (d->kT < 1 || d->kH < 1 || d->kW < 1 || d->kT > 8 || d->kH > 8 || d->kW > 8) return MSCL_E_SHAPE;
  if (d->To != (d->T + 2 * d->pT - d->kT) / d->sT + 1 || d->Ho != (d->H + 2 * d->pH - d->kH) / d->sH + 1 ||
      d->Wo != (d->W + 2 * d->pW - d->kW) / d->sW + 1) return MSCL_E_SHAPE;
  if (d->T + d->pT >= 1000 || d->H + d->pH >= 1000 || d->W + d->pW >= 1000) return MSCL_E_SHAPE;
  return 0;
}

extern "C" int mscl_conv3d_fwd(const mscl_conv_desc* d, const uint16_t* x, const uint16_t* w, uint16_t* y,
                               const float* bias, const uint16_t* addend, int relu, float* ssum, float* ssq,
                               void* stream) {
  int e = check_desc(d); if (e) return e;
  if (!x || !w || !y) return MSCL_E_ARG;
  if ((ssum == nullptr) != (ssq == nullptr)) return MSCL_E_ARG;
  IGemmGeom g{};
  g.N = d->N; g.Ts = d->T; g.Hs = d->H; g.Ws = d->W; g.Cs = d->C;
  g.Tr = d->To; g.Hr = d->Ho; g.Wr = d->Wo; g.Cr = d->K;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  g.M = d->N * d->To * d->Ho * d->Wo; g.ntaps = d->kT * d->kH * d->kW;
  g.cgs = ilog2_exact(d->C / 8); g.KG = g.ntaps * (d->C / 8); g.mode = 0;
  return launch_igemm(g, x, w, y, bias, addend, ssum, ssq, relu, (hipStream_t)stream);
}

extern "C" int mscl_conv3d_dgrad(const mscl_conv_desc* d, const uint16_t* dy, const uint16_t* wT, uint16_t* dx,
                                 const uint16_t* addend, void* stream) {
  int e = check_desc(d); if (e) return e;
  if (!dy || !wT || !dx) return MSCL_E_ARG;
  IGemmGeom g{};
  g.N = d->N; g.Ts = d->To; g.Hs = d->Ho; g.Ws = d->Wo; g.Cs = d->K;
  g.Tr = d->T; g.Hr = d->H; g.Wr = d->W; g.Cr = d->C;
  g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.sT = d->sT; g.sH = d->sH; g.sW = d->sW;
  g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  g.M = d->N * d->T * d->H * d->W; g.ntaps = d->kT * d->kH * d->kW;
  g.cgs = ilog2_exact(d->K / 8); g.KG = g.ntaps * (d->K / 8);
  const bool unit = d->sT == 1 && d->sH == 1 && d->sW == 1;
  g.mode = unit ? 1 : 2;
  if (!unit) {
    g.lsT = ilog2_exact(d->sT); g.lsH = ilog2_exact(d->sH); g.lsW = ilog2_exact(d->sW);
    if (g.lsT < 0 || g.lsH < 0 || g.lsW < 0) return MSCL_E_STRIDE;
  }
  return launch_igemm(g, dy, wT, dx, nullptr, addend, nullptr, nullptr, 0, (hipStream_t)stream);
}

__global__ void weight_transpose_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wT, int Cout, int taps, int Cin) {
  const long total = (long)Cout * taps * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout); const long r = i / Cout;       // i indexes wT[ci][tap][co]
    const int tap = (int)(r % taps); const int ci = (int)(r / taps);
    wT[i] = w[((long)co * taps + tap) * Cin + ci];
  }
}

extern "C" int mscl_weight_transpose(const uint16_t* w, uint16_t* wT, int Cout, int taps, int Cin, void* stream) {
  if (!w || !wT || Cout <= 0 || taps <= 0 || Cin <= 0) return MSCL_E_ARG;
  const long total = (long)Cout * taps * Cin;
  const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wT, Cout, taps, Cin);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_abi_version(void) { return 1; }
