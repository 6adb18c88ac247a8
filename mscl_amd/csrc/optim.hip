// Parameter-sized elementwise passes over the flat fp32 parameter arena (gfx950, HBM-bound):
// key-encoder EMA, gradient sum of squares, clip + SGD-momentum, bf16 shadow refresh.
#include "common.h"
#include <cstdlib>
static long ema_cap() { return 4096; }       // grid cap of the EMA / SGD passes (swept inside the step: no gain from 2048 / 8192)

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pk, const float* __restrict__ pq,
                                                  bf16_t* __restrict__ pkb, long n, float m_val, const float* __restrict__ m_dev) {
  const float m = m_dev ? *m_dev : m_val;
  const float om = 1.0f - m;
  const long n4 = n >> 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    float4 k = reinterpret_cast<float4*>(pk)[e];
    const float4 q = reinterpret_cast<const float4*>(pq)[e];
    k.x = k.x * m + q.x * om; k.y = k.y * m + q.y * om; k.z = k.z * m + q.z * om; k.w = k.w * m + q.w * om;
    reinterpret_cast<float4*>(pk)[e] = k;
    if (pkb) { uint2 b; b.x = pack2bf(k.x, k.y); b.y = pack2bf(k.z, k.w); reinterpret_cast<uint2*>(pkb)[e] = b; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long e = (n4 << 2) + threadIdx.x;
    const float k = pk[e] * m + pq[e] * om; pk[e] = k; if (pkb) pkb[e] = f2bf(k);
  }
}
extern "C" int mscl_ema_update(float* pk, const float* pq, uint16_t* pk_bf16, int64_t n, float m, void* stream) {
  if (!pk || !pq || n <= 0) return MSCL_E_ARG;
  if (((uintptr_t)pk | (uintptr_t)pq) & 15 || ((uintptr_t)pk_bf16 & 7)) return MSCL_E_SHAPE;
  long blocks = (n / 4 + 255) / 256; if (blocks > ema_cap()) blocks = ema_cap(); if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk, pq, pk_bf16, (long)n, m,
                     (const float*)nullptr);
  MSCL_LAUNCH_CHECK();
  return 0;
}
extern "C" int mscl_ema_update_dev(float* pk, const float* pq, uint16_t* pk_bf16, int64_t n, const float* m_dev, void* stream) {
  if (!pk || !pq || !m_dev || n <= 0) return MSCL_E_ARG;
  if (((uintptr_t)pk | (uintptr_t)pq) & 15 || ((uintptr_t)pk_bf16 & 7)) return MSCL_E_SHAPE;
  long blocks = (n / 4 + 255) / 256; if (blocks > ema_cap()) blocks = ema_cap(); if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk, pq, pk_bf16, (long)n, 0.f, m_dev);
  MSCL_LAUNCH_CHECK();
  return 0;
}

/* Deterministic two-phase sum of squares: every data-parallel replica must derive the SAME clip coefficient from the
 * same all-reduced gradient, or the replicas drift apart (there is no parameter broadcast after step 0).  Phase 1
 * writes one partial per block (fixed grid-stride order), phase 2 adds the partials in a fixed tree -- no atomics. */
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ partial, long n) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[e];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)red[0];
}
extern "C" int mscl_sumsq(const float* g, float* out, int64_t n, float* partials, int n_partials, void* stream) {
  if (!g || !out || !partials || n <= 0 || n_partials < 1) return MSCL_E_ARG;
  if ((uintptr_t)g & 15) return MSCL_E_SHAPE;
  long blocks = (n / 4 + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  if (blocks > n_partials) blocks = n_partials;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, partials, (long)n);
  MSCL_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, (int)blocks, out);
  MSCL_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  bf16_t* __restrict__ pb, long n, const float* __restrict__ sumsq,
                                                  float max_norm, float lr_val, float mom, float wd, int first,
                                                  const float* __restrict__ lr_dev) {
  const float lr = lr_dev ? *lr_dev : lr_val;
  float coef = 1.f;
  if (max_norm > 0.f && sumsq) { const float tn = sqrtf(*sumsq); coef = fminf(max_norm / (tn + 1e-6f), 1.f); }
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float pv = p[e];
    const float d = g[e] * coef + wd * pv;
    const float b = first ? d : mom * buf[e] + d;
    buf[e] = b;
    const float np = pv - lr * b;
    p[e] = np;
    if (pb) pb[e] = f2bf(np);
  }
}
extern "C" int mscl_sgd_step(float* p, const float* g, float* buf, uint16_t* p_bf16, int64_t n, const float* sumsq,
                             float max_norm, float lr, float momentum, float wd, int first, void* stream) {
  if (!p || !g || !buf || n <= 0) return MSCL_E_ARG;
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, p_bf16, (long)n, sumsq,
                     max_norm, lr, momentum, wd, first, (const float*)nullptr);
  MSCL_LAUNCH_CHECK();
  return 0;
}
/* lr read from device memory (graph replay with a changing schedule); momentum buffers must start at zero
 * (buf = mom*0 + d reproduces torch.optim.SGD's first-step buffer initialisation exactly) */
extern "C" int mscl_sgd_step_dev(float* p, const float* g, float* buf, uint16_t* p_bf16, int64_t n, const float* sumsq,
                                 float max_norm, const float* lr_dev, float momentum, float wd, void* stream) {
  if (!p || !g || !buf || !lr_dev || n <= 0) return MSCL_E_ARG;
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, p_bf16, (long)n, sumsq,
                     max_norm, 0.f, momentum, wd, 0, lr_dev);
  MSCL_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) d[e] = f2bf(s[e]);
}
extern "C" int mscl_cast_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return MSCL_E_ARG;
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  MSCL_LAUNCH_CHECK();
  return 0;
}
