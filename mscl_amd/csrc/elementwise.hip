// Layout conversion, pooling, resize-add, small dense layers (gfx950).  All HBM- or launch-bound:
// 16-byte accesses along the channel axis, one wave per output where a reduction is needed.
#include "common.h"
#include <cstdlib>
// grid cap of the elementwise passes (MSCL_EW_CAP: tuning aid)
static long ew_cap() { return 2048; }        // grid cap of the element-wise passes (swept inside the step: 1024 / 2048 / 4096, no gain)

// ---------------------------------------------------------------- input packing NCTHW fp32 -> NDHWC8 bf16
// frames [t_off, t_off+T) of a clip holding T_total frames (the base / rotated halves of the flow clip,
// recognizers/mscl.py:230-235, are packed straight from the concatenated tensor)
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B,
                                                         int Cin, long THW, long THW_total, long off, float m0, float m1,
                                                         float m2, float i0, float i1, float i2,
                                                         const unsigned char* __restrict__ flip, int W,
                                                         const float* const* __restrict__ xind) {
  // xind != NULL: the clip's address is read from device memory (mscl_pack_input_ind: a captured launch whose input changes per replay)
  if (xind != nullptr) x = *xind;
  // blockIdx.y = sample: no per-element division
  const long b = blockIdx.y;
  const bool fl = flip != nullptr && flip[b];
  for (long p0 = (long)blockIdx.x * blockDim.x + threadIdx.x; p0 < THW; p0 += (long)gridDim.x * blockDim.x) {
    const long e = b * THW + p0;
    long p = p0;
    if (fl) { const long row = p / W; p = row * W + (W - 1 - (p - row * W)); }   // torch.flip(x, [-1])
    const float* xb = x + b * Cin * THW_total + off + p;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f[0] = (xb[0] - m0) * i0;
    if (Cin > 1) f[1] = (xb[THW_total] - m1) * i1;
    if (Cin > 2) f[2] = (xb[2 * THW_total] - m2) * i2;
    *reinterpret_cast<uint4*>(out + e * 8) = pack8(f);
  }
}

extern "C" int mscl_pack_input(const float* x, uint16_t* out, int B, int Cin, int T, int H, int W, int T_total, int t_off,
                               const float* mean3, const float* std3, const uint8_t* flip_mask, void* stream) {
  if (!x || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || t_off < 0 || t_off + T > T_total) return MSCL_E_ARG;
  if (Cin < 1 || Cin > 3) return MSCL_E_SHAPE;
  float m[3] = {0, 0, 0}, iv[3] = {1, 1, 1};
  if (mean3 && std3) for (int i = 0; i < 3; ++i) { m[i] = mean3[i]; iv[i] = 1.f / std3[i]; }   // host arrays
  const long THW = (long)T * H * W;
  long blocks = (THW + 255) / 256; if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, out, B, Cin, THW,
                     (long)T_total * H * W, (long)t_off * H * W, m[0], m[1], m[2], iv[0], iv[1], iv[2], flip_mask, W,
                     (const float* const*)nullptr);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_pack_input_ind(const float* const* xpp, uint16_t* out, int B, int Cin, int T, int H, int W, int T_total, int t_off,
                                   const float* mean3, const float* std3, const uint8_t* flip_mask, void* stream) {
  if (!xpp || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || t_off < 0 || t_off + T > T_total) return MSCL_E_ARG;
  if (Cin < 1 || Cin > 3) return MSCL_E_SHAPE;
  float m[3] = {0, 0, 0}, iv[3] = {1, 1, 1};
  if (mean3 && std3) for (int i = 0; i < 3; ++i) { m[i] = mean3[i]; iv[i] = 1.f / std3[i]; }
  const long THW = (long)T * H * W;
  long blocks = (THW + 255) / 256; if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, out, B,
                     Cin, THW, (long)T_total * H * W, (long)t_off * H * W, m[0], m[1], m[2], iv[0], iv[1], iv[2], flip_mask, W, xpp);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- W-pairing of a packed 3-channel clip (RGB stem)
// (rows, W, 8) -> (rows, (W+1)/2 + 1, 8): pair j holds pixels w = 2j - 1 and 2j of its row as channels [3p + c] (p = 0, 1; channels
// 6, 7 zero; pixels outside the row zero).  A 7-wide stride-2 pad-3 convolution along W over 3 (padded to 8) channels is then
// a 4-wide stride-1 pad-1 convolution over the pairs -- kw = 2j' + p -- with the SAME outputs: 4 x 8 = 32 reduction slots per
// (kt, kh) instead of 7 x 8 = 56, 21 of them live either way.  The stem's implicit GEMM shrinks from K = 1176 to 672.
__global__ __launch_bounds__(256) void pair_w_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long rows, int W, int Wp) {
  const long total = rows * Wp;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / Wp; const int j = (int)(e - r * Wp);
    const int w0 = 2 * j - 1, w1 = 2 * j;
    uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
    if (w0 >= 0 && w0 < W) a = *reinterpret_cast<const uint4*>(x + (r * W + w0) * 8);
    if (w1 < W) b = *reinterpret_cast<const uint4*>(x + (r * W + w1) * 8);
    // bf16 channels: a = [a0 a1 | a2 .. ], b likewise; out = [a0 a1 | a2 b0 | b1 b2 | 0 0]
    uint4 o;
    o.x = a.x;
    o.y = (a.y & 0xFFFFu) | (b.x << 16);
    o.z = (b.x >> 16) | (b.y << 16);
    o.w = 0u;
    *reinterpret_cast<uint4*>(out + e * 8) = o;
  }
}
extern "C" int mscl_pair_w(const uint16_t* x, uint16_t* out, int64_t rows, int W, void* stream) {
  if (!x || !out || rows <= 0 || W <= 0) return MSCL_E_ARG;
  const int Wp = (W + 1) / 2 + 1;          // odd W: one more (all-zero) pair so that the paired conv yields (W + 1) / 2 outputs
  const long total = rows * Wp;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pair_w_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)rows, W, Wp);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- optical flow (u, v) -> colour wheel image
// ref: common/ssl_aug.py:87-136 (flow_uv_to_colors / FlowVisualizer), colour wheel tools/RAFT/core/utils/flow_viz.py:19-68.
// The reference mixes precisions and this kernel follows it operation by operation: radius, angle and the wheel
// position fk are float32; the colour interpolation runs in float64 (the wheel is a float64 numpy array); the result is
// floor(255 * col) as a byte, then byte / 255 in float32.  No FMA contraction where the reference rounds twice.
__device__ const unsigned char kColorWheel[55][3] = {
  {255,0,0},{255,17,0},{255,34,0},{255,51,0},{255,68,0},{255,85,0},{255,102,0},{255,119,0},{255,136,0},{255,153,0},{255,170,0},
  {255,187,0},{255,204,0},{255,221,0},{255,238,0},{255,255,0},{213,255,0},{170,255,0},{128,255,0},{85,255,0},{43,255,0},{0,255,0},
  {0,255,63},{0,255,127},{0,255,191},{0,255,255},{0,232,255},{0,209,255},{0,186,255},{0,163,255},{0,140,255},{0,116,255},{0,93,255},
  {0,70,255},{0,47,255},{0,24,255},{0,0,255},{19,0,255},{39,0,255},{58,0,255},{78,0,255},{98,0,255},{117,0,255},{137,0,255},
  {156,0,255},{176,0,255},{196,0,255},{215,0,255},{235,0,255},{255,0,255},{255,0,213},{255,0,170},{255,0,128},{255,0,85},{255,0,43}};

// colour-wheel levels of one flow vector, operation by operation as the reference computes them (see above)
__device__ __forceinline__ void flow_uv_to_levels(float u, float v, unsigned char* lv) {
  const float rad = __fsqrt_rn(__fadd_rn(__fmul_rn(u, u), __fmul_rn(v, v)));
  const float a = __fdiv_rn(atan2f(-v, -u), 3.14159265358979323846f);
  const float fk = __fmul_rn(__fdiv_rn(__fadd_rn(a, 1.0f), 2.0f), 54.0f);
  const float k0f = floorf(fk);
  int k0 = (int)k0f, k1 = k0 + 1;
  if (k1 == 55) k1 = 0;
  const float f = __fsub_rn(fk, k0f);
  const double w0 = (double)__fsub_rn(1.0f, f), w1 = (double)f, radd = (double)rad;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double col0 = (double)kColorWheel[k0][i] / 255.0, col1 = (double)kColorWheel[k1][i] / 255.0;
    double col = __dadd_rn(__dmul_rn(w0, col0), __dmul_rn(w1, col1));
    if (rad <= 1.0f) col = __dsub_rn(1.0, __dmul_rn(radd, __dsub_rn(1.0, col)));
    else col = __dmul_rn(col, 0.75);
    lv[i] = (unsigned char)(int)floor(__dmul_rn(255.0, col));
  }
}

__global__ __launch_bounds__(256) void flow_visualize_kernel(const float* __restrict__ uv, bf16_t* __restrict__ out,
                                                             unsigned char* __restrict__ levels, int B, long THW, long THW_total,
                                                             long off, const unsigned char* __restrict__ flip, int W) {
  const long total = (long)B * THW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / THW;
    long p = e - b * THW;
    if (flip != nullptr && flip[b]) { const long row = p / W; p = row * W + (W - 1 - (p - row * W)); }   // flip of the IMAGE (ssl_aug_v2.py:118)
    const float* ub = uv + b * 2 * THW_total + off + p;
    unsigned char lv[3];
    flow_uv_to_levels(ub[0], ub[THW_total], lv);
    float c3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (levels != nullptr) levels[e * 3 + i] = lv[i];
      c3[i] = __fdiv_rn((float)lv[i], 255.0f);
    }
    *reinterpret_cast<uint4*>(out + e * 8) = pack8(c3);
  }
}

extern "C" int mscl_flow_visualize(const float* uv, uint16_t* out, uint8_t* levels, int B, int T, int H, int W, int T_total,
                                   int t_off, const uint8_t* flip_mask, void* stream) {
  if (!uv || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || t_off < 0 || t_off + T > T_total) return MSCL_E_ARG;
  const long THW = (long)T * H * W, total = (long)B * THW;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flow_visualize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, uv, out, levels, B, THW,
                     (long)T_total * H * W, (long)t_off * H * W, flip_mask, W);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- Flow Rotation Augmentation + visualiser, fused
// ref: datasets/pipelines/transforms_motion.py:7-29,103-142 (NormFlowWithStidedAug as configured at
// mscl_r18_cosm_lr2e-2.py:70,82) followed by FlowVisualizer: every frame is divided by (its max radius + 1e-5); a second
// copy is first rotated by beta = (start + stride * cid) * pi.  The reference mixes precisions (NumPy 2 rules: the base
// copy stays float32, the rotated copy is float64 because sin/cos are float64 scalars); both are followed here.
// Output: frames [0, T) = base, [T, 2T) = rotated (merge_aug=True), as the colour image the flow trunk consumes.
__global__ __launch_bounds__(256) void fra_maxrad_kernel(const float* __restrict__ uv, const int* __restrict__ cid, double start,
                                                         double stride, double* __restrict__ maxrad, int T, long HW) {
  const int bt = blockIdx.x, b = bt / T, t = bt - b * T;
  const float* u = uv + ((long)b * 2 * T + t) * HW;
  const float* v = u + (long)T * HW;
  const double beta = (start + stride * (double)cid[b]) * 3.141592653589793;
  const double sb = sin(beta), cb = cos(beta);
  float m0 = 0.f; double m1 = 0.0;
  for (long p = threadIdx.x; p < HW; p += 256) {
    const float uu = u[p], vv = v[p];
    m0 = fmaxf(m0, __fsqrt_rn(__fadd_rn(__fmul_rn(uu, uu), __fmul_rn(vv, vv))));
    const double nu = __dsub_rn(__dmul_rn(cb, (double)uu), __dmul_rn(sb, (double)vv));
    const double nv = __dadd_rn(__dmul_rn(sb, (double)uu), __dmul_rn(cb, (double)vv));
    m1 = fmax(m1, __dsqrt_rn(__dadd_rn(__dmul_rn(nu, nu), __dmul_rn(nv, nv))));
  }
  __shared__ double red0[256], red1[256];
  red0[threadIdx.x] = (double)m0; red1[threadIdx.x] = m1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { red0[threadIdx.x] = fmax(red0[threadIdx.x], red0[threadIdx.x + w]); red1[threadIdx.x] = fmax(red1[threadIdx.x], red1[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { maxrad[bt * 2] = red0[0]; maxrad[bt * 2 + 1] = red1[0]; }
}

__global__ __launch_bounds__(256) void fra_visualize_kernel(const float* __restrict__ uv, const int* __restrict__ cid, double start,
                                                            double stride, const double* __restrict__ maxrad,
                                                            bf16_t* __restrict__ out, unsigned char* __restrict__ levels,
                                                            float* __restrict__ normed, int B, int T, long HW,
                                                            const unsigned char* __restrict__ flip, int W) {
  const long total = (long)B * 2 * T * HW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / (2 * T * HW);
    const long r = e - b * 2 * T * HW;
    const int t2 = (int)(r / HW);
    long p = r - (long)t2 * HW;
    if (flip != nullptr && flip[b]) { const long row = p / W; p = row * W + (W - 1 - (p - row * W)); }
    const int t = t2 < T ? t2 : t2 - T;
    const float uu = uv[((long)b * 2 * T + t) * HW + p], vv = uv[((long)b * 2 * T + T + t) * HW + p];
    float un, vn;
    if (t2 < T) {
      const float den = __fadd_rn((float)maxrad[(b * T + t) * 2], 1e-5f);
      un = __fdiv_rn(uu, den); vn = __fdiv_rn(vv, den);
    } else {
      const double beta = (start + stride * (double)cid[b]) * 3.141592653589793;
      const double sb = sin(beta), cb = cos(beta);
      const double nu = __dsub_rn(__dmul_rn(cb, (double)uu), __dmul_rn(sb, (double)vv));
      const double nv = __dadd_rn(__dmul_rn(sb, (double)uu), __dmul_rn(cb, (double)vv));
      const double den = __dadd_rn(maxrad[(b * T + t) * 2 + 1], 1e-5);
      un = (float)__ddiv_rn(nu, den); vn = (float)__ddiv_rn(nv, den);
    }
    if (normed != nullptr) { normed[((long)b * 2 * T + t2) * HW * 2 + (r - (long)t2 * HW) * 2] = un; normed[((long)b * 2 * T + t2) * HW * 2 + (r - (long)t2 * HW) * 2 + 1] = vn; }
    unsigned char lv[3];
    flow_uv_to_levels(un, vn, lv);
    float c3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (levels != nullptr) levels[e * 3 + i] = lv[i];
      c3[i] = __fdiv_rn((float)lv[i], 255.0f);
    }
    *reinterpret_cast<uint4*>(out + e * 8) = pack8(c3);
  }
}

extern "C" int mscl_flow_fra_visualize(const float* uv, const int32_t* cid, float ratio_lo, float ratio_hi, int num_chunks,
                                       uint16_t* out, uint8_t* levels, float* normed, double* scratch, int B, int T, int H, int W,
                                       const uint8_t* flip_mask, void* stream) {
  if (!uv || !cid || !out || !scratch || B <= 0 || T <= 0 || H <= 0 || W <= 0 || num_chunks <= 0) return MSCL_E_ARG;
  const double start = (double)ratio_lo, stride = ((double)ratio_hi - (double)ratio_lo) / num_chunks;
  const long HW = (long)H * W;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fra_maxrad_kernel, dim3((unsigned)(B * T)), dim3(256), 0, st, uv, cid, start, stride, scratch, T, HW);
  MSCL_LAUNCH_CHECK();
  const long total = (long)B * 2 * T * HW;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fra_visualize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, uv, cid, start, stride,
                     (const double*)scratch, out, levels, normed, B, T, HW, flip_mask, W);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- add / relu
__global__ __launch_bounds__(256) void add_relu_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       const bf16_t* __restrict__ c, bf16_t* __restrict__ out, long n8,
                                                       int relu) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
    float f[8]; unpack8(*reinterpret_cast<const uint4*>(a + e * 8), f);
    if (b) { float g[8]; unpack8(*reinterpret_cast<const uint4*>(b + e * 8), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += g[i]; }
    if (c) { float g[8]; unpack8(*reinterpret_cast<const uint4*>(c + e * 8), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += g[i]; }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f); }
    *reinterpret_cast<uint4*>(out + e * 8) = pack8(f);
  }
}
extern "C" int mscl_add_relu(const uint16_t* a, const uint16_t* b, const uint16_t* c, uint16_t* out, int64_t n, int relu,
                             void* stream) {
  if (!a || !out || n <= 0) return MSCL_E_ARG;
  if (n % 8) return MSCL_E_SHAPE;
  long blocks = (n / 8 + 255) / 256; if (blocks > ew_cap()) blocks = ew_cap();
  hipLaunchKernelGGL(add_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, out, (long)(n / 8), relu);
  MSCL_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                       bf16_t* __restrict__ din, long n8) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
    float d[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(dout + e * 8), d);
    unpack8(*reinterpret_cast<const uint4*>(out + e * 8), o);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = o[i] > 0.f ? d[i] : 0.f;
    *reinterpret_cast<uint4*>(din + e * 8) = pack8(d);
  }
}
extern "C" int mscl_relu_bwd(const uint16_t* dout, const uint16_t* out, uint16_t* din, int64_t n, void* stream) {
  if (!dout || !out || !din || n <= 0) return MSCL_E_ARG;
  if (n % 8) return MSCL_E_SHAPE;
  long blocks = (n / 8 + 255) / 256; if (blocks > ew_cap()) blocks = ew_cap();
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, out, din, (long)(n / 8));
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- nearest / trilinear resize (+add)
// PyTorch semantics: nearest: src = floor(dst * in/out); trilinear align_corners=False:
// s = max(0, (dst+0.5)*in/out - 0.5), i0 = floor(s), i1 = min(i0+1, in-1), w1 = s - i0.
__device__ __forceinline__ void lin_coord(int d, int in, int outn, int& i0, int& i1, float& w1) {
  const float sc = (float)in / (float)outn;
  float s = ((float)d + 0.5f) * sc - 0.5f; s = s < 0.f ? 0.f : s;
  i0 = (int)s; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 > in - 1 ? in - 1 : i0 + 1; w1 = s - (float)i0;
}
// index decode by multiply-shift (FastDiv): 64-bit / and % by runtime values cost ~100 instructions each, five per element
struct UpDiv { FastDiv G, Wd, Hd, Td, Ws, Hs, Ts; };
__global__ __launch_bounds__(256) void upsample_add_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N,
                                                           int Ts, int Hs, int Ws, int Td, int Hd, int Wd, int C,
                                                           int trilinear, int accumulate, UpDiv dv) {
  const int G = C >> 3;
  const int total = N * Td * Hd * Wd * G;            // < 2^31 (checked by the launcher)
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int r = fdiv(e, dv.G); const int gq = e - r * G;
    int q = fdiv(r, dv.Wd); const int w = r - q * Wd; r = q;
    q = fdiv(r, dv.Hd); const int h = r - q * Hd; r = q;
    const int n = fdiv(r, dv.Td), t = r - n * Td;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!trilinear) {
      const int ts = fdiv(t * Ts, dv.Td), hs = fdiv(h * Hs, dv.Hd), ws = fdiv(w * Ws, dv.Wd);
      unpack8(*reinterpret_cast<const uint4*>(src + ((((long)n * Ts + ts) * Hs + hs) * Ws + ws) * C + gq * 8), f);
    } else {
      int t0, t1, h0, h1, w0, w1; float a, b, c;
      lin_coord(t, Ts, Td, t0, t1, a); lin_coord(h, Hs, Hd, h0, h1, b); lin_coord(w, Ws, Wd, w0, w1, c);
      // All eight corner offsets first, then eight 16-byte BUFFER loads in flight (32-bit offsets from one descriptor: the launcher
      // bounds the source at 2 GiB).  History: round 5 saw one output row of this kernel lose ONE corner term for lanes 48-63 about once
      // in 4000 launches inside the three-stream step and read it as a load that returned zeros; round 6 found the corner's WEIGHT was
      // zero -- a `v_pk_mul_f32` with a cross-half op_sel (what hipcc makes of the products below) returns 0 in the low half for the
      // last lane quarter beside MFMA kernels of another queue (profiles/r06_flake.md).  The library is built without packed fp32
      // instructions since (build.sh), so the products below are plain v_mul_f32.
      const bf16_t* ap[8]; float wt8[8]; uint4 v8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tt = (k & 4) ? t1 : t0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
        wt8[k] = ((k & 4) ? a : 1.f - a) * ((k & 2) ? b : 1.f - b) * ((k & 1) ? c : 1.f - c);
        ap[k] = src + ((((long)n * Ts + tt) * Hs + hh) * Ws + ww) * C + gq * 8;
      }
      {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const uint64_t sa = reinterpret_cast<uint64_t>(src);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sa), hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((ap[k] - src) * 2), 0, 0);
          v8[k] = make_uint4(q[0], q[1], q[2], q[3]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float g8[8];
        unpack8(v8[k], g8);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += wt8[k] * g8[i];
      }
    }
    if (accumulate) {
      float d8[8]; unpack8(*reinterpret_cast<const uint4*>(dst + (long)e * 8), d8);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += d8[i];
    }
    *reinterpret_cast<uint4*>(dst + (long)e * 8) = pack8(f);
  }
}
static UpDiv make_updiv(int G, int Ts, int Hs, int Ws, int Td, int Hd, int Wd) {
  UpDiv d; d.G = make_fastdiv(G); d.Wd = make_fastdiv(Wd); d.Hd = make_fastdiv(Hd); d.Td = make_fastdiv(Td);
  d.Ws = make_fastdiv(Ws); d.Hs = make_fastdiv(Hs); d.Ts = make_fastdiv(Ts);
  return d;
}
extern "C" int mscl_upsample_add(const uint16_t* src, uint16_t* dst, int N, int Ts, int Hs, int Ws, int Td, int Hd, int Wd,
                                 int C, int trilinear, int accumulate, void* stream) {
  if (!src || !dst || N <= 0 || Ts <= 0 || Hs <= 0 || Ws <= 0 || Td <= 0 || Hd <= 0 || Wd <= 0) return MSCL_E_ARG;
  if (C % 8) return MSCL_E_SHAPE;
  const long total = (long)N * Td * Hd * Wd * (C / 8);
  if (total >= (1L << 31) || (long)Td * Ts >= (1L << 31)) return MSCL_E_SHAPE;
  if (trilinear && (long)N * Ts * Hs * Ws * C * 2 >= (1L << 31)) return MSCL_E_SHAPE;      // 32-bit byte offsets into the source
  long blocks = (total + 255) / 256; if (blocks > ew_cap()) blocks = ew_cap();
  hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, N, Ts, Hs, Ws,
                     Td, Hd, Wd, C, trilinear, accumulate, make_updiv(C / 8, Ts, Hs, Ws, Td, Hd, Wd));
  MSCL_LAUNCH_CHECK();
  return 0;
}

// backward: gather form.  For each coarse cell, sum the fine cells that read it (with their weights).
// Per axis a coarse index i receives from fine d in a window; we scan the (small) fine extent per axis.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const bf16_t* __restrict__ dd, bf16_t* __restrict__ ds, int N,
                                                           int Ts, int Hs, int Ws, int Td, int Hd, int Wd, int C,
                                                           int trilinear, UpDiv dv) {
  const int G = C >> 3;
  const int total = N * Ts * Hs * Ws * G;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int r = fdiv(e, dv.G); const int gq = e - r * G;
    int q = fdiv(r, dv.Ws); const int ws = r - q * Ws; r = q;
    q = fdiv(r, dv.Hs); const int hs = r - q * Hs; r = q;
    const int n = fdiv(r, dv.Ts), ts = r - n * Ts;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // candidate fine ranges.  nearest: fine d reads coarse floor(d * in / out), so coarse i is read by d in [i*s, (i+1)*s),
    // s = out / in.  trilinear (align_corners = False): d reads i0 = floor(src), i1 = i0 + 1 with src = (d + 0.5) / s - 0.5,
    // so i is touched when src is in (i - 1, i + 1), i.e. d in ((i - 0.5) s - 0.5, (i + 1.5) s - 0.5).  floor / ceil of the
    // OPEN bounds already add a zero-weight candidate each side when they are integers, which is also the only case float
    // rounding could move a bound; the weights below stay the exact test.
    // (The window used to be +-(s + 2) around the centre: 9^3 candidates at s = 2, now 6^3 trilinear / 3^3 nearest.)
    const float st_ = (float)Td / (float)Ts, sh_ = (float)Hd / (float)Hs, sw_ = (float)Wd / (float)Ws;
    int tlo, thi, hlo, hhi, wlo, whi;
    if (!trilinear) {
      tlo = (int)floorf(ts * st_); thi = (int)ceilf((ts + 1) * st_);
      hlo = (int)floorf(hs * sh_); hhi = (int)ceilf((hs + 1) * sh_);
      wlo = (int)floorf(ws * sw_); whi = (int)ceilf((ws + 1) * sw_);
    } else {
      tlo = (int)floorf((ts - 0.5f) * st_ - 0.5f); thi = (int)ceilf((ts + 1.5f) * st_ - 0.5f);
      hlo = (int)floorf((hs - 0.5f) * sh_ - 0.5f); hhi = (int)ceilf((hs + 1.5f) * sh_ - 0.5f);
      wlo = (int)floorf((ws - 0.5f) * sw_ - 0.5f); whi = (int)ceilf((ws + 1.5f) * sw_ - 0.5f);
    }
    for (int t = max(0, tlo); t <= min(Td - 1, thi); ++t) {
      float wt_t;
      if (!trilinear) { wt_t = (fdiv(t * Ts, dv.Td) == ts) ? 1.f : 0.f; }
      else { int i0, i1; float a; lin_coord(t, Ts, Td, i0, i1, a); wt_t = (i0 == ts ? 1.f - a : 0.f) + (i1 == ts ? a : 0.f); }
      if (wt_t == 0.f) continue;
      for (int h = max(0, hlo); h <= min(Hd - 1, hhi); ++h) {
        float wt_h;
        if (!trilinear) { wt_h = (fdiv(h * Hs, dv.Hd) == hs) ? 1.f : 0.f; }
        else { int i0, i1; float a; lin_coord(h, Hs, Hd, i0, i1, a); wt_h = (i0 == hs ? 1.f - a : 0.f) + (i1 == hs ? a : 0.f); }
        if (wt_h == 0.f) continue;
        for (int w = max(0, wlo); w <= min(Wd - 1, whi); ++w) {
          float wt_w;
          if (!trilinear) { wt_w = (fdiv(w * Ws, dv.Wd) == ws) ? 1.f : 0.f; }
          else { int i0, i1; float a; lin_coord(w, Ws, Wd, i0, i1, a); wt_w = (i0 == ws ? 1.f - a : 0.f) + (i1 == ws ? a : 0.f); }
          if (wt_w == 0.f) continue;
          float g8[8];
          unpack8(*reinterpret_cast<const uint4*>(dd + ((((long)n * Td + t) * Hd + h) * Wd + w) * C + gq * 8), g8);
          const float wt = wt_t * wt_h * wt_w;
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += wt * g8[i];
        }
      }
    }
    *reinterpret_cast<uint4*>(ds + (long)e * 8) = pack8(f);
  }
}
// The same gather for scales <= 2 per axis (every upsample of the FPN / SEPC necks): the fine cells that touch a coarse cell are
// at most FOUR consecutive ones per axis (nearest: 2; trilinear, align_corners = False: 2i-1 .. 2i+2), so each axis is decoded once
// into (first index, four weights) -- the general kernel above re-derives the weights inside the triple loop, 126 coordinate
// computations and up to 216 trips per element for 64 loads, and took 26-31 us on a 12.8-MB map.  Same candidates in the same
// (t, h, w) order with the same weight products: the same bits.
__device__ __forceinline__ void up_axis4(int is, int n_src, int n_dst, FastDiv dD, int trilinear, int& d0, float (&wt)[4]) {
  const float sc = (float)n_dst / (float)n_src;
  int lo, hi;
  if (!trilinear) { lo = (int)floorf(is * sc); hi = (int)ceilf((is + 1) * sc); }
  else { lo = (int)floorf((is - 0.5f) * sc - 0.5f); hi = (int)ceilf((is + 1.5f) * sc - 0.5f); }
  lo = max(0, lo); hi = min(n_dst - 1, hi);
  auto weight = [&](int d) -> float {
    if (d > hi) return 0.f;
    if (!trilinear) return (fdiv(d * n_src, dD) == is) ? 1.f : 0.f;
    int i0, i1; float a; lin_coord(d, n_src, n_dst, i0, i1, a);
    return (i0 == is ? 1.f - a : 0.f) + (i1 == is ? a : 0.f);
  };
  d0 = lo;
  for (int k = 0; k < 3 && weight(d0) == 0.f && d0 < hi; ++k) ++d0;      // at most two zero-weight candidates lead (open bounds)
#pragma unroll
  for (int k = 0; k < 4; ++k) wt[k] = weight(d0 + k);
}
__global__ __launch_bounds__(256) void upsample_bwd4_kernel(const bf16_t* __restrict__ dd, bf16_t* __restrict__ ds, int N,
                                                            int Ts, int Hs, int Ws, int Td, int Hd, int Wd, int C,
                                                            int trilinear, UpDiv dv) {
  const int G = C >> 3;
  const int total = N * Ts * Hs * Ws * G;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int r = fdiv(e, dv.G); const int gq = e - r * G;
    int q = fdiv(r, dv.Ws); const int ws = r - q * Ws; r = q;
    q = fdiv(r, dv.Hs); const int hs = r - q * Hs; r = q;
    const int n = fdiv(r, dv.Ts), ts = r - n * Ts;
    int t0, h0, w0; float wt_t[4], wt_h[4], wt_w[4];
    up_axis4(ts, Ts, Td, dv.Td, trilinear, t0, wt_t);
    up_axis4(hs, Hs, Hd, dv.Hd, trilinear, h0, wt_h);
    up_axis4(ws, Ws, Wd, dv.Wd, trilinear, w0, wt_w);
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wt_t[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (wt_h[b] == 0.f) continue;
        const bf16_t* row = dd + ((((long)n * Td + t0 + a) * Hd + h0 + b) * Wd + w0) * C + gq * 8;
        uint4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = wt_w[c] != 0.f ? *reinterpret_cast<const uint4*>(row + (long)c * C) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (wt_w[c] == 0.f) continue;
          float g8[8]; unpack8(v[c], g8);
          const float wt = wt_t[a] * wt_h[b] * wt_w[c];
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += wt * g8[i];
        }
      }
    }
    *reinterpret_cast<uint4*>(ds + (long)e * 8) = pack8(f);
  }
}
extern "C" int mscl_upsample_bwd(const uint16_t* ddst, uint16_t* dsrc, int N, int Ts, int Hs, int Ws, int Td, int Hd, int Wd,
                                 int C, int trilinear, void* stream) {
  if (!ddst || !dsrc || N <= 0 || Ts <= 0 || Hs <= 0 || Ws <= 0 || Td <= 0 || Hd <= 0 || Wd <= 0) return MSCL_E_ARG;
  if (C % 8) return MSCL_E_SHAPE;
  const long total = (long)N * Ts * Hs * Ws * (C / 8);
  if ((long)N * Td * Hd * Wd * (C / 8) >= (1L << 31) || (long)Td * Ts >= (1L << 31)) return MSCL_E_SHAPE;
  long blocks = (total + 255) / 256; if (blocks > ew_cap()) blocks = ew_cap();
  if (Td <= 2 * Ts && Hd <= 2 * Hs && Wd <= 2 * Ws && Td >= Ts && Hd >= Hs && Wd >= Ws)
    hipLaunchKernelGGL(upsample_bwd4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ddst, dsrc, N, Ts, Hs, Ws,
                       Td, Hd, Wd, C, trilinear, make_updiv(C / 8, Ts, Hs, Ws, Td, Hd, Wd));
  else
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ddst, dsrc, N, Ts, Hs, Ws,
                       Td, Hd, Wd, C, trilinear, make_updiv(C / 8, Ts, Hs, Ws, Td, Hd, Wd));
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- mean over the middle axis
// x (outer, inner, C) bf16 -> out (outer, C) fp32.  One 1024-thread block per (outer, channel chunk): G = chunk / 8 channel granules x
// 1024 / G row lanes, eight rows in flight per thread, wave shuffles over the row lanes, then the 16 waves through LDS in wave order
// (a fixed order: deterministic).  These launches sit on the way into the projection head of every chain and are pure latency:
// with one row in flight the 8-block launch over the layer-4 map (98 rows x 512 channels per clip) was 25 dependent round trips =
// 40 us for 0.8 MB (round 3); with 256 threads and 512-channel chunks it was still 4 trips (10 us), and the 6272-row pyramid level
// 49 trips (34 us).  Round 6: chunks of 128 channels (64 on long maps) and 1024 threads: one trip on the layer-4 map, six on the
// pyramid level, and 2-4 x the blocks.  `C` is the chunk width the thread layout sees, `ldc` the row pitch of the map in elements.
__global__ __launch_bounds__(1024) void pool_fwd_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int inner, int C, int ldc) {
  __shared__ float red[16 * 128];
  x += blockIdx.y * C; out += blockIdx.y * C;
  const int G = C >> 3;                           // 8 or 16: a divisor of 64
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 1024 / G;
  const long base = (long)blockIdx.x * inner * ldc;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int UNR = 8;
  for (int r0 = tr; r0 < inner; r0 += RP * UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = r0 + u * RP;
      v[u] = *reinterpret_cast<const uint4*>(x + base + (long)(r < inner ? r : r0) * ldc + tg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + u * RP >= inner) break;
      float f[8]; unpack8(v[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += __shfl_xor(s[i], o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave * C + tg * 8 + i] = s[i];
  }
  __syncthreads();
  const float inv = 1.f / (float)inner;
  for (int i = threadIdx.x; i < C; i += 1024) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w * C + i];
    out[(long)blockIdx.x * ldc + i] = t * inv;
  }
}
extern "C" int mscl_pool_fwd(const uint16_t* x, float* out, int outer, int inner, int C, void* stream) {
  if (!x || !out || outer <= 0 || inner <= 0 || C <= 0) return MSCL_E_ARG;
  if (C % 8 || ilog2_exact(C / 8) < 0 || C > 4096) return MSCL_E_SHAPE;
  int Cc = inner > 2048 ? 64 : 128;               // (G = 8 / 16 granules: 128 / 64 row lanes)
  if (Cc > C) Cc = C;                             // (C / 8 is a power of two: G divides 64 whatever the width)
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(outer, C / Cc), dim3(1024), 0, (hipStream_t)stream, x, out, inner, Cc, C);
  MSCL_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, bf16_t* __restrict__ dx, long inner,
                                                       int C, long total, int accumulate) {
  const int G = C >> 3;
  const float inv = 1.f / (float)inner;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int gq = (int)(e % G); const long o = (e / G) / inner;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = dout[o * C + gq * 8 + i] * inv;
    if (accumulate) { float d8[8]; unpack8(*reinterpret_cast<const uint4*>(dx + e * 8), d8);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += d8[i]; }
    *reinterpret_cast<uint4*>(dx + e * 8) = pack8(f);
  }
}
extern "C" int mscl_pool_bwd(const float* dout, uint16_t* dx, int outer, int inner, int C, int accumulate, void* stream) {
  if (!dout || !dx || outer <= 0 || inner <= 0 || C <= 0) return MSCL_E_ARG;
  if (C % 8) return MSCL_E_SHAPE;
  const long total = (long)outer * inner * (C / 8);
  long blocks = (total + 255) / 256; if (blocks > ew_cap()) blocks = ew_cap();
  hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, dx, (long)inner, C, total, accumulate);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- Linear on a few rows (fp32)
// y[r][o] = relu?( sum_i x[r][i] w[o][i] + b[o] ).  One wave per output feature, all rows at once.
#define LIN_MAX_ROWS 32
// The row count is a template parameter (8 / 16 / 32, rows beyond the real count are clamped duplicates that are never
// stored): with a run-time `if (r < rows)` inside the unrolled row loops every row became its own branch with its own
// dependent load, and these two kernels -- a few KB of work each, on the path into and out of the loss -- took 10-36 us.
template <int ROWS>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int rows,
                                                         int in_f, int out_f, int relu) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= out_f) return;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  for (int i = lane; i < in_f; i += 64) {
    const float wv = w[(long)o * in_f + i];
    float xv[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) xv[r] = x[(long)(r < rows ? r : rows - 1) * in_f + i];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = fmaf(wv, xv[r], acc[r]);
  }
  const float bv = b ? b[o] : 0.f;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float v = wave_sum(acc[r]) + bv;
    if (relu) v = fmaxf(v, 0.f);
    if (lane == 0 && r < rows) y[(long)r * out_f + o] = v;
  }
}
extern "C" int mscl_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in_f, int out_f,
                               int relu, void* stream) {
  if (!x || !w || !y || rows <= 0 || in_f <= 0 || out_f <= 0) return MSCL_E_ARG;
  if (rows > LIN_MAX_ROWS) return MSCL_E_SHAPE;
  const dim3 grid((out_f + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 8) hipLaunchKernelGGL(linear_fwd_kernel<8>, grid, dim3(256), 0, st, x, w, b, y, rows, in_f, out_f, relu);
  else if (rows <= 16) hipLaunchKernelGGL(linear_fwd_kernel<16>, grid, dim3(256), 0, st, x, w, b, y, rows, in_f, out_f, relu);
  else hipLaunchKernelGGL(linear_fwd_kernel<32>, grid, dim3(256), 0, st, x, w, b, y, rows, in_f, out_f, relu);
  MSCL_LAUNCH_CHECK();
  return 0;
}
// backward: g = dy * (y>0 if relu);  dx[r][i] = sum_o g[r][o] w[o][i];  dw[o][i] += sum_r g[r][o] x[r][i];  db[o] += sum_r g[r][o]
// dx[r][i] += sum over this block's 32 output features; grid = (in_f/256, out_f/32); dx is pre-zeroed
#define LIN_OCHUNK 32
template <int ROWS>
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                            int in_f, int out_f, int relu) {
  __shared__ float g[LIN_OCHUNK * ROWS];             // [oo][r]: a thread reads its ROWS values of one oo as float4s
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ic = i < in_f ? i : in_f - 1;
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  // gridDim.y blocks share the output features chunk by chunk and add their parts atomically; with gridDim.y == 1
  // (deterministic mode) one block walks every chunk in order and stores plainly
  for (int o0 = blockIdx.y * LIN_OCHUNK; o0 < out_f; o0 += gridDim.y * LIN_OCHUNK) {
    __syncthreads();
    for (int e = threadIdx.x; e < ROWS * LIN_OCHUNK; e += 256) {
      const int r = e / LIN_OCHUNK, oo = e % LIN_OCHUNK, o = o0 + oo;
      float v = 0.f;
      if (r < rows && o < out_f) { v = dy[(long)r * out_f + o]; if (relu && !(y[(long)r * out_f + o] > 0.f)) v = 0.f; }
      g[oo * ROWS + r] = v;                           // rows beyond the real count and features beyond out_f hold zeros
    }
    __syncthreads();
    float wv[LIN_OCHUNK];
#pragma unroll
    for (int oo = 0; oo < LIN_OCHUNK; ++oo) wv[oo] = w[(long)(o0 + oo < out_f ? o0 + oo : out_f - 1) * in_f + ic];
#pragma unroll
    for (int oo = 0; oo < LIN_OCHUNK; ++oo) {
#pragma unroll
      for (int r4 = 0; r4 < ROWS; r4 += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(&g[oo * ROWS + r4]);
        acc[r4] = fmaf(gv.x, wv[oo], acc[r4]); acc[r4 + 1] = fmaf(gv.y, wv[oo], acc[r4 + 1]);
        acc[r4 + 2] = fmaf(gv.z, wv[oo], acc[r4 + 2]); acc[r4 + 3] = fmaf(gv.w, wv[oo], acc[r4 + 3]);
      }
    }
  }
  if (i >= in_f) return;
  if (gridDim.y == 1) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) if (r < rows) dx[(long)r * in_f + i] = acc[r];
  } else {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) if (r < rows) atomicAdd(&dx[(long)r * in_f + i], acc[r]);
  }
}
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ dw,
                                                            float* __restrict__ db, int rows, int in_f, int out_f, int relu) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)out_f * in_f) return;
  const int o = (int)(e / in_f), i = (int)(e % in_f);
  float s = 0.f, sb = 0.f;
  for (int r = 0; r < rows; ++r) {
    float g = dy[(long)r * out_f + o];
    if (relu && !(y[(long)r * out_f + o] > 0.f)) g = 0.f;
    s += g * x[(long)r * in_f + i]; sb += g;
  }
  dw[e] += s;
  if (i == 0 && db) db[o] += sb;
}
// hipMemsetAsync is NOT used for the pre-zeroing: on ROCm 7.0 a memset node inside any HIP graph other than the first one
// a process instantiates writes garbage from its second replay on (tools/_ms.py reproduces it with this very entry
// point); a fill kernel has no such problem and costs the same launch.
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
extern "C" int mscl_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                               float* db, int rows, int in_f, int out_f, int relu, void* stream) {
  if (!x || !w || !y || !dy || !dw || rows <= 0 || in_f <= 0 || out_f <= 0) return MSCL_E_ARG;
  if (rows > LIN_MAX_ROWS) return MSCL_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    const long nz = (long)rows * in_f;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, dx, nz);
    MSCL_LAUNCH_CHECK();
    const dim3 grid((in_f + 255) / 256, mscl_det() ? 1 : (out_f + LIN_OCHUNK - 1) / LIN_OCHUNK);
    if (rows <= 8) hipLaunchKernelGGL(linear_bwd_dx_kernel<8>, grid, dim3(256), 0, st, w, y, dy, dx, rows, in_f, out_f, relu);
    else if (rows <= 16) hipLaunchKernelGGL(linear_bwd_dx_kernel<16>, grid, dim3(256), 0, st, w, y, dy, dx, rows, in_f, out_f, relu);
    else hipLaunchKernelGGL(linear_bwd_dx_kernel<32>, grid, dim3(256), 0, st, w, y, dy, dx, rows, in_f, out_f, relu);
    MSCL_LAUNCH_CHECK();
  }
  const long tot = (long)out_f * in_f;
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, y, dy, dw, db, rows, in_f, out_f, relu);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- F.normalize(dim=1, eps=1e-12)
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ norms, int dim) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) { const float v = x[(long)r * dim + i]; s += v * v; }
  s = wave_sum(s);
  const float nrm = fmaxf(sqrtf(s), 1e-12f);
  for (int i = lane; i < dim; i += 64) y[(long)r * dim + i] = x[(long)r * dim + i] / nrm;
  if (lane == 0 && norms) norms[r] = nrm;
}
extern "C" int mscl_l2norm_fwd(const float* x, float* y, float* norms, int rows, int dim, void* stream) {
  if (!x || !y || rows <= 0 || dim <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, y, norms, dim);
  MSCL_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norms,
                                                        const float* __restrict__ dy, float* __restrict__ dx, int dim) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += y[(long)r * dim + i] * dy[(long)r * dim + i];
  s = wave_sum(s);
  const float inv = 1.f / norms[r];
  for (int i = lane; i < dim; i += 64) dx[(long)r * dim + i] = (dy[(long)r * dim + i] - y[(long)r * dim + i] * s) * inv;
}
extern "C" int mscl_l2norm_bwd(const float* y, const float* norms, const float* dy, float* dx, int rows, int dim, void* stream) {
  if (!y || !norms || !dy || !dx || rows <= 0 || dim <= 0) return MSCL_E_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, y, norms, dy, dx, dim);
  MSCL_LAUNCH_CHECK();
  return 0;
}
