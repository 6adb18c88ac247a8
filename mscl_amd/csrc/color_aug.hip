// Per-clip colour augmentation GIVEN its sampled parameters: the deterministic arithmetic of the kornia ops the reference
// chains in common/ssl_aug_v2.py:31-43 (ColorJitter(0.4, 0.4, 0.4, 0.1) -> RandomGrayscale -> GaussianBlur(radius 11 at
// 112 px, common/ssl_aug.py:163-171)).  kornia is not vendored by the reference: the per-op arithmetic below restates
// kornia's published enhance/colour functions (additive brightness, multiplicative contrast, saturation and hue through
// HSV with h in [0, 2 pi), ITU-R 601 luma, normalised Gaussian taps with reflect border); the parameter sampling lives on
// the host (mscl_amd/augment.py).  Everything is HBM-bound elementwise work on fp32 NCTHW clips (19 MB per view).
#include "common.h"

#define AUG_PARAM_STRIDE 16
// params[b][0] jitter on/off, [1..4] order of the four jitter ops (0 brightness, 1 contrast, 2 saturation, 3 hue),
// [5] brightness factor (applied as x + (f - 1)), [6] contrast factor, [7] saturation factor, [8] hue shift in radians,
// [9] grayscale on/off, [10] blur sigma (0 = no blur)

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  float d = mx - mn;
  v = mx;
  s = d / (mx + 1e-6f);
  if (d == 0.f) d = 1.f;
  const float rc = mx - r, gc = mx - g, bc = mx - b;
  float hh;
  if (r == mx) hh = bc - gc;                       // first maximum wins, as a first-index argmax does
  else if (g == mx) hh = 2.f * d + rc - bc;
  else hh = 4.f * d + gc - rc;
  hh = hh / d / 6.f;
  hh = hh - floorf(hh);                            // python-style % 1
  h = 6.283185307179586f * hh;
}

__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float h6 = h / 6.283185307179586f * 6.f;
  float hi = floorf(h6);
  hi = hi - 6.f * floorf(hi / 6.f);                // floor mod 6 (negative hue wraps)
  const float hm = h6 - 6.f * floorf(h6 / 6.f);
  const float f = hm - hi;
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  switch ((int)hi) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__global__ __launch_bounds__(256) void color_aug_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        const float* __restrict__ params, int plane) {
  const int b = blockIdx.y;
  const float* P = params + (size_t)b * AUG_PARAM_STRIDE;
  const bool jitter = P[0] != 0.f, gray = P[9] != 0.f;
  const int o0 = (int)P[1], o1 = (int)P[2], o2 = (int)P[3], o3 = (int)P[4];
  const float fb = P[5] - 1.f, fc = P[6], fs = P[7], fh = P[8];
  const float* xb = x + (size_t)b * 3 * plane;
  float* ob = out + (size_t)b * 3 * plane;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) {
    float r = xb[i], g = xb[plane + i], bl = xb[2 * plane + i];
    if (jitter) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int op = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
        if (op == 0) {
          r = clamp01(r + fb); g = clamp01(g + fb); bl = clamp01(bl + fb);
        } else if (op == 1) {
          r = clamp01(r * fc); g = clamp01(g * fc); bl = clamp01(bl * fc);
        } else if (op == 2) {
          float h, s, v;
          rgb_to_hsv(r, g, bl, h, s, v);
          s = clamp01(s * fs);
          hsv_to_rgb(h, s, v, r, g, bl);
        } else {
          float h, s, v;
          rgb_to_hsv(r, g, bl, h, s, v);
          h = fmodf(h + fh, 6.283185307179586f);   // sign follows the dividend; hsv_to_rgb wraps it
          hsv_to_rgb(h, s, v, r, g, bl);
        }
      }
    }
    if (gray) {
      const float y = 0.299f * r + 0.587f * g + 0.114f * bl;
      r = g = bl = y;
    }
    ob[i] = r; ob[plane + i] = g; ob[2 * plane + i] = bl;
  }
}

extern "C" int mscl_color_aug(const float* x, float* out, const float* params, int B, int T, int H, int W, void* stream) {
  if (B < 0 || T <= 0 || H <= 0 || W <= 0) return -1;
  if (B == 0) return 0;
  if (!x || !out || !params) return -1;
  const int64_t plane = (int64_t)T * H * W;
  if (plane * 3 >= (1ll << 31) || B > 65535) return -2;
  const int gx = (int)std::min<int64_t>(cdiv64(plane, 256), 4096);
  color_aug_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(x, out, params, (int)plane);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// One 1-D pass of the separable Gaussian: DIR 0 along W, 1 along H; frames are (H, W) images, `frames` per sample.
#define BLUR_MAX_K 33
template <int DIR>
__global__ __launch_bounds__(256) void gauss_blur_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const float* __restrict__ params, int ksize, int frames, int H, int W) {
  __shared__ float taps[BLUR_MAX_K];
  const int b = blockIdx.y;
  const float sigma = params[(size_t)b * AUG_PARAM_STRIDE + 10];
  const int r = ksize >> 1;
  if (sigma > 0.f) {
    if (threadIdx.x < 64) {
      const int k = threadIdx.x;
      const float d = (float)(k - r);
      const float e = k < ksize ? expf(-d * d / (2.f * sigma * sigma)) : 0.f;
      const float tot = wave_sum(e);
      if (k < ksize) taps[k] = e / tot;
    }
    __syncthreads();
  }
  const int n = frames * H * W;
  const float* xb = x + (size_t)b * n;
  float* ob = out + (size_t)b * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (sigma <= 0.f) { ob[i] = xb[i]; continue; }
    const int w = i % W, h = (i / W) % H;
    const int pos = DIR == 0 ? w : h, L = DIR == 0 ? W : H, stride = DIR == 0 ? 1 : W;
    const int base = i - pos * stride;
    float acc = 0.f;
    for (int k = 0; k < ksize; ++k) {
      int p = pos + k - r;
      p = p < 0 ? -p : p;                          // reflect without repeating the edge sample
      p = p >= L ? 2 * (L - 1) - p : p;
      acc += taps[k] * xb[base + p * stride];
    }
    ob[i] = acc;
  }
}

extern "C" int mscl_gauss_blur(const float* x, float* tmp, float* out, const float* params, int ksize, int B, int frames,
                               int H, int W, void* stream) {
  if (B < 0 || frames <= 0 || H <= 0 || W <= 0) return -1;
  if (ksize < 1 || !(ksize & 1) || ksize > BLUR_MAX_K || H <= ksize / 2 || W <= ksize / 2) return -2;
  if (B == 0) return 0;
  if (!x || !tmp || !out || !params) return -1;
  if (tmp == x || tmp == out) return -2;
  const int64_t n = (int64_t)frames * H * W;
  if (n >= (1ll << 31) || B > 65535) return -2;
  const int gx = (int)std::min<int64_t>(cdiv64(n, 256), 8192);
  gauss_blur_kernel<0><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(x, tmp, params, ksize, frames, H, W);
  MSCL_LAUNCH_CHECK();
  gauss_blur_kernel<1><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(tmp, out, params, ksize, frames, H, W);
  MSCL_LAUNCH_CHECK();
  return 0;
}
