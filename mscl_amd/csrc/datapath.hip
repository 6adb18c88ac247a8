// Input-side data path on the GPU: paired crop + resize of decoded clips.
// Replaces, per view (query / key), MoCoRandomResizedCrop's crop (moco_augmentations.py:110-163: the box is drawn on the host,
// `img[y1:y2, x1:x2]`), MoCoResize (moco_augmentations.py:236-321: mmcv.imresize -> cv2.resize(..., INTER_LINEAR)) and
// MoCoNormalize (moco_augmentations.py:324-354: / 255, HWC frames -> CTHW fp32) in ONE pass over the raw frames: the crop is
// never materialised, the uint8 frames cross PCIe once (a quarter of the fp32 bytes) and the host does no per-pixel work.
// Roofline: HBM (reads the cropped region once through L2, writes 4 B per output element); a 32-sample batch of 8 x 112^2
// views is ~10 MB out, microseconds.
//
// Arithmetic of cv2.resize INTER_LINEAR (OpenCV imgproc/resize.cpp; OpenCV is NOT vendored in the reference and is absent
// here: restated from its published algorithm, parity with the library itself is unpinned):
//   source coordinate  f = (d + 0.5) * (src / dst) - 0.5 in double, cast to float; s = floor(f); f -= s;
//   x axis: s < 0 -> (s, f) = (0, 0); s >= src_w - 1 -> (src_w - 1, 0) [no right neighbour]; y axis: rows s, s + 1 clamped
//   uint8: coefficients round(c * 2048) as int16; horizontal sums in int32; vertical
//          dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
//          an exact 2 x 2 reduction takes the INTER_AREA path: (a + b + c + d + 2) >> 2
//   float: the same taps and float coefficients, horizontal then vertical, no rounding.
#include "common.h"

struct Tap { int s0, s1; float f; };

__device__ __forceinline__ Tap axis_tap(int d, int src, int dst, bool clamp_no_neighbour) {
  const double scale = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  Tap t;
  if (clamp_no_neighbour) {                 // x axis (HResize): taps and coefficient adjusted together
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    t.s0 = s; t.s1 = min(s + 1, src - 1); t.f = f;
  } else {                                  // y axis: rows clamped, coefficient kept
    t.s0 = min(max(s, 0), src - 1); t.s1 = min(max(s + 1, 0), src - 1); t.f = f;
  }
  return t;
}

// one thread per output pixel, all channels; box = {x1, y1, x2, y2} in source pixels (x2, y2 exclusive)
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ boxes,
                                                             float* __restrict__ out, int B, int T, int Hs, int Ws, int Ho, int Wo,
                                                             long bstride) {
  const long total = (long)B * T * Ho * Wo;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % Wo); long r = e / Wo;
    const int y = (int)(r % Ho); r /= Ho;
    const int t = (int)(r % T); const int b = (int)(r / T);
    const int x1 = boxes[4 * b], y1 = boxes[4 * b + 1], cw = boxes[4 * b + 2] - x1, ch = boxes[4 * b + 3] - y1;
    const uint8_t* frame = src + (long)b * bstride + ((long)t * Hs) * (long)Ws * 3;
    int v[3];
    if (cw == 2 * Wo && ch == 2 * Ho) {       // exact halving: cv2 switches INTER_LINEAR to the 2 x 2 area average
      const uint8_t* p0 = frame + ((long)(y1 + 2 * y) * Ws + x1 + 2 * x) * 3;
      const uint8_t* p1 = p0 + (long)Ws * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
    } else {
      const Tap tx = axis_tap(x, cw, Wo, true), ty = axis_tap(y, ch, Ho, false);
      const int a1 = __float2int_rn(tx.f * 2048.f), a0 = __float2int_rn((1.f - tx.f) * 2048.f);
      const int b1 = __float2int_rn(ty.f * 2048.f), b0 = __float2int_rn((1.f - ty.f) * 2048.f);
      const uint8_t* r0 = frame + ((long)(y1 + ty.s0) * Ws + x1) * 3;
      const uint8_t* r1 = frame + ((long)(y1 + ty.s1) * Ws + x1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = r0[tx.s0 * 3 + c] * a0 + r0[tx.s1 * 3 + c] * a1;
        const int S1 = r1[tx.s0 * 3 + c] * a0 + r1[tx.s1 * 3 + c] * a1;
        int d = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v[c] = min(max(d, 0), 255);
      }
    }
    const long plane = (long)T * Ho * Wo;
    float* o = out + (long)b * 3 * plane + ((long)t * Ho + y) * Wo + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = (float)v[c] / 255.0f;      // MoCoNormalize: float32 division
  }
}

__global__ __launch_bounds__(256) void crop_resize_f32_kernel(const float* __restrict__ src, const int* __restrict__ boxes,
                                                              float* __restrict__ out, int B, int T, int Hs, int Ws, int C, int Ho,
                                                              int Wo, long bstride) {
#pragma clang fp contract(off)       // cv2 computes taps with separate multiplies and adds: no fused multiply-add here
  const long total = (long)B * T * Ho * Wo;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % Wo); long r = e / Wo;
    const int y = (int)(r % Ho); r /= Ho;
    const int t = (int)(r % T); const int b = (int)(r / T);
    const int x1 = boxes[4 * b], y1 = boxes[4 * b + 1], cw = boxes[4 * b + 2] - x1, ch = boxes[4 * b + 3] - y1;
    const float* frame = src + (long)b * bstride + ((long)t * Hs) * (long)Ws * C;
    const Tap tx = axis_tap(x, cw, Wo, true), ty = axis_tap(y, ch, Ho, false);
    const float a1 = tx.f, a0 = 1.f - tx.f, b1 = ty.f, b0 = 1.f - ty.f;
    const float* r0 = frame + ((long)(y1 + ty.s0) * Ws + x1) * C;
    const float* r1 = frame + ((long)(y1 + ty.s1) * Ws + x1) * C;
    const long plane = (long)T * Ho * Wo;
    float* o = out + (long)b * C * plane + ((long)t * Ho + y) * Wo + x;
    for (int c = 0; c < C; ++c) {
      // separate multiplies and adds, horizontal first: the order of cv2's HResizeLinear / VResizeLinear (no fused multiply-add)
      const float S0 = __fadd_rn(__fmul_rn(r0[tx.s0 * C + c], a0), __fmul_rn(r0[tx.s1 * C + c], a1));
      const float S1 = __fadd_rn(__fmul_rn(r1[tx.s0 * C + c], a0), __fmul_rn(r1[tx.s1 * C + c], a1));
      o[c * plane] = __fadd_rn(__fmul_rn(S0, b0), __fmul_rn(S1, b1));
    }
  }
}

static int check_boxes_args(const void* src, const int* boxes, const float* out, int B, int T, int Hs, int Ws, int Ho, int Wo) {
  if (!src || !boxes || !out || B <= 0 || T <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return MSCL_E_ARG;
  return 0;
}

extern "C" int mscl_crop_resize_u8(const uint8_t* src, const int32_t* boxes, float* out, int B, int T, int Hs, int Ws, int Ho, int Wo,
                                   int64_t src_batch_stride, void* stream) {
  int e = check_boxes_args(src, boxes, out, B, T, Hs, Ws, Ho, Wo); if (e) return e;
  const long bstride = src_batch_stride > 0 ? (long)src_batch_stride : (long)T * Hs * Ws * 3;
  if (bstride < (long)T * Hs * Ws * 3) return MSCL_E_ARG;
  const long total = (long)B * T * Ho * Wo;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(crop_resize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, boxes, out, B, T, Hs, Ws, Ho, Wo, bstride);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_crop_resize_f32(const float* src, const int32_t* boxes, float* out, int B, int T, int Hs, int Ws, int C, int Ho,
                                    int Wo, int64_t src_batch_stride, void* stream) {
  int e = check_boxes_args(src, boxes, out, B, T, Hs, Ws, Ho, Wo); if (e) return e;
  if (C <= 0 || C > 16) return MSCL_E_SHAPE;
  const long bstride = src_batch_stride > 0 ? (long)src_batch_stride : (long)T * Hs * Ws * C;
  if (bstride < (long)T * Hs * Ws * C) return MSCL_E_ARG;
  const long total = (long)B * T * Ho * Wo;
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(crop_resize_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, boxes, out, B, T, Hs, Ws, C,
                     Ho, Wo, bstride);
  MSCL_LAUNCH_CHECK();
  return 0;
}
