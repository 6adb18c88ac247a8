// Max-pool (1,3,3) / stride (1,2,2) / pad (0,1,1) on NDHWC bf16 maps: the pool1 of the Bottleneck trunks
// (reference: backbones/resnet3d.py:461-467 for ResNet3dSlowOnly, backbones/fastonly.py:222-235 for r2d_50's BottleneckStem).
// HBM-bound: one 16-byte (8-channel) granule per lane, lanes run along the channel axis.  The forward also writes, per output
// granule, which of the 9 window taps won each channel (4 bits per channel, one uint32 per granule); the backward is a GATHER:
// every input granule looks at the <= 4 windows that contain it and adds the gradients of the channels it won -- no atomics,
// bit-reproducible.  Ties go to the first tap in (h, w) scan order, as torch's max_pool3d does (its comparison is `val > max`),
// which matters behind a ReLU where whole windows are zero.
#include "common.h"

__global__ __launch_bounds__(256) void maxpool_hw_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                             uint32_t* __restrict__ win, int H, int W, int Ho, int Wo, int G,
                                                             long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int g = (int)(e % G); long p = e / G;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho); const long nt = p / Ho;
    float best[8]; uint32_t arg = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = -INFINITY;
    uint4 v[9]; bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int h = ho * 2 - 1 + k / 3, w = wo * 2 - 1 + k % 3;
      ok[k] = h >= 0 && h < H && w >= 0 && w < W;
      if (ok[k]) v[k] = *reinterpret_cast<const uint4*>(x + (((nt * H + h) * W + w) * (long)G + g) * 8);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (!ok[k]) continue;
      float f[8]; unpack8(v[k], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (f[i] > best[i] || f[i] != f[i]) { best[i] = f[i]; arg = (arg & ~(0xFu << (4 * i))) | ((uint32_t)k << (4 * i)); }
      }
    }
    *reinterpret_cast<uint4*>(out + e * 8) = pack8(best);
    win[e] = arg;
  }
}

__global__ __launch_bounds__(256) void maxpool_hw_bwd_kernel(const bf16_t* __restrict__ dout, const uint32_t* __restrict__ win,
                                                             bf16_t* __restrict__ dx, int H, int W, int Ho, int Wo, int G,
                                                             long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int g = (int)(e % G); long p = e / G;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H); const long nt = p / H;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows ho with ho*2-1 <= h <= ho*2+1  <=>  ho in {ceil((h-1)/2) .. floor((h+1)/2)}
    const int ho0 = h >> 1, ho1 = (h + 1) >> 1, wo0 = w >> 1, wo1 = (w + 1) >> 1;
    for (int ho = ho0; ho <= ho1; ++ho) {
      if (ho >= Ho) continue;
      const int kh = h - (ho * 2 - 1);
      for (int wo = wo0; wo <= wo1; ++wo) {
        if (wo >= Wo) continue;
        const uint32_t k = (uint32_t)(kh * 3 + (w - (wo * 2 - 1)));
        const long o = ((nt * Ho + ho) * Wo + wo) * (long)G + g;
        const uint32_t a = win[o];
        float d[8]; unpack8(*reinterpret_cast<const uint4*>(dout + o * 8), d);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += ((a >> (4 * i)) & 0xFu) == k ? d[i] : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(dx + e * 8) = pack8(acc);
  }
}

extern "C" int mscl_maxpool_hw_fwd(const uint16_t* x, uint16_t* out, uint32_t* win, int NT, int H, int W, int C, void* stream) {
  if (!x || !out || !win || NT <= 0 || H <= 0 || W <= 0 || C <= 0) return MSCL_E_ARG;
  if (C % 8) return MSCL_E_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C / 8;
  const long total = (long)NT * Ho * Wo * G;
  long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_hw_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, win, H, W, Ho, Wo, G, total);
  MSCL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mscl_maxpool_hw_bwd(const uint16_t* dout, const uint32_t* win, uint16_t* dx, int NT, int H, int W, int C, void* stream) {
  if (!dout || !win || !dx || NT <= 0 || H <= 0 || W <= 0 || C <= 0) return MSCL_E_ARG;
  if (C % 8) return MSCL_E_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, G = C / 8;
  const long total = (long)NT * H * W * G;
  long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_hw_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, win, dx, H, W, Ho, Wo, G, total);
  MSCL_LAUNCH_CHECK();
  return 0;
}
