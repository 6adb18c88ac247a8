// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o /tmp/store_bw): read [M][64] bf16, write [M][256] bf16 -- the
// bytes of a 64 -> 256 1x1x1 conv on the 56^2 maps -- with 1-KB contiguous wave stores and with the MFMA fragment's store shape.  The
// yardstick csrc/conv_k1.hip is measured against (profiles/r05_ab_sweeps.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// mode 0: a wave writes 1 KB contiguous (two full 512-B rows), 16 B per lane
// mode 1: a wave writes 16 rows x 64 B (the MFMA-fragment shape: lane row fr = row, fq = 16-B piece), 4 instr for 256 B of a 128-col half
__global__ __launch_bounds__(256) void k_full(const uint4* __restrict__ in, uint4* __restrict__ out, long rows) {
  // thread -> (row, 16-B piece p of 32): reads in[row][p % 8]
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * 32; i += (long)gridDim.x * 256) {
    const long row = i >> 5; const int p = (int)(i & 31);
    uint4 v = in[row * 8 + (p & 7)];
    v.x += p;
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void k_frag(const uint4* __restrict__ in, uint4* __restrict__ out, long rows) {
  // block = 128 rows x 128 channels (half a row: 256 B = 16 pieces); wave w: rows w*32.. ; per instr: 16 rows x 4 pieces (64 B per row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const long ntile = rows / 128 * 2;
  for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
    const long m0 = (t >> 1) * 128; const int n0 = (int)(t & 1) * 16;      // piece offset
    uint4 a[2];
    for (int i = 0; i < 2; ++i) a[i] = in[(m0 + wave * 32 + i * 16 + fr) * 8 + fq * 2];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j) {
        uint4 v = a[i]; v.x += j;
        out[(m0 + wave * 32 + i * 16 + fr) * 32 + n0 + j * 4 + fq] = v;
      }
  }
}
int main() {
  const long rows = 401408;
  uint4 *in, *out;
  CK(hipMalloc(&in, rows * 128)); CK(hipMalloc(&out, rows * 512));
  CK(hipMemset(in, 1, rows * 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {2048, 8192, 32768}) {
      for (int it = 0; it < 3; ++it) { if (mode == 0) k_full<<<grid, 256>>>(in, out, rows); else k_frag<<<grid, 256>>>(in, out, rows); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) { if (mode == 0) k_full<<<grid, 256>>>(in, out, rows); else k_frag<<<grid, 256>>>(in, out, rows); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / 20;
      printf("mode %d grid %6d: %7.1f us  %6.0f GB/s (in + out)\n", mode, grid, us, rows * 640.0 / us / 1e3);
    }
  return 0;
}
