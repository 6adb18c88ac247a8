// Stand-alone reproducer for the dropped-corner event of the trilinear up-sampling kernel (profiles/r05_flake_det.md, r06_flake.md).
// No PyTorch: the HIP runtime, the C ABI of libmscl_hip.so (include/mscl_hip.h) through dlopen, and optionally the self-checking
// twin of the kernel (libups_diag.so).  Test infrastructure; nothing here is linked into the product.
//
// What it launches -- the shapes of the SMALL deterministic step (B = 2, T = 8, 32^2) in which the event was seen:
//   stream A (the RGB query neck's level-0 sum, necks/sepc.py:125-129): R x [ P0 conv 3x3x3 128 -> 128 on the (2,2,4,4) level-1 map
//     (split-K over 9 blocks + splitk_finalize) -> trilinear up-sampling to (2,4,8,8) -> compare with the first launch's output ];
//   stream B: the RGB key trunk's forward convs (r3d_18 at (2,8,32,32)), an element-wise launch after each;
//   stream C: the flow trunk's forward convs (r2d_18 at (4,8,32,32) -> stride (2,2,2) stem), an element-wise launch after each.
// One replay of a captured three-branch HIP graph (default), or three host threads launching eagerly (--eager).
//
// usage: flake_repro [--lib PATH] [--diag PATH] [--replays N] [--reps R] [--side convs|elem|both|none] [--eager] [--one-stream]
//        prints one line per differing comparison (replay, rep, row, lanes, which corner explains it) and a summary line.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Desc { int N, T, H, W, C, To, Ho, Wo, K, kT, kH, kW, sT, sH, sW, pT, pH, pW; };      // = mscl_conv_desc
typedef int (*conv_fwd_t)(const Desc*, const uint16_t*, const uint16_t*, uint16_t*, const float*, const uint16_t*, int, float*, float*,
                          float*, int64_t, void*);
typedef int (*upsample_t)(const uint16_t*, uint16_t*, int, int, int, int, int, int, int, int, int, int, void*);
typedef int (*add_relu_t)(const uint16_t*, const uint16_t*, const uint16_t*, uint16_t*, int64_t, int, void*);
typedef int (*set_det_t)(int);

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t g_seed = 12345;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

static uint16_t* dev_bf16(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (auto& v : h) v = f2bf(rnd() * scale);
  uint16_t* d; CHECK(hipMalloc(&d, n * 2)); CHECK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

__device__ unsigned g_cmp[4];                 // [0] comparisons made, [1] differing comparisons, [2] differing 16-byte granules
__device__ unsigned g_bad[256 * 4];           // {comparison index, granule index, got.x, want.x} of the first 256 differing granules
__global__ void cmp_kernel(const uint4* got, const uint4* want, int n16) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  const unsigned it = g_cmp[0];
  for (int i = threadIdx.x; i < n16; i += blockDim.x) {
    const uint4 a = got[i], b = want[i];
    if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) {
      any = 1;
      const unsigned s = atomicAdd(&g_cmp[2], 1u);
      if (s < 256) { g_bad[4 * s] = it; g_bad[4 * s + 1] = (unsigned)i; g_bad[4 * s + 2] = a.x; g_bad[4 * s + 3] = b.x; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { if (any) atomicAdd(&g_cmp[1], 1u); g_cmp[0] = it + 1; }
}

// --probe M: stream A runs THIS kernel instead of the conv -> up-sampling pair.  No memory traffic inside the loop: every lane forms the
// eight trilinear weights from (a, b, c) the way the compiled up-sampling kernel does -- four "producer" products p = {(1-b)(1-a),
// b(1-a), (1-b)a, ba}, each in BOTH halves of a register pair, then four "consumer" products w = (1-c, c) x p -- in the instruction
// form M selects (inline asm, so the form is fixed), and compares every half of every result with plain v_mul_f32 arithmetic.
//   M = 0  producers v_pk_mul_f32 with op_sel (what hipcc emits), consumers v_pk_mul_f32           (the compiled kernel's form)
//   M = 1  producers plain v_mul_f32,                              consumers v_pk_mul_f32
//   M = 2  producers v_pk_mul_f32 with op_sel,                     consumers plain v_mul_f32
//   M = 3  as 0 with v_pk_fma_f32 (+ 0) in place of v_pk_mul_f32
//   M = 4  all plain v_mul_f32 (control)
//   M = 5  as 0 with an `s_nop 7` between any two packed instructions
//   M = 6  producers plain v_mul_f32, consumers v_pk_add_f32 (w = (1-c, c) + p)
// g_probe[0] += iterations, [1] += mismatching halves, [2 + q] += those in lane quarter q, [6 + i] += those in half i of
// {p00.lo, p00.hi, p10.lo, p10.hi, p01.lo, p01.hi, p11.lo, p11.hi, w0 .. w7}.
__device__ unsigned g_probe[32];
template <int M>
__global__ __launch_bounds__(256) void valu_probe_kernel(float a0, float b0, float c0, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned bad = 0, nbad = 0;
  float a = a0, b = b0, c = c0;
  for (int it = 0; it < iters; ++it) {
    float2 ba = make_float2(b, a), nba = make_float2(1.f - b, 1.f - a), cc = make_float2(1.f - c, c);
    asm volatile("" : "+v"(ba), "+v"(nba), "+v"(cc));
    float2 p00, p10, p01, p11, w01, w23, w45, w67;
    if constexpr (M == 0 || M == 2) {
      asm volatile("v_pk_mul_f32 %0, %4, %4 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                   "v_pk_mul_f32 %1, %5, %4 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                   "v_pk_mul_f32 %2, %4, %5 op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                   "v_pk_mul_f32 %3, %5, %5 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 1"
                   : "=&v"(p00), "=&v"(p10), "=&v"(p01), "=&v"(p11) : "v"(nba), "v"(ba));
    } else if constexpr (M == 5) {
      asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %4, %4 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 7\n\t"
                   "v_pk_mul_f32 %1, %5, %4 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 7\n\t"
                   "v_pk_mul_f32 %2, %4, %5 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 7\n\t"
                   "v_pk_mul_f32 %3, %5, %5 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 7"
                   : "=&v"(p00), "=&v"(p10), "=&v"(p01), "=&v"(p11) : "v"(nba), "v"(ba));
    } else if constexpr (M == 3) {
      asm volatile("v_pk_fma_f32 %0, %4, %4, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
                   "v_pk_fma_f32 %1, %5, %4, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
                   "v_pk_fma_f32 %2, %4, %5, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
                   "v_pk_fma_f32 %3, %5, %5, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\ts_nop 1"
                   : "=&v"(p00), "=&v"(p10), "=&v"(p01), "=&v"(p11) : "v"(nba), "v"(ba));
    } else {
      asm volatile("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %6, %5\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %6, %7\n\ts_nop 1"
                   : "=&v"(p00.x), "=&v"(p10.x), "=&v"(p01.x), "=&v"(p11.x) : "v"(nba.x), "v"(nba.y), "v"(ba.x), "v"(ba.y));
      p00.y = p00.x; p10.y = p10.x; p01.y = p01.x; p11.y = p11.x;
      asm volatile("" : "+v"(p00), "+v"(p10), "+v"(p01), "+v"(p11));
    }
    if constexpr (M == 0 || M == 1) {
      asm volatile("v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %4, %7\n\tv_pk_mul_f32 %3, %4, %8\n\ts_nop 1"
                   : "=&v"(w01), "=&v"(w23), "=&v"(w45), "=&v"(w67) : "v"(cc), "v"(p00), "v"(p10), "v"(p01), "v"(p11));
    } else if constexpr (M == 5) {
      asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %4, %5\n\ts_nop 7\n\tv_pk_mul_f32 %1, %4, %6\n\ts_nop 7\n\tv_pk_mul_f32 %2, %4, %7\n\ts_nop 7\n\tv_pk_mul_f32 %3, %4, %8\n\ts_nop 7"
                   : "=&v"(w01), "=&v"(w23), "=&v"(w45), "=&v"(w67) : "v"(cc), "v"(p00), "v"(p10), "v"(p01), "v"(p11));
    } else if constexpr (M == 6) {
      asm volatile("v_pk_add_f32 %0, %4, %5\n\tv_pk_add_f32 %1, %4, %6\n\tv_pk_add_f32 %2, %4, %7\n\tv_pk_add_f32 %3, %4, %8\n\ts_nop 1"
                   : "=&v"(w01), "=&v"(w23), "=&v"(w45), "=&v"(w67) : "v"(cc), "v"(p00), "v"(p10), "v"(p01), "v"(p11));
    } else if constexpr (M == 3) {
      asm volatile("v_pk_fma_f32 %0, %4, %5, 0\n\tv_pk_fma_f32 %1, %4, %6, 0\n\tv_pk_fma_f32 %2, %4, %7, 0\n\tv_pk_fma_f32 %3, %4, %8, 0\n\ts_nop 1"
                   : "=&v"(w01), "=&v"(w23), "=&v"(w45), "=&v"(w67) : "v"(cc), "v"(p00), "v"(p10), "v"(p01), "v"(p11));
    } else {
      asm volatile("v_mul_f32 %0, %8, %10\n\tv_mul_f32 %1, %9, %11\n\tv_mul_f32 %2, %8, %12\n\tv_mul_f32 %3, %9, %13\n\t"
                   "v_mul_f32 %4, %8, %14\n\tv_mul_f32 %5, %9, %15\n\tv_mul_f32 %6, %8, %16\n\tv_mul_f32 %7, %9, %17\n\ts_nop 1"
                   : "=&v"(w01.x), "=&v"(w01.y), "=&v"(w23.x), "=&v"(w23.y), "=&v"(w45.x), "=&v"(w45.y), "=&v"(w67.x), "=&v"(w67.y)
                   : "v"(cc.x), "v"(cc.y), "v"(p00.x), "v"(p00.y), "v"(p10.x), "v"(p10.y), "v"(p01.x), "v"(p01.y), "v"(p11.x), "v"(p11.y));
    }
    // expected values: exact multiples of 1/64, so association and fusion cannot matter
    const float na = 1.f - a, nb = 1.f - b, nc = 1.f - c;
    const float ep[4] = {nb * na, b * na, nb * a, b * a};
    const float gp[8] = {p00.x, p00.y, p10.x, p10.y, p01.x, p01.y, p11.x, p11.y};
    const float gw[8] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y, w67.x, w67.y};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (__float_as_uint(gp[k]) != __float_as_uint(ep[k >> 1])) { bad |= 1u << k; ++nbad; }
      const float ew = M == 6 ? ep[k >> 1] + ((k & 1) ? c : nc) : ep[k >> 1] * ((k & 1) ? c : nc);
      if (__float_as_uint(gw[k]) != __float_as_uint(ew)) { bad |= 1u << (8 + k); ++nbad; }
    }
    a = a == 0.25f ? 0.75f : 0.25f; b = b == 0.75f ? 0.25f : 0.75f; c = (it & 2) ? 0.25f : 0.75f;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_probe[0], (unsigned)iters);
  if (nbad) {
    atomicAdd(&g_probe[1], nbad); atomicAdd(&g_probe[2 + (lane >> 4)], nbad);
    for (int k = 0; k < 16; ++k) if (bad >> k & 1) atomicAdd(&g_probe[6 + k], 1u);
  }
}

struct Layer { Desc d; uint16_t *x, *w, *y; float* ws; int64_t ws_floats; };
static Desc mk(int N, int T, int H, int W, int C, int K, int kT, int kH, int kW, int sT, int sH, int sW) {
  Desc d; d.N = N; d.T = T; d.H = H; d.W = W; d.C = C; d.K = K; d.kT = kT; d.kH = kH; d.kW = kW; d.sT = sT; d.sH = sH; d.sW = sW;
  d.pT = (kT - 1) / 2; d.pH = (kH - 1) / 2; d.pW = (kW - 1) / 2;
  d.To = (T + 2 * d.pT - kT) / sT + 1; d.Ho = (H + 2 * d.pH - kH) / sH + 1; d.Wo = (W + 2 * d.pW - kW) / sW + 1;
  return d;
}
static Layer alloc_layer(const Desc& d, uint16_t* x) {
  Layer l; l.d = d;
  const size_t nx = (size_t)d.N * d.T * d.H * d.W * d.C, ny = (size_t)d.N * d.To * d.Ho * d.Wo * d.K;
  l.x = x ? x : dev_bf16(nx, 1.0f);
  l.w = dev_bf16((size_t)d.K * d.kT * d.kH * d.kW * d.C, 1.0f / std::sqrt((float)(d.C * d.kT * d.kH * d.kW)));
  CHECK(hipMalloc(&l.y, ny * 2)); CHECK(hipMemset(l.y, 0, ny * 2));
  l.ws_floats = (int64_t)ny * 16; CHECK(hipMalloc(&l.ws, l.ws_floats * 4));
  return l;
}
// a trunk's forward convs: stem, then four stages of [entry (stride s) + shortcut, 3 more convs]
static std::vector<Layer> trunk(bool flow) {
  std::vector<Layer> L;
  const int N = flow ? 4 : 2, base = flow ? 16 : 64, kT = flow ? 1 : 3;
  Desc s = flow ? mk(N, 8, 32, 32, 8, 16, 1, 7, 7, 2, 2, 2) : mk(N, 8, 32, 32, 8, 64, 3, 7, 7, 1, 2, 2);
  L.push_back(alloc_layer(s, nullptr));
  int T = s.To, H = s.Ho, W = s.Wo, cin = base;
  uint16_t* cur = L.back().y;
  for (int li = 1; li <= 4; ++li) {
    const int cout = base << (li - 1), st = li == 1 ? 1 : 2, stT = flow ? 1 : st;
    Desc e = mk(N, T, H, W, cin, cout, kT, 3, 3, stT, st, st);
    L.push_back(alloc_layer(e, cur));
    if (st != 1) L.push_back(alloc_layer(mk(N, T, H, W, cin, cout, 1, 1, 1, stT, st, st), cur));
    T = e.To; H = e.Ho; W = e.Wo; cin = cout; cur = L[L.size() - (st != 1 ? 2 : 1)].y;
    for (int j = 0; j < 3; ++j) { L.push_back(alloc_layer(mk(N, T, H, W, cin, cout, kT, 3, 3, 1, 1, 1), cur)); cur = L.back().y; }
  }
  return L;
}

int main(int argc, char** argv) {
  std::string libp = "mscl_amd/csrc/libmscl_hip.so", diagp, side = "both";
  long replays = 2000; int reps = 20; bool eager = false, one_stream = false, probe = false; int pmode = 0;
  int lay_lo = 0, lay_hi = 1000; std::string streams = "BC";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--lib" && i + 1 < argc) libp = argv[++i];
    else if (a == "--diag" && i + 1 < argc) diagp = argv[++i];
    else if (a == "--replays" && i + 1 < argc) replays = atol(argv[++i]);
    else if (a == "--reps" && i + 1 < argc) reps = atoi(argv[++i]);
    else if (a == "--side" && i + 1 < argc) side = argv[++i];
    else if (a == "--eager") eager = true;
    else if (a == "--one-stream") one_stream = true;
    else if (a == "--probe") { probe = true; if (i + 1 < argc && argv[i + 1][0] != '-') pmode = atoi(argv[++i]); }
    else if (a == "--streams" && i + 1 < argc) streams = argv[++i];                       // which side chains run: B, C or BC
    else if (a == "--layers" && i + 1 < argc) { sscanf(argv[++i], "%d:%d", &lay_lo, &lay_hi); }   // side layers [lo, hi) of each trunk list
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  void* h = dlopen(libp.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp.c_str(), dlerror()); return 2; }
  auto conv_fwd = (conv_fwd_t)dlsym(h, "mscl_conv3d_fwd");
  auto upsample = (upsample_t)dlsym(h, "mscl_upsample_add");
  auto add_relu = (add_relu_t)dlsym(h, "mscl_add_relu");
  auto set_det = (set_det_t)dlsym(h, "mscl_set_deterministic");
  if (!conv_fwd || !upsample || !add_relu || !set_det) { fprintf(stderr, "missing symbols in %s\n", libp.c_str()); return 2; }
  typedef int (*diag_read_t)(unsigned*, unsigned*, int);
  diag_read_t diag_read = nullptr; int diag_words = 0;
  if (!diagp.empty()) {
    void* hd = dlopen(diagp.c_str(), RTLD_NOW);
    if (!hd) { fprintf(stderr, "dlopen %s: %s\n", diagp.c_str(), dlerror()); return 2; }
    upsample = (upsample_t)dlsym(hd, "ups_diag_upsample_add");
    diag_read = (diag_read_t)dlsym(hd, "ups_diag_read");
    diag_words = ((int (*)())dlsym(hd, "ups_diag_rec_words"))();
  }
  set_det(1);
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; lib %s; diag %s; replays %ld x reps %d; side %s; %s%s\n", prop.name, prop.multiProcessorCount, libp.c_str(),
         diagp.empty() ? "-" : diagp.c_str(), replays, reps, side.c_str(), eager ? "eager, three host threads" : "one captured graph per replay",
         one_stream ? ", ONE stream" : "");

  // stream A's work
  Layer p0 = alloc_layer(mk(2, 2, 4, 4, 128, 128, 3, 3, 3, 1, 1, 1), nullptr);
  float* bias; CHECK(hipMalloc(&bias, 128 * 4)); CHECK(hipMemset(bias, 0, 128 * 4));
  const int Td = 4, Hd = 8, Wd = 8, C = 128, rows = 2 * Td * Hd * Wd, n16 = rows * C / 8;
  uint16_t *dst, *ref; CHECK(hipMalloc(&dst, (size_t)rows * C * 2)); CHECK(hipMalloc(&ref, (size_t)rows * C * 2));
  std::vector<Layer> LB = trunk(false), LC = trunk(true);
  hipStream_t sA, sB, sC;
  CHECK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
  if (one_stream) { sB = sA; sC = sA; }
  else { CHECK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking)); }

  auto chainA = [&](int n, bool compare) {
    if (probe) {
      for (int r = 0; r < n; ++r) {
        switch (pmode) {
          case 0: hipLaunchKernelGGL(valu_probe_kernel<0>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          case 1: hipLaunchKernelGGL(valu_probe_kernel<1>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          case 2: hipLaunchKernelGGL(valu_probe_kernel<2>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          case 3: hipLaunchKernelGGL(valu_probe_kernel<3>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          case 4: hipLaunchKernelGGL(valu_probe_kernel<4>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          case 6: hipLaunchKernelGGL(valu_probe_kernel<6>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
          default: hipLaunchKernelGGL(valu_probe_kernel<5>, dim3(32), dim3(256), 0, sA, 0.25f, 0.75f, 0.25f, 64); break;
        }
      }
      return;
    }
    for (int r = 0; r < n; ++r) {
      int rc = conv_fwd(&p0.d, p0.x, p0.w, p0.y, bias, nullptr, 0, nullptr, nullptr, p0.ws, p0.ws_floats, sA);
      if (rc) { fprintf(stderr, "conv_fwd -> %d\n", rc); exit(2); }
      rc = upsample(p0.y, dst, 2, 2, 4, 4, Td, Hd, Wd, C, 1, 0, sA);
      if (rc) { fprintf(stderr, "upsample -> %d\n", rc); exit(2); }
      if (compare) hipLaunchKernelGGL(cmp_kernel, dim3(1), dim3(256), 0, sA, (const uint4*)dst, (const uint4*)ref, n16);
    }
  };
  auto chainS = [&](std::vector<Layer>& L, hipStream_t st) {
    if (side == "none") return;
    if ((&L == &LB && streams.find('B') == std::string::npos) || (&L == &LC && streams.find('C') == std::string::npos)) return;
    int li = -1;
    for (auto& l : L) {
      ++li;
      if (li < lay_lo || li >= lay_hi) continue;
      const int64_t ny = (int64_t)l.d.N * l.d.To * l.d.Ho * l.d.Wo * l.d.K;
      if (side != "elem") {
        const int rc = conv_fwd(&l.d, l.x, l.w, l.y, nullptr, nullptr, 0, nullptr, nullptr, l.ws, l.ws_floats, st);
        if (rc) { fprintf(stderr, "side conv_fwd (C %d K %d k %d%d%d) -> %d\n", l.d.C, l.d.K, l.d.kT, l.d.kH, l.d.kW, rc); exit(2); }
      }
      if (side != "convs") {
        const int rc = add_relu(l.y, nullptr, nullptr, l.y, ny, 1, st);
        if (rc) { fprintf(stderr, "add_relu -> %d\n", rc); exit(2); }
      }
    }
  };

  chainS(LB, sB); chainS(LC, sC);            // warm-up outside the capture (lazy kernel attributes)
  CHECK(hipDeviceSynchronize());
  // reference: the first launch's output, itself checked against a CPU evaluation of conv + trilinear up-sampling
  chainA(1, false);
  CHECK(hipStreamSynchronize(sA));
  CHECK(hipMemcpy(ref, dst, (size_t)rows * C * 2, hipMemcpyDeviceToDevice));
  if (!probe) {
    const Desc& d = p0.d;
    std::vector<uint16_t> hx((size_t)64 * 128), hw((size_t)128 * 27 * 128), hy((size_t)64 * 128), hd((size_t)rows * C);
    CHECK(hipMemcpy(hx.data(), p0.x, hx.size() * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hw.data(), p0.w, hw.size() * 2, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hy.data(), p0.y, hy.size() * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hd.data(), dst, hd.size() * 2, hipMemcpyDeviceToHost));
    double worst_y = 0, worst_d = 0, ymax = 0;
    std::vector<float> y((size_t)64 * 128);
    for (int n = 0; n < 2; ++n) for (int t = 0; t < 2; ++t) for (int hh = 0; hh < 4; ++hh) for (int ww = 0; ww < 4; ++ww) for (int k = 0; k < 128; ++k) {
      double acc = 0;
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int c = 0; c < 3; ++c) {
        const int ti = t + a - 1, hi = hh + b - 1, wi = ww + c - 1;
        if (ti < 0 || ti >= 2 || hi < 0 || hi >= 4 || wi < 0 || wi >= 4) continue;
        const uint16_t* xp = &hx[((((size_t)n * 2 + ti) * 4 + hi) * 4 + wi) * 128];
        const uint16_t* wp = &hw[(((size_t)k * 3 + a) * 3 + b) * 3 * 128 + (size_t)c * 128];
        for (int ci = 0; ci < 128; ++ci) acc += (double)bf2f(xp[ci]) * bf2f(wp[ci]);
      }
      const size_t o = ((((size_t)n * 2 + t) * 4 + hh) * 4 + ww) * 128 + k;
      y[o] = (float)acc; ymax = std::fmax(ymax, std::fabs(acc));
      worst_y = std::fmax(worst_y, std::fabs(acc - bf2f(hy[o])));
    }
    auto lin = [](int dd, int in, int out, int& i0, int& i1, float& w1) {
      const float sc = (float)in / (float)out; float s = ((float)dd + 0.5f) * sc - 0.5f; s = s < 0.f ? 0.f : s;
      i0 = (int)s; if (i0 > in - 1) i0 = in - 1; i1 = i0 + 1 > in - 1 ? in - 1 : i0 + 1; w1 = s - (float)i0; };
    for (int n = 0; n < 2; ++n) for (int t = 0; t < Td; ++t) for (int hh = 0; hh < Hd; ++hh) for (int ww = 0; ww < Wd; ++ww) {
      int t0, t1, h0, h1, w0, w1; float a, b, c; lin(t, 2, Td, t0, t1, a); lin(hh, 4, Hd, h0, h1, b); lin(ww, 4, Wd, w0, w1, c);
      for (int k = 0; k < 128; ++k) {
        double acc = 0;
        for (int q = 0; q < 8; ++q) {
          const int tt = (q & 4) ? t1 : t0, h2 = (q & 2) ? h1 : h0, w2 = (q & 1) ? w1 : w0;
          const float wt = ((q & 4) ? a : 1.f - a) * ((q & 2) ? b : 1.f - b) * ((q & 1) ? c : 1.f - c);
          acc += (double)wt * bf2f(hy[((((size_t)n * 2 + tt) * 4 + h2) * 4 + w2) * 128 + k]);
        }
        worst_d = std::fmax(worst_d, std::fabs(acc - bf2f(hd[((((size_t)n * Td + t) * Hd + hh) * Wd + ww) * 128 + k])));
      }
    }
    printf("reference launch vs CPU: conv max |err| %.4g (max |y| %.3g), up-sampling max |err| %.4g -> %s\n", worst_y, ymax, worst_d,
           (worst_y <= ymax / 128 && worst_d <= ymax / 128) ? "ok" : "MISMATCH");
    if (!(worst_y <= ymax / 128 && worst_d <= ymax / 128)) return 3;
  }

  hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  CHECK(hipEventRecord(t0, sA));
  if (!eager) {
    hipEvent_t fork, jB, jC; CHECK(hipEventCreate(&fork)); CHECK(hipEventCreate(&jB)); CHECK(hipEventCreate(&jC));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(sA, hipStreamCaptureModeGlobal));
    if (!one_stream) {
      CHECK(hipEventRecord(fork, sA)); CHECK(hipStreamWaitEvent(sB, fork, 0)); CHECK(hipStreamWaitEvent(sC, fork, 0));
    }
    chainS(LB, sB); chainS(LC, sC);
    chainA(reps, true);
    if (!one_stream) {
      CHECK(hipEventRecord(jB, sB)); CHECK(hipEventRecord(jC, sC)); CHECK(hipStreamWaitEvent(sA, jB, 0)); CHECK(hipStreamWaitEvent(sA, jC, 0));
    }
    CHECK(hipStreamEndCapture(sA, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (long r = 0; r < replays; ++r) {
      CHECK(hipGraphLaunch(ge, sA));
      if (r % 500 == 499) { CHECK(hipStreamSynchronize(sA)); printf("replay %ld\n", r + 1); fflush(stdout); }
    }
  } else {
    std::thread tb([&] { for (long r = 0; r < replays; ++r) { chainS(LB, sB); if (r % 64 == 63) CHECK(hipStreamSynchronize(sB)); } });
    std::thread tc([&] { for (long r = 0; r < replays; ++r) { chainS(LC, sC); if (r % 64 == 63) CHECK(hipStreamSynchronize(sC)); } });
    for (long r = 0; r < replays; ++r) {
      chainA(reps, true);
      if (r % 64 == 63) CHECK(hipStreamSynchronize(sA));
      if (r % 500 == 499) { printf("round %ld\n", r + 1); fflush(stdout); }
    }
    tb.join(); tc.join();
  }
  CHECK(hipEventRecord(t1, sA));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, t0, t1));
  unsigned cmp[4], bad[256 * 4];
  CHECK(hipMemcpyFromSymbol(cmp, HIP_SYMBOL(g_cmp), sizeof(cmp))); CHECK(hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof(bad)));
  const unsigned nb = cmp[2] < 256 ? cmp[2] : 256;
  for (unsigned i = 0; i < nb; ++i) {
    const unsigned gi = bad[4 * i + 1];
    printf("  differing granule: comparison %u, output row %u (w %u), channel granule %u (lane %u of its wave), got %08x want %08x\n", bad[4 * i],
           gi / 16, (gi / 16) % Wd, gi % 16, gi % 64, bad[4 * i + 2], bad[4 * i + 3]);
  }
  if (diag_read) {
    std::vector<unsigned> rec((size_t)2048 * diag_words); unsigned cnt[8];
    const int n = diag_read(cnt, rec.data(), 2048);
    printf("diag records %u (kept %d), diag launches %u\n", cnt[0], n, cnt[1]);
    for (int i = 0; i < n && i < 64; ++i) {
      const unsigned* R = &rec[(size_t)i * diag_words];
      printf("  rec %d: off %02x wt %02x load %02x zeroA %02x zeroB %02x res %02x | launch %u block %u thread %u HW_ID %08x XCC %u corner %u | A %08x B %08x C %08x\n", i,
             R[0] & 255, (R[0] >> 8) & 255, (R[0] >> 16) & 255, R[0] >> 24, R[1] & 255, (R[1] >> 8) & 255, R[2], R[3], R[4], R[6], R[7] & 15, R[10], R[15], R[19], R[23]);
    }
  }
  if (probe) {
    unsigned pr[32]; CHECK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr)));
    printf("PROBE form %d: iterations/lane %u, lanes x iterations %.3g, mismatching halves %u; by lane quarter %u %u %u %u; producers p00.lo..p11.hi %u %u %u %u %u %u %u %u; "
           "consumers w0..w7 %u %u %u %u %u %u %u %u\n", pmode, pr[0], (double)pr[0] * 32 * 256, pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11],
           pr[12], pr[13], pr[14], pr[15], pr[16], pr[17], pr[18], pr[19], pr[20], pr[21]);
  }
  printf("SUMMARY comparisons %u differing %u differing_granules %u elapsed_ms %.1f us_per_rep %.2f\n", cmp[0], cmp[1], cmp[2], ms,
         1e3 * ms / ((double)replays * reps));
  return cmp[1] ? 1 : 0;
}
