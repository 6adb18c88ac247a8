// Halo-resident 3x3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels (R3D-18 layer 1: 44 % of the
// trunk's FLOPs; forward AND stride-1 input gradient) on bf16 MFMA, gfx950.
//
// Why a second conv kernel: the implicit-GEMM kernel (conv_igemm.hip) re-stages every input row once per
// tap, 27x.  With only 64 output channels a staged byte feeds 64 FLOP, and the measured L2 -> LDS gather rate
// of a CU (~30 B/clk) then caps the kernel near 45 % of MFMA peak (it reaches ~500 TFLOP/s = 20 %).
// Here a block owns BM = 256 consecutive output positions of ONE (n, t) plane and stages the input window
// they can touch ONCE: for each of the 3 source planes the linear range [p0 - W - 1, p0 + BM + W + 1)
// (BM + 2W + 2 rows of 128 B) = 142 KB of LDS at W = 56.  A tap is then just a constant row offset into that
// window -- the 27 taps read their MFMA operands straight from it.  Staged bytes per output drop 6x.
//  * rows outside the plane / planes outside the clip are zero-filled by the DMA's out-of-range rule;
//  * the only positions the linear window gets wrong are the w = 0 / w = W-1 columns for the kw = 0 / 2 taps
//    (the neighbour in memory belongs to the adjacent image row): those lanes' fragments are zeroed (v_cndmask);
//  * weights (8 KB per tap) stream through a double-buffered LDS tile by LDS-DMA, one barrier per tap;
//  * 4 waves, 64 x 64 outputs each (LDS reads: 8 x ds_read_b128 per 16 MFMAs = 128 B/clk/CU, half the LDS rate);
//  * epilogue as in conv_igemm.hip: BatchNorm sum / sum-of-squares, optional addend, bf16, 8-byte stores.
#include "common.h"

struct HaloGeom {
  int N, T, H, W, HW, NH;          // NH = BM + 2W + 2 window rows per source plane
  int tiles, mode;                 // tiles per plane; mode 0 forward, 1 input gradient (taps mirrored)
  FastDiv dW;                      // division by W
};

__device__ __forceinline__ auto halo_rsrc(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

constexpr int HBM = 256;           // output positions per block
constexpr int HC = 64;             // channels (in = out)
constexpr int PF = 4;              // weight fragments are prefetched this many taps ahead (registers)

// Wave layout 2 (positions) x 2 (channels): a wave owns 128 positions x 32 output channels.
//  * A operands (positions) come from the LDS window: 8 fragments per 32-deep k step, read 4 at a time;
//  * B operands (weights, 32 rows x 64 k per tap = 4 fragments) are loaded from global memory / L2 straight
//    into registers PF taps ahead -- no weight tile in LDS, hence NO barrier inside the 27-tap loop: the only
//    barriers are the three "source plane landed" points, so planes 1 and 2 stream in under plane 0's taps.
__global__ __launch_bounds__(256, 1) void conv_halo64_kernel(const HaloGeom g, const bf16_t* __restrict__ src,
                                                             const bf16_t* __restrict__ wgt, bf16_t* __restrict__ out,
                                                             const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                                             float* __restrict__ stat_sq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Hs = smem;                           // [3][NH][128 B] input window
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = bid % g.tiles, plane = bid / g.tiles;        // plane = n*T + t
  const int t = plane % g.T;
  const int p0 = tile * HBM;
  const int mode = __builtin_amdgcn_readfirstlane(g.mode);

  const unsigned src_bytes = (unsigned)((long)g.N * g.T * g.HW * HC * 2);
  const auto rs_src = halo_rsrc(src, src_bytes);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  // ---- weight fragments: rows n = 32*wn + 16*j + fr, k granule = ks*4 + fq ----
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_t* wbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wbase[j] = wgt + ((long)(32 * wn + 16 * j + fr) * 27) * HC + fq * 8;
  bf16x8_t bq[PF][2][2];                              // [tap slot][j][ks]
  auto load_b = [&](int tap, int slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bq[slot][j][ks] = *reinterpret_cast<const bf16x8_t*>(wbase[j] + tap * HC + ks * 32);
  };

  // ---- stage the input window plane by plane, in the order the tap loop visits them.  NH is padded to a
  // multiple of 32 rows so that every wave issues exactly `npass` DMA instructions per plane: the first wait
  // below can then be a COUNTED vmcnt that leaves the two later planes in flight under the first taps.
  // Issue order (vmcnt retires in order): plane A, weight prefetch, plane B, plane C.
  const int npass = g.NH >> 5;                         // NH*8 granules / 256 per pass
  auto stage_plane = [&](int k3) {
    const int hp = mode ? 2 - k3 : k3;                 // window plane used by the k3-th group of taps
    const int tt = t + hp - 1;
    for (int ps = 0; ps < npass; ++ps) {
      const int G = ps * 256 + tid;
      const int j = G >> 3, pg = G & 7;
      const int lg = pg ^ (j & 7);                     // source-side swizzle keyed on the window row (any-offset conflict-free)
      const int q = p0 - g.W - 1 + j;
      const bool ok = (unsigned)tt < (unsigned)g.T && (unsigned)q < (unsigned)g.HW;
      const unsigned off = ok ? (unsigned)((((plane + hp - 1) * g.HW + q) * HC + lg * 8) * 2) : src_bytes;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(Hs + hp * g.NH * 128 + (ps * 256 + wave * 64) * 16), 16, off, 0, 0, 0);
    }
  };
  stage_plane(0);
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0) load_b(s0, s0);
  stage_plane(1);
  stage_plane(2);

  // ---- per-lane output rows: m = 128*wm + 16*i + fr ----
  int hq0[8];                       // window row of the CENTRE tap for fragment i
  unsigned okbits = 0;              // bit i: row valid, bit 8+i: w-1 neighbour exists, bit 16+i: w+1 neighbour exists
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = 128 * wm + 16 * i + fr;
    const int p = p0 + m;
    const int wcol = p - fdiv(p, g.dW) * g.W;
    hq0[i] = m + g.W + 1;
    const bool okrow = p < g.HW;
    okbits |= (okrow ? 1u : 0u) << i;
    okbits |= ((okrow && wcol >= 1) ? 1u : 0u) << (8 + i);
    okbits |= ((okrow && wcol <= g.W - 2) ? 1u : 0u) << (16 + i);
  }
  f32x4_t acc[2][8];                // [j: channel tile][i: position tile]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fully unrolled: ring slots and plane bases are compile-time constants (a rolled kt loop with PF = 3 measured
  // 10 % slower although it needs 428 instead of 512 VGPRs)
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int kt = tap / 9, t9 = tap % 9;
    if (tap == 0) {                 // first plane + weight prefetch landed; 2*npass younger DMA instructions stay in flight
      switch (2 * npass) {
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      __builtin_amdgcn_s_barrier();
    } else if (t9 == 0) {
      // planes B and C are older than every weight load issued inside the loop, and vmcnt retires in order: the
      // wait that delivered tap 4's weights already covered them for THIS wave; the barrier publishes that to all
      __builtin_amdgcn_s_barrier();
    }
    const int kw = t9 % 3, kh = t9 / 3;
    const int hp = mode ? 2 - kt : kt;
    const int dlt = mode ? (1 - kh) * g.W + (1 - kw) : (kh - 1) * g.W + (kw - 1);
    const int side = mode ? 2 - kw : kw;        // 0: reads the w-1 neighbour, 2: the w+1 neighbour, 1: centre
    const unsigned okm = side == 1 ? okbits : (side == 0 ? (okbits >> 8) : (okbits >> 16));
    const unsigned char* hb = Hs + hp * g.NH * 128;
    const int slot = tap % PF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8_t fa[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = half * 4 + ii;
          const int hq = hq0[i] + dlt;
          uint4 v = *reinterpret_cast<const uint4*>(hb + hq * 128 + (((ks * 4 + fq) ^ (hq & 7)) * 16));
          const bool ok = (okm >> i) & 1u;      // (a wave-uniform "skip if no lane is masked" branch was tried: it
          v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;   // splits the MFMA stream into tiny blocks, 25 % slower)
          fa[ii] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
            acc[j][half * 4 + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot][j][ks], fa[ii], acc[j][half * 4 + ii], 0, 0, 0);
      }
    }
    if (tap + PF < 27) load_b(tap + PF, slot);
  }
  __syncthreads();                  // the epilogue reuses the window memory

  // ---- epilogue: BatchNorm statistics (rows beyond the plane were zeroed above) ----
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [2][64]
    for (int i = tid; i < 2 * HC; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
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
          const int nl = 32 * wn + j * 16 + fq * 4 + r;
          atomicAdd(&red[nl], s[r]);
          atomicAdd(&red[HC + nl], q[r]);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < HC; i += 256) { atomicAdd(&stat_sum[i], red[i]); atomicAdd(&stat_sq[i], red[HC + i]); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (!((okbits >> i) & 1u)) continue;
    const long o0 = ((long)plane * g.HW + p0 + 128 * wm + 16 * i + fr) * HC;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 32 * wn + j * 16 + fq * 4;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if (addend != nullptr) {
        const uint2 av = *reinterpret_cast<const uint2*>(addend + o0 + n);
        v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
        v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
      }
      uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(out + o0 + n) = pv;
    }
  }
}

// returns 1 if launched, 0 if the shape is not covered (caller falls back to the implicit-GEMM kernel), <0 / >0 on error
extern "C" int mscl_conv_halo64(const mscl_conv_desc* d, int mode, const uint16_t* src, const uint16_t* w, uint16_t* out,
                                const uint16_t* addend, float* ssum, float* ssq, void* stream) {
  if (!d || !src || !w || !out) return MSCL_E_ARG;
  if (d->C != HC || d->K != HC || d->kT != 3 || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 ||
      d->pT != 1 || d->pH != 1 || d->pW != 1) return 0;
  HaloGeom g{};
  g.N = d->N; g.T = d->T; g.H = d->H; g.W = d->W; g.HW = d->H * d->W; g.NH = (HBM + 2 * d->W + 2 + 31) / 32 * 32;
  const size_t lds = (size_t)3 * g.NH * 128;
  if (lds > 160 * 1024 || (long)d->N * d->T * g.HW * HC * 2 >= (1L << 31)) return 0;
  g.tiles = (g.HW + HBM - 1) / HBM; g.mode = mode;
  g.dW = make_fastdiv(d->W);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_halo64_kernel, dim3((unsigned)(d->N * d->T * g.tiles)), dim3(256), lds, (hipStream_t)stream, g, src, w, out,
                     addend, ssum, ssq);
  MSCL_LAUNCH_CHECK();
  return 1;
}
