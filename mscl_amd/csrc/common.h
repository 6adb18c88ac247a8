// Shared device helpers for the gfx950 kernels (wave = 64 lanes, bf16 storage as uint16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mscl_hip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept a NaN (the compiler lowers the cast to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xFFFF0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15), result in every lane of the row.  VALU + DPP only:
// __shfl_xor compiles to ds_bpermute_b32 (an LDS crossbar instruction with LDS latency), 128 of them per wave in a conv
// epilogue that reduces BatchNorm statistics over the 16 positions a row of lanes holds.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
  return v;
}

// The same sums for NV values at once, as a reduce-scatter: the first two butterfly steps (distance 8, distance 4) HALVE the
// values a lane carries -- each half of the row, then each quad, keeps its own share and receives the partner's through one
// bank-masked DPP move per direction -- and only the quarter that is left goes through the two in-quad steps.  On return every
// lane of quad q (lanes 4q .. 4q + 3 of its row) holds in v[0 .. NV/4) the row totals of values q * NV/4 .. (q + 1) * NV/4 - 1.
// 32 values: 60 DPP moves + 40 adds instead of 128 + 128 (the statistics epilogue of a conv block is VALU time: 13 such
// epilogues per SIMD and launch on the layer-1 map).
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_take(float old, float src) {      // enabled banks: src of the DPP partner lane; others: old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xF, BANKS, false));
}
template <int NV>
__device__ __forceinline__ void row16_reduce_scatter(float (&v)[NV]) {
  static_assert(NV % 4 == 0, "values per lane");
  constexpr int H = NV / 2, Q = NV / 4;
  // distance 8 (row_ror:8): lanes 0-7 (banks 0, 1) go on with values [0, H), lanes 8-15 (banks 2, 3) with [H, NV)
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const float a = dpp_take<0x128, 0x3>(v[H + k], v[k]);         // lanes 0-7: the partner's v[k];     lanes 8-15: own v[H + k]
    const float b = dpp_take<0x128, 0xC>(v[k], v[H + k]);         // lanes 8-15: the partner's v[H + k]; lanes 0-7: own v[k]
    v[k] = a + b;
  }
  // distance 4: lanes 0-3 / 8-11 (banks 0, 2; partner through row_shl:4) go on with [0, Q) of their half, lanes 4-7 / 12-15 (banks
  // 1, 3; row_shr:4) with [Q, H)
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const float a = dpp_take<0x104, 0x5>(v[Q + k], v[k]);
    const float b = dpp_take<0x114, 0xA>(v[k], v[Q + k]);
    v[k] = a + b;
  }
  // inside the quad: plain butterfly
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0xB1, 0xF, 0xF, false));
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x4E, 0xF, 0xF, false));
  }
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)));
  return v;
}
// whole-wave reductions, result wave-uniform: DPP inside the four rows, then one v_readlane per row
__device__ __forceinline__ float wave_sum(float v) {
  const int r = __builtin_bit_cast(int, row16_sum(v));
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
  const int r = __builtin_bit_cast(int, row16_max(v));
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48))));
}

// Block-wide per-channel sum for the "G = C/8 threads per row, 256/G rows per pass" thread layout used by the
// channel-reduction kernels: lanes that own the same channel granule sit G apart inside a wave (G < 64), so
// they are combined with shuffles, then the (<= 4) waves are combined through LDS with plain stores --
// no LDS atomics (64-way contention cost ~10 us per launch on small maps).  v[8] are this thread's partial
// sums for channels c0..c0+7; on return red[c] holds the block total for every channel (after a barrier).
__device__ __forceinline__ void block_channel_sum(float* v, float* red, int G, int C, int nvec, int vec) {
  // red layout: [nvec][4 waves][C]
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += __shfl_xor(v[i], o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tg = threadIdx.x % G;
  if (G >= 64 || lane < G) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(vec * 4 + wave) * C + tg * 8 + i] = v[i];
  }
}

// XCD-aware, bijective remap of a 1-D block id: blocks that share an XCD (bid % 8) get a contiguous
// run of logical ids, so neighbouring tiles (shared halos / weight panels) meet in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

extern int g_mscl_deterministic;
static inline int mscl_det_flag_value() { return g_mscl_deterministic; }
// BatchNorm statistics are accumulated into MSCL_STAT_SLOTS copies of the [2][C] sums (slot = block index mod slots;
// consumers add the slots up): thousands of blocks adding into ONE 512-byte row run an order of magnitude below the
// float-atomic rate (MI355X_MICROARCH.md, Global float atomics, row 'contention').  Slot stride = 2 * C floats.
#ifndef MSCL_STAT_SLOTS
#define MSCL_STAT_SLOTS 16
#endif
// Of the MSCL_STAT_SLOTS slots a statistics buffer holds, the atomic producers spread over the first MSCL_STAT_ACTIVE only and the
// consumers add just those: every block of a consuming pass re-reads all active slots of every channel in its prologue, which costs
// more than the contention the extra slots avoid (measured inside the step with 1 / 2 / 4 / 8 / 16 / 32 slots: 948 / 964 / 970 / 971 /
// 956 / 897 clip-pairs/s).  The deterministic mode's producers plain-store per-block partials (up to MSCL_DET_PARTS of them, in
// caller scratch) and fold them in index order into slot 0 (bn_act.hip, det_fold_kernel): its consumers read slot 0 alone.  The
// `nslots` argument of the consuming kernels says which.
#ifndef MSCL_STAT_ACTIVE
#define MSCL_STAT_ACTIVE 4
#endif
#define MSCL_DET_PARTS 128
static inline int mscl_stat_nslots() { return mscl_det_flag_value() ? 1 : MSCL_STAT_ACTIVE; }

// exact floor(n / d) for 0 <= n < 2^31 via one 32x32->64 multiply (Granlund-Montgomery round-up method)
struct FastDiv { uint32_t magic; int shift; };
static inline FastDiv make_fastdiv(int d) {
  int L = 0; while ((1 << L) < d) ++L;
  FastDiv f; f.shift = 31 + L;
  f.magic = (uint32_t)((((uint64_t)1) << f.shift) / (uint64_t)d + 1);
  return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)(((uint64_t)(uint32_t)n * f.magic) >> f.shift); }

// Deterministic mode (mscl_set_deterministic, include/mscl_hip.h): every fp32 sum whose order the hardware would otherwise pick
// (float atomics between blocks) is taken in a fixed order instead, so two runs on the same inputs are bit-identical.
static inline bool mscl_det() { return g_mscl_deterministic != 0; }

// Tuning switches (MSCL_* environment variables, listed in INTEGRATION.md) are read ONCE and cached: getenv per launch costs the
// host-bound eager step, and a captured graph bakes in whatever was read at capture time anyway.  mscl_tuning_reload() (C ABI)
// makes every switch re-read its variable at its next use: tests and A/B sweeps that flip a switch inside one process call it.
extern int g_mscl_tune_gen;
struct MsclTune {
  const char* name; int gen; bool set; int val; char c0;
  explicit MsclTune(const char* n) : name(n), gen(-1), set(false), val(0), c0(0) {}
  bool read() {
    if (gen != g_mscl_tune_gen) {
      const char* e = getenv(name);
      set = e != nullptr; val = e ? atoi(e) : 0; c0 = e ? e[0] : 0; gen = g_mscl_tune_gen;
    }
    return set;
  }
  int get(int dflt) { return read() ? val : dflt; }
};

#define MSCL_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
static inline int ilog2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
