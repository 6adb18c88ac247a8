// BatchNorm3d (training) + ReLU + residual, forward and backward, on NDHWC bf16 maps (gfx950).
// HBM-bound passes: 16-byte (8-channel) accesses, lanes run along the channel axis so every wave
// touches whole contiguous rows; per-channel constants live in LDS.
#include "common.h"
#include <cstdlib>

struct BnDev {
  const float* sum; const float* sumsq; const float* gamma; const float* beta;
  float* rmean; float* rvar; int64_t* nbt; float* smean; float* sinv;
};

// `groups` statistics groups (see mscl_conv3d_fwd_groups): scale / shift are [groups][C]; the running statistics take the
// groups' updates one after the other, in group order -- what the reference's consecutive module calls do
__device__ __forceinline__ void bn_prepare(const BnDev& b, float* scale, float* shift, int C, float inv_n,
                                           float unbias, float eps, float momentum, bool writer, int groups, int nslots) {
  if (b.sum == nullptr) {           // evaluation mode (nn.BatchNorm3d.eval()): running statistics, nothing is written
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float sc = b.gamma[c] * rsqrtf(b.rvar[c] + eps);
      for (int gi = 0; gi < groups; ++gi) { scale[gi * C + c] = sc; shift[gi * C + c] = b.beta[c] - b.rmean[c] * sc; }
    }
    return;
  }
  // Every load of a channel -- the active statistics slots of every group, gamma / beta, the running statistics the writer block
  // updates -- goes out BEFORE the first use, two channels per trip: as a loop over the slots with the loads inside, this section was
  // 4 + 1 + 2 dependent memory round trips per channel trip (the compiler waits on each slot before it issues the next), i.e. most
  // of the 7-8 us a pass over a small map took.
  auto prep = [&](int cb) {
    float a[2][2][MSCL_STAT_ACTIVE], q[2][2][MSCL_STAT_ACTIVE], gam[2], bet[2], rm[2], rv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = cb + k * (int)blockDim.x < C ? cb + k * (int)blockDim.x : cb;      // (the second channel of a short trip repeats the first)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        if (gi >= groups) break;                           // (uniform: no load is waited for on the way)
        const float* gs = b.sum + (long)gi * MSCL_STAT_SLOTS * 2 * C;
        const float* gq = b.sumsq + (long)gi * MSCL_STAT_SLOTS * 2 * C;
#pragma unroll
        for (int sl = 0; sl < MSCL_STAT_ACTIVE; ++sl) {
          const int si = sl < nslots ? sl : 0;             // a valid slot either way; what lies beyond nslots is dropped below
          a[k][gi][sl] = gs[si * 2 * C + c]; q[k][gi][sl] = gq[si * 2 * C + c];
        }
      }
      gam[k] = b.gamma[c]; bet[k] = b.beta[c];
      rm[k] = writer ? b.rmean[c] : 0.f; rv[k] = writer ? b.rvar[c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = cb + k * (int)blockDim.x;
      if (c >= C) break;
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        if (gi >= groups) break;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int sl = 0; sl < MSCL_STAT_ACTIVE; ++sl) if (sl < nslots) { s1 += a[k][gi][sl]; s2 += q[k][gi][sl]; }
        const float mean = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
        const float inv = rsqrtf(var + eps);
        const float sc = gam[k] * inv;
        scale[gi * C + c] = sc; shift[gi * C + c] = bet[k] - mean * sc;
        if (writer) {
          b.smean[gi * C + c] = mean; b.sinv[gi * C + c] = inv;
          rm[k] = (1.f - momentum) * rm[k] + momentum * mean;
          rv[k] = (1.f - momentum) * rv[k] + momentum * var * unbias;
        }
      }
      if (writer) { b.rmean[c] = rm[k]; b.rvar[c] = rv[k]; }
    }
  };
  // (maps of up to 512 channels -- every map of the R3D-18 step -- make ONE trip: written without the loop, whose header would
  // wait for everything in flight, the caller's first data loads included)
  if (C <= 2 * (int)blockDim.x) { if ((int)threadIdx.x < C) prep(threadIdx.x); }
  else for (int cb = threadIdx.x; cb < C; cb += 2 * blockDim.x) prep(cb);
  if (writer && threadIdx.x == 0 && b.nbt) *b.nbt += groups;
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16_t* __restrict__ y, BnDev bn,
                                                         const bf16_t* __restrict__ res, BnDev rbn, int res_is_bn,
                                                         bf16_t* __restrict__ out, long rows, int C, float eps,
                                                         float momentum, int relu, int groups, int nslots) {
  extern __shared__ float sm[];          // scale[groups][C], shift, rscale, rshift
  const int GC = groups * C;
  float* scale = sm; float* shift = sm + GC; float* rscale = sm + 2 * GC; float* rshift = sm + 3 * GC;
  const long rows_g = rows / groups;     // rows of one statistics group (groups <= 2: rows below rows_g are group 0)
  const float inv_n = 1.f / (float)rows_g;
  const float unbias = rows_g > 1 ? (float)rows_g / (float)(rows_g - 1) : 1.f;
  const bool writer = blockIdx.x == 0;
  const int G = C >> 3;
  const long total = rows * G;
  const unsigned ebound = groups > 1 ? (unsigned)(rows_g * G) : 0xFFFFFFFFu;     // first granule of group 1
  // channel granule of an element: a 64-bit modulo by a runtime value costs ~100 instructions per 16 bytes moved, so the
  // index is kept in 32 bits (the launcher guarantees it fits) and G, a power of two on this path, becomes a mask
  const unsigned gmask = ((G & (G - 1)) == 0) ? (unsigned)(G - 1) : 0xFFFFFFFFu;
  const unsigned total32 = (unsigned)total, step32 = gridDim.x * blockDim.x;
  // UNR granules per trip with every load issued before the first use: a thread makes only ~6 trips on the largest map,
  // so one 16-byte load in flight per thread leaves the pass latency-bound (2.6 TB/s measured on the layer-1 map).
  // The FIRST trip's loads are issued before the per-channel constants are rebuilt from the statistics slots (bn_prepare: a
  // dependent chain of slot loads, rsqrt and an LDS round trip): on the maps of layers 3-4 a thread makes one trip, and the
  // launch was two memory round trips one after the other (7-8 us for 0.8-3 MB).
  constexpr int UNR = 4;
  uint4 v[UNR], rv[UNR];
  auto load_trip = [&](unsigned e0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned e = e0 + u * step32;
      const unsigned ec = e < total32 ? e : (e0 < total32 ? e0 : 0u);             // clamped: the duplicate is not stored
      v[u] = *reinterpret_cast<const uint4*>(y + (long)ec * 8);
      if (res != nullptr) rv[u] = *reinterpret_cast<const uint4*>(res + (long)ec * 8);
    }
  };
  unsigned e0 = blockIdx.x * blockDim.x + threadIdx.x;
  load_trip(e0);
  bn_prepare(bn, scale, shift, C, inv_n, unbias, eps, momentum, writer, groups, nslots);
  if (res_is_bn) bn_prepare(rbn, rscale, rshift, C, inv_n, unbias, eps, momentum, writer, groups, nslots);
  __syncthreads();
  for (; e0 < total32; e0 += step32 * UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned e = e0 + u * step32;
      if (e >= total32) break;
      const int gq = gmask != 0xFFFFFFFFu ? (int)(e & gmask) : (int)(e % (unsigned)G);
      float f[8]; unpack8(v[u], f);
      const int c0 = gq * 8 + (e >= ebound ? C : 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = f[i] * scale[c0 + i] + shift[c0 + i];
      if (res != nullptr) {
        float r[8]; unpack8(rv[u], r);
        if (res_is_bn) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += r[i] * rscale[c0 + i] + rshift[c0 + i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += r[i];
        }
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
      }
      *reinterpret_cast<uint4*>(out + (long)e * 8) = pack8(f);
    }
    if (e0 + step32 * UNR < total32) load_trip(e0 + step32 * UNR);
  }
}

extern "C" int mscl_bn_act_fwd(const uint16_t* y, const mscl_bn_params* bn, const uint16_t* residual,
                               const mscl_bn_params* res_bn, uint16_t* out, int64_t rows, int C, float eps,
                               float momentum, int relu, void* stream) {
  return mscl_bn_act_fwd_groups(y, bn, residual, res_bn, out, rows, C, eps, momentum, relu, 1, stream);
}

extern "C" int mscl_bn_act_fwd_groups(const uint16_t* y, const mscl_bn_params* bn, const uint16_t* residual,
                                      const mscl_bn_params* res_bn, uint16_t* out, int64_t rows, int C, float eps,
                                      float momentum, int relu, int groups, void* stream) {
  if (!y || !bn || !out || rows <= 0 || C <= 0) return MSCL_E_ARG;
  if (groups < 1 || groups > 2 || rows % groups != 0) return MSCL_E_SHAPE;
  if (C % 8 || C > 2048 || rows * (C / 8) >= (1LL << 31)) return MSCL_E_SHAPE;
  if (!bn->gamma || !bn->beta || !bn->running_mean || !bn->running_var) return MSCL_E_ARG;
  if ((bn->sum == nullptr) != (bn->sumsq == nullptr)) return MSCL_E_ARG;
  if (bn->sum && (!bn->save_mean || !bn->save_invstd)) return MSCL_E_ARG;       // training mode keeps mean / invstd for backward
  BnDev b{bn->sum, bn->sumsq, bn->gamma, bn->beta, bn->running_mean, bn->running_var, bn->num_batches_tracked,
          bn->save_mean, bn->save_invstd};
  BnDev rb{};
  int res_is_bn = 0;
  if (res_bn) {
    if (!residual || (res_bn->sum == nullptr) != (bn->sum == nullptr)) return MSCL_E_ARG;
    rb = BnDev{res_bn->sum, res_bn->sumsq, res_bn->gamma, res_bn->beta, res_bn->running_mean, res_bn->running_var,
               res_bn->num_batches_tracked, res_bn->save_mean, res_bn->save_invstd};
    res_is_bn = 1;
  }
  const long total = rows * (C / 8);
  // Grid cap 512 (two blocks per CU), was 2048: every block first rebuilds the per-channel scale / shift from the 16 statistics
  // slots (bn_prepare), so blocks are not free, and a pass that fills every wave slot of the chip shuts the other chains of the
  // step out.  Measured (caps 512 / 512 / 512 for forward / reduce / apply against 2048 / 1024 / 2048, alternating in one call):
  // step 951.6 vs 926.5-928.8 clip-pairs/s, R3D-18 trunk alone 1750 vs 1701 clips/s, SlowOnly-50 trunk at 8 x 32 x 224^2 424.7 vs
  // 400.4 clips/s; 256 and 384 measured like 512 on the step, 768-1024 in between.
  constexpr long fwd_cap = 512;                // (round 4, re-swept inside the step with the two-block layer-1 kernels: 256 / 512 / 1024 for each
                                               // of the three passes within +-0.5 % of one another; again after the constants' loads
                                               // went out together: 512 / 512 / 512 1127, forward 1024 / 2048 1124 / 1117, apply 1024 /
                                               // 2048 1121 / 1120, reduce 1024 1119, all 1024 1115 clip-pairs/s; 512 stays)
  constexpr long gpt = 1;                      // granules per thread (swept 1 / 2 / 4 inside the step: no gain)
  long blocks = (total + 256 * gpt - 1) / (256 * gpt); if (blocks > fwd_cap) blocks = fwd_cap;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * groups * C * sizeof(float),
                     (hipStream_t)stream, y, b, residual, rb, res_is_bn, out, (long)rows, C, eps, momentum, relu, groups, mscl_stat_nslots());
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---- BatchNorm statistics of a stored map in a fixed summation order (deterministic mode; also a stand-alone entry point) ----
// Two levels, no atomics anywhere.  Level 1, grid (P, channel chunks, groups): block x of a group reduces the x-th contiguous share
// of the group's rows and plain-stores its sums as partial x.  Level 2 (det_fold_kernel): one thread per channel adds the P
// partials in index order into slot 0.  P depends on the shape alone (mscl_det_parts), so the order of every sum is fixed.
// The partials live in caller scratch (`parts`); without it they live in the MSCL_STAT_SLOTS slots themselves (P = slots: 16 blocks
// for the whole map -- the round-2 form of this pass, 5x slower on the large maps) and the fold zeroes slots 1.. again.
__global__ __launch_bounds__(256) void bn_stats_det_kernel(const bf16_t* __restrict__ y, float* __restrict__ part, long rows, int C,
                                                           int ldc, long part_gstride) {
  extern __shared__ float sm[];     // red[2][4 waves][C]
  const int ch = blockIdx.y * C;
  y += ch + (long)blockIdx.z * rows * ldc;
  const long go = blockIdx.z * part_gstride + (long)blockIdx.x * 2 * ldc + ch;      // partial [group][x][2][ldc]
  const int G = C >> 3;
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 256 / G;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long rbeg = (long)blockIdx.x * per, rend = rbeg + per < rows ? rbeg + per : rows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int UNR = 4;
  for (long r0 = rbeg + tr; r0 < rend; r0 += (long)RP * UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = r0 + (long)u * RP;
      v[u] = *reinterpret_cast<const uint4*>(y + (r < rend ? r : r0) * ldc + tg * 8);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + (long)u * RP >= rend) break;
      float f[8]; unpack8(v[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += f[i]; q[i] += f[i] * f[i]; }
    }
  }
  block_channel_sum(s, sm, G, C, 2, 0);
  block_channel_sum(q, sm, G, C, 2, 1);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    part[go + i] = (sm[i] + sm[C + i]) + (sm[2 * C + i] + sm[3 * C + i]);
    part[go + ldc + i] = (sm[4 * C + i] + sm[5 * C + i]) + (sm[6 * C + i] + sm[7 * C + i]);
  }
}

// level 2 of every deterministic BatchNorm sum: dst[group][v][c] = sum over p of part[group][p][v][c], v < nvec, in a FIXED order:
// eight lanes per channel each add a contiguous eighth of the partials in index order (their loads in flight together: one
// thread adding 128 partials one after the other took 5 us, 140 times a step), then lane 0 adds the eight sums in lane order.
// part rows are [pvec][ldc] wide, dst rows [ldc]; dst_gstride = floats between the groups' slot 0.  zero_tail > 0: the partials
// ARE slots 0 .. P-1 of dst (in place): slots 1 .. P-1 are zeroed again so that consumers may add any number of slots.
__global__ __launch_bounds__(256) void det_fold_kernel(const float* __restrict__ part, float* __restrict__ dst, int P, int nvec, int pvec,
                                                       int ldc, long part_gstride, long dst_gstride, int zero_tail) {
  __shared__ float red[8][32];
  const int ix = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + ix;
  const bool live = i < nvec * ldc;
  const long pitch = (long)pvec * ldc;
  const float* src = part + blockIdx.y * part_gstride + i;
  const int per = (P + 7) >> 3;
  const int pe = min(P, (py + 1) * per);
  float t = 0.f;
  if (live) {
    int p = py * per;
    for (; p + 8 <= pe; p += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(p + u) * pitch];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; p < pe; ++p) t += src[p * pitch];
  }
  red[py][ix] = t;
  __syncthreads();
  if (py != 0 || !live) return;
  t = red[0][ix];
#pragma unroll
  for (int k = 1; k < 8; ++k) t += red[k][ix];
  float* d = dst + blockIdx.y * dst_gstride;
  d[i] = t;
  if (zero_tail)
    for (int q = 1; q < P; ++q) d[q * pitch + i] = 0.f;
}

// partial blocks of a deterministic pass over `rows_g` rows of a C-channel map: enough rows per block that its loads overlap
// (a few trips of the unrolled loop), at most MSCL_DET_PARTS
static inline int det_parts_of(long rows_g, int C) {
  const int Cc = C > 512 ? 512 : C, RP = 256 / (Cc / 8);
  long p = (rows_g + RP * 8 - 1) / (RP * 8);
  if (p > MSCL_DET_PARTS) p = MSCL_DET_PARTS;
  return p < 1 ? 1 : (int)p;
}

extern "C" int64_t mscl_det_parts_floats(int64_t rows, int C, int groups, int vecs) {
  if (rows <= 0 || C <= 0 || groups < 1 || vecs < 1) return 0;
  return (int64_t)groups * det_parts_of(rows / groups, C) * vecs * C;
}

extern "C" int mscl_bn_stats(const uint16_t* y, float* ssum, float* ssq, int64_t rows, int C, int groups, float* parts,
                             int64_t parts_floats, void* stream) {
  if (!y || !ssum || !ssq || rows <= 0 || C <= 0 || groups < 1 || rows % groups) return MSCL_E_ARG;
  if (C % 8 || ilog2_exact(C / 8) < 0 || C > 4096) return MSCL_E_SHAPE;
  if (ssq != ssum + C) return MSCL_E_ARG;           // [slot][2][C]: the two sums of a slot are adjacent rows
  const int Cc = C > 512 ? 512 : C;
  hipStream_t st = (hipStream_t)stream;
  const bool own = parts != nullptr && parts_floats >= mscl_det_parts_floats(rows, C, groups, 2);
  const int P = own ? det_parts_of(rows / groups, C) : MSCL_STAT_SLOTS;
  float* part = own ? parts : ssum;
  const long pgs = (long)P * 2 * C;                  // (in place: P = MSCL_STAT_SLOTS, the slots' own group stride)
  hipLaunchKernelGGL(bn_stats_det_kernel, dim3(P, C / Cc, groups), dim3(256), (size_t)8 * Cc * sizeof(float), st, y, part,
                     (long)(rows / groups), Cc, C, pgs);
  MSCL_LAUNCH_CHECK();
  hipLaunchKernelGGL(det_fold_kernel, dim3((2 * C + 31) / 32, groups), dim3(256), 0, st, (const float*)part, ssum, P, 2, 2, C, pgs,
                     (long)MSCL_STAT_SLOTS * 2 * C, own ? 0 : 1);
  MSCL_LAUNCH_CHECK();
  return 0;
}

// ---- backward pass 1: per-channel sum(dz), sum(dz*xhat) [and the same against the residual's xhat] ----
// scratch layout: [0:C) sum dz, [C:2C) sum dz*xhat, [2C:3C) sum dz*xhat_res
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, const bf16_t* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ inv, const bf16_t* __restrict__ ry,
    const float* __restrict__ rmean, const float* __restrict__ rinv, float* __restrict__ scratch, long rows, int C,
    int relu, const float* __restrict__ gamma, const float* __restrict__ beta, int ldc, float* __restrict__ det_part,
    long part_gstride) {
  extern __shared__ float sm[];     // red[3][4 waves][C]
  // maps wider than 512 channels (Bottleneck trunks, up to 2048) are cut into 512-channel chunks along blockIdx.y: C is the
  // chunk width the thread layout sees, ldc the row pitch of the maps and the channel count of the scratch layout
  {
    const int ch = blockIdx.y * C;
    dout += ch; y += ch; mean += ch; inv += ch; scratch += ch;
    if (out) out += ch;
    if (ry) { ry += ch; rmean += ch; rinv += ch; }
    if (gamma) gamma += ch;
    if (beta) beta += ch;
    // statistics groups along blockIdx.z: `rows` is the row count of ONE group; group z owns rows [z * rows, (z + 1) * rows),
    // the z-th [ldc] block of mean / invstd and the z-th [slots][4 * ldc] block of the scratch sums
    const long ro = (long)blockIdx.z * rows * ldc;
    dout += ro; y += ro;
    if (out) out += ro;
    if (ry) { ry += ro; rmean += blockIdx.z * ldc; rinv += blockIdx.z * ldc; }
    mean += blockIdx.z * ldc; inv += blockIdx.z * ldc;
    scratch += (long)blockIdx.z * MSCL_STAT_SLOTS * 4 * ldc;
    if (det_part) det_part += ch + blockIdx.z * part_gstride + (long)blockIdx.x * 4 * ldc;      // partial [group][x][4][ldc]
  }
  const int G = C >> 3;             // threads per row; 256 % G == 0 required (C/8 power of two)
  const int tg = threadIdx.x % G, tr = threadIdx.x / G, RP = 256 / G;
  const int c0 = tg * 8;
  float mu[8], iv[8], rmu[8], riv[8], msc[8], msh[8];
  // beta != NULL (no residual): the ReLU mask is recomputed as bn(y) > 0 with the forward's own arithmetic instead of being
  // read back from `out` -- one of the three maps of this pass
  const bool mask_y = relu && beta != nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = mean[c0 + i]; iv[i] = inv[c0 + i];
    rmu[i] = ry ? rmean[c0 + i] : 0.f; riv[i] = ry ? rinv[c0 + i] : 0.f;
    msc[i] = mask_y ? gamma[c0 + i] * iv[i] : 0.f;
    msh[i] = mask_y ? beta[c0 + i] - mu[i] * msc[i] : 0.f;
  }
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // UNR rows per trip, every load issued before the first use: the loop is latency-bound (a thread makes only ~12
  // trips on the largest map), so the trips must overlap their round trips to HBM / Infinity Cache
  constexpr int UNR = 4;
  const long stride = (long)gridDim.x * RP;
  for (long r0 = (long)blockIdx.x * RP + tr; r0 < rows; r0 += stride * UNR) {
    uint4 vd[UNR], vy[UNR], va[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long r = r0 + u * stride;
      const long o = (r < rows ? r : r0) * ldc + c0;        // clamped: the duplicate is masked out below
      vd[u] = *reinterpret_cast<const uint4*>(dout + o);
      vy[u] = *reinterpret_cast<const uint4*>(y + o);
      if (relu && !mask_y) va[u] = *reinterpret_cast<const uint4*>(out + o);
      if (ry) vr[u] = *reinterpret_cast<const uint4*>(ry + o);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r0 + u * stride >= rows) continue;
      float d[8], a[8], yy[8];
      unpack8(vd[u], d);
      unpack8(vy[u], yy);
      if (mask_y) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = (yy[i] * msc[i] + msh[i]) > 0.f ? d[i] : 0.f;
      } else if (relu) {
        unpack8(va[u], a);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = a[i] > 0.f ? d[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { s0[i] += d[i]; s1[i] += d[i] * (yy[i] - mu[i]) * iv[i]; }
      if (ry) {
        float rr[8]; unpack8(vr[u], rr);
#pragma unroll
        for (int i = 0; i < 8; ++i) s2[i] += d[i] * (rr[i] - rmu[i]) * riv[i];
      }
    }
  }
  // block reduction without LDS atomics: shuffles inside a wave, plain LDS stores across waves
  const int nvec = ry ? 3 : 2;
  block_channel_sum(s0, sm, G, C, nvec, 0);
  block_channel_sum(s1, sm, G, C, nvec, 1);
  if (ry) block_channel_sum(s2, sm, G, C, nvec, 2);
  __syncthreads();
  const int lim = nvec * C;
  for (int i = threadIdx.x; i < lim; i += 256) {
    const int vv = i / C, c = i % C;
    float t = 0.f;
    for (int w = 0; w < 4; ++w) t += sm[(vv * 4 + w) * C + c];
    // 16 slots: with one, the 1024 blocks of the layer-1 map each ended on the same 128 addresses, and same-address float
    // atomics serialise in L2 at ~25 ns apiece -- a 25-us tail on a 45-us pass
    // deterministic mode: block x plain-stores partial x; det_fold_kernel adds the partials in index order into slot 0
    if (det_part) det_part[vv * ldc + c] = t;
    else atomicAdd(&scratch[(blockIdx.x % MSCL_STAT_ACTIVE) * 4 * ldc + vv * ldc + c], t);
  }
}

// ---- backward pass 2: dy = gamma*inv*(dz - sum_dz/n - xhat*sum_dzxhat/n); residual gradient ----
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, const bf16_t* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ inv,
    const bf16_t* __restrict__ ry, const float* __restrict__ rgamma, const float* __restrict__ rmean,
    const float* __restrict__ rinv, const float* __restrict__ scratch, bf16_t* __restrict__ dy,
    bf16_t* __restrict__ dres, int identity_dres, long rows, int C, int relu,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ rdgamma, float* __restrict__ rdbeta,
    const float* __restrict__ beta, int groups, int ldc, int nslots) {
  extern __shared__ float sm[];   // [10][groups][C]: gamma*inv, mean, inv, a, b; the residual's four; mask shift
  // maps wider than 512 channels run as 512-channel chunks along blockIdx.y (C = chunk width, ldc = the map's channel count): the
  // constants of a 2048-channel map took 80 KB of LDS = one block per CU, and the pass ran at 1.8 TB/s
  const int ch = blockIdx.y * C;
  gamma += ch; mean += ch; inv += ch; scratch += ch; dgamma += ch; dbeta += ch;
  if (beta) beta += ch;
  if (ry) { rgamma += ch; rmean += ch; rinv += ch; rdgamma += ch; rdbeta += ch; }
  const int GC = groups * C;
  float* gi = sm; float* mu = sm + GC; float* iv = sm + 2 * GC; float* ca = sm + 3 * GC; float* cb = sm + 4 * GC;
  float* rgi = sm + 5 * GC; float* rmu = sm + 6 * GC; float* riv = sm + 7 * GC; float* rcb = sm + 8 * GC;
  float* msh = sm + 9 * GC;       // mask from y: bn(y) = y * gi + msh  (see the reduce pass)
  const bool mask_y = relu && beta != nullptr;
  const long rows_g = rows / groups;         // statistics groups (<= 2): rows below rows_g are group 0
  const float inv_n = 1.f / (float)rows_g;
  const int G = C >> 3;                      // a power of two (checked by the launcher)
  const int gsh = 31 - __clz(G);
  const unsigned total32 = (unsigned)(rows * G), step32 = gridDim.x * blockDim.x, gmask = (unsigned)(G - 1);
  const unsigned ebound = groups > 1 ? (unsigned)(rows_g * G) : 0xFFFFFFFFu;
  const long ldg = ldc >> 3, chg = ch >> 3;  // granules per map row, first granule of this chunk
  // UNR granules per trip, every load of a trip in flight before the first use (one granule per trip ran the pass at 3.0 TB/s
  // inside the step against 5-6 for the forward pass); the first trip's loads go out before the per-channel constants are
  // rebuilt from the scratch slots, as in the forward pass
  constexpr int UNR = 4;
  const bool need_out = relu && !mask_y;
  uint4 vd[UNR], vy[UNR], va[UNR], vr[UNR];
  auto gaddr = [&](unsigned e32) { return ((long)(e32 >> gsh) * ldg + chg + (e32 & gmask)) * 8; };
  auto load_trip = [&](unsigned e0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned e32 = e0 + u * step32;
      const long o = gaddr(e32 < total32 ? e32 : (e0 < total32 ? e0 : 0u));
      vd[u] = *reinterpret_cast<const uint4*>(dout + o);
      vy[u] = *reinterpret_cast<const uint4*>(y + o);
      if (need_out) va[u] = *reinterpret_cast<const uint4*>(out + o);
      if (ry) vr[u] = *reinterpret_cast<const uint4*>(ry + o);
    }
  };
  unsigned e0 = blockIdx.x * blockDim.x + threadIdx.x;
  load_trip(e0);
  // Constants of this chunk's (at most 512) channels, two per thread, every load -- the reduce pass's slots, gamma, mean, invstd, beta,
  // the residual's three -- issued before the first use (as a loop over the slots this was 4 + 1 dependent round trips per channel
  // on top of the first data trip: half of what a small map's pass took; see bn_prepare)
  {
    float t[2][2][3][MSCL_STAT_ACTIVE], gm[2], bt[2], rg[2], mn[2][2], vi[2][2], rmn[2][2], rvi[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = (int)threadIdx.x + k * 256 < C ? (int)threadIdx.x + k * 256 : (int)threadIdx.x % C;
      gm[k] = gamma[c]; bt[k] = mask_y ? beta[c] : 0.f; rg[k] = ry ? rgamma[c] : 0.f;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        if (gq >= groups) break;
        const int kg = gq * ldc + c;
        const float* sc = scratch + (long)gq * MSCL_STAT_SLOTS * 4 * ldc;
        mn[k][gq] = mean[kg]; vi[k][gq] = inv[kg];
        rmn[k][gq] = ry ? rmean[kg] : 0.f; rvi[k][gq] = ry ? rinv[kg] : 0.f;
#pragma unroll
        for (int sl = 0; sl < MSCL_STAT_ACTIVE; ++sl) {
          const int si = sl < nslots ? sl : 0;             // a valid slot either way; what lies beyond nslots is dropped below
          t[k][gq][0][sl] = sc[si * 4 * ldc + c]; t[k][gq][1][sl] = sc[si * 4 * ldc + ldc + c];
          t[k][gq][2][sl] = ry ? sc[si * 4 * ldc + 2 * ldc + c] : 0.f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = (int)threadIdx.x + k * 256;
      if (c >= C) break;
      float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        if (gq >= groups) break;
        const int kk = gq * C + c;                           // LDS index
        gi[kk] = gm[k] * vi[k][gq]; mu[kk] = mn[k][gq]; iv[kk] = vi[k][gq];
        msh[kk] = mask_y ? bt[k] - mn[k][gq] * (gm[k] * vi[k][gq]) : 0.f;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int sl = 0; sl < MSCL_STAT_ACTIVE; ++sl) if (sl < nslots) { t0 += t[k][gq][0][sl]; t1 += t[k][gq][1][sl]; t2 += t[k][gq][2][sl]; }
        ca[kk] = t0 * inv_n; cb[kk] = t1 * inv_n;
        if (ry) { rgi[kk] = rg[k] * rvi[k][gq]; rmu[kk] = rmn[k][gq]; riv[kk] = rvi[k][gq]; rcb[kk] = t2 * inv_n; }
        p0 += t0; p1 += t1; p2 += t2;
      }
      if (blockIdx.x == 0) {      // parameter gradients (+=: the flow trunk is traversed twice per step, or once with two groups)
        atomicAdd(&dgamma[c], p1); atomicAdd(&dbeta[c], p0);
        if (ry) { atomicAdd(&rdgamma[c], p2); atomicAdd(&rdbeta[c], p0); }
      }
    }
  }
  __syncthreads();
  for (; e0 < total32; e0 += step32 * UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const unsigned e32 = e0 + u * step32;
      if (e32 >= total32) break;
      const long o = gaddr(e32);
      const int c0 = (int)(e32 & gmask) * 8 + (e32 >= ebound ? C : 0);
      float d[8], yy[8], o8[8];
      unpack8(vd[u], d);
      unpack8(vy[u], yy);
      if (mask_y) {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = (yy[i] * gi[c0 + i] + msh[c0 + i]) > 0.f ? d[i] : 0.f;
      } else if (relu) {
        float a[8]; unpack8(va[u], a);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = a[i] > 0.f ? d[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i;
        o8[i] = gi[c] * (d[i] - ca[c] - (yy[i] - mu[c]) * iv[c] * cb[c]);
      }
      *reinterpret_cast<uint4*>(dy + o) = pack8(o8);
      if (ry) {
        float rr[8]; unpack8(vr[u], rr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = c0 + i;
          o8[i] = rgi[c] * (d[i] - ca[c] - (rr[i] - rmu[c]) * riv[c] * rcb[c]);
        }
        *reinterpret_cast<uint4*>(dres + o) = pack8(o8);
      } else if (identity_dres) {
        *reinterpret_cast<uint4*>(dres + o) = pack8(d);
      }
    }
    if (e0 + step32 * UNR < total32) load_trip(e0 + step32 * UNR);
  }
}

extern "C" int mscl_bn_act_bwd(const uint16_t* dout, const uint16_t* out, const uint16_t* y, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                               const uint16_t* res_y, const float* res_gamma, const float* res_mean,
                               const float* res_invstd, float* res_dgamma, float* res_dbeta, uint16_t* dy,
                               uint16_t* dres, int want_identity_dres, float* scratch, int64_t rows, int C, int relu,
                               void* stream) {
  return mscl_bn_act_bwd_groups(dout, out, y, gamma, beta, save_mean, save_invstd, dgamma, dbeta, res_y, res_gamma, res_mean,
                                res_invstd, res_dgamma, res_dbeta, dy, dres, want_identity_dres, scratch, rows, C, relu, 1, nullptr, 0,
                                stream);
}

extern "C" int mscl_bn_act_bwd_groups(const uint16_t* dout, const uint16_t* out, const uint16_t* y, const float* gamma,
                                      const float* beta, const float* save_mean, const float* save_invstd, float* dgamma,
                                      float* dbeta, const uint16_t* res_y, const float* res_gamma, const float* res_mean,
                                      const float* res_invstd, float* res_dgamma, float* res_dbeta, uint16_t* dy,
                                      uint16_t* dres, int want_identity_dres, float* scratch, int64_t rows, int C, int relu,
                                      int groups, float* det_parts, int64_t det_parts_floats, void* stream) {
  if (!dout || !y || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !dy || !scratch) return MSCL_E_ARG;
  if (groups < 1 || groups > 2 || rows % groups != 0) return MSCL_E_SHAPE;
  if (relu != 0 && relu != 1) return MSCL_E_ARG;
  if (beta && (res_y || want_identity_dres || !relu)) return MSCL_E_ARG;     // the mask depends on the residual too: pass `out`
  if (relu && !out && !beta) return MSCL_E_ARG;
  if (rows <= 0 || C <= 0) return MSCL_E_ARG;
  if (C % 8 || ilog2_exact(C / 8) < 0 || C > 2048 || rows * (C / 8) >= (1LL << 31)) return MSCL_E_SHAPE;     // block_channel_sum needs C/8 <= 64: wider maps run in 512-channel chunks
  if (res_y && (!res_gamma || !res_mean || !res_invstd || !res_dgamma || !res_dbeta || !dres)) return MSCL_E_ARG;
  if (want_identity_dres && !dres) return MSCL_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Cc = C > 512 ? 512 : C, chunks = C / Cc;
  const int RP = 256 / (Cc / 8);
  const long rows_g = rows / groups;
  long blocks = (rows_g + RP * 8 - 1) / (RP * 8);
  constexpr long red_cap = 512;                // see mscl_bn_act_fwd_groups
  const long cap = red_cap / (chunks * groups);                    // wider grids measured slower (more atomics)
  if (blocks > cap) blocks = cap; if (blocks < 1) blocks = 1;
  // deterministic mode: partial x of block x (in the caller's `det_parts`, or in the slots themselves: 16 blocks), folded into slot 0
  const bool det = mscl_det();
  const bool own = det && det_parts != nullptr && det_parts_floats >= mscl_det_parts_floats(rows, C, groups, 4);
  if (det) blocks = own ? det_parts_of(rows_g, C) : (blocks > MSCL_STAT_SLOTS ? MSCL_STAT_SLOTS : blocks);
  {
    float* part = det ? (own ? det_parts : scratch) : nullptr;
    const long pgs = (own ? blocks : (long)MSCL_STAT_SLOTS) * 4 * C;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)blocks, chunks, groups), dim3(256), (size_t)12 * Cc * sizeof(float), st, dout,
                       out, y, save_mean, save_invstd, res_y, res_mean, res_invstd, scratch, (long)rows_g, Cc, relu, gamma, beta, C, part, pgs);
    MSCL_LAUNCH_CHECK();
    if (det) {
      hipLaunchKernelGGL(det_fold_kernel, dim3((3 * C + 31) / 32, groups), dim3(256), 0, st, (const float*)part, scratch, (int)blocks,
                         res_y ? 3 : 2, 4, C, pgs, (long)MSCL_STAT_SLOTS * 4 * C, own ? 0 : 1);
      MSCL_LAUNCH_CHECK();
    }
  }
  int Ca = Cc, achunks = chunks;                                 // the apply pass's own chunking
  if (Ca * groups > 1024) {      // 10 * C floats of constants per group: past the 64-KB default for dynamic LDS
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn_bwd_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  }
  const long total = rows * (Ca / 8);
  constexpr long app_cap = 512;                // see mscl_bn_act_fwd_groups
  constexpr long gpt = 1;                      // granules per thread (swept 1 / 2 / 4 inside the step: no gain)
  long b2 = (total + 256 * gpt - 1) / (256 * gpt); if (b2 > app_cap / achunks) b2 = app_cap / achunks; if (b2 < 1) b2 = 1;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)b2, achunks), dim3(256), (size_t)10 * groups * Ca * sizeof(float), st, dout, out, y,
                     gamma, save_mean, save_invstd, res_y, res_gamma, res_mean, res_invstd, scratch, dy, dres,
                     want_identity_dres, (long)rows, Ca, relu, dgamma, dbeta, res_dgamma, res_dbeta, beta, groups, C,
                     mscl_stat_nslots());
  MSCL_LAUNCH_CHECK();
  return 0;
}
