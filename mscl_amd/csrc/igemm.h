// Shared by the implicit-GEMM conv kernels (conv_igemm.hip, conv_pp.hip): geometry, descriptor helper, LDS swizzle, epilogue.
#pragma once
#include "common.h"

struct ClassInfo { unsigned char ro[3]; unsigned char ntl; unsigned char taps[8]; int TrS, HrS, WrS, M; FastDiv dW, dH, dT; };

struct IGemmGeom {
  int N, Ts, Hs, Ws, Cs;   // source (gathered) tensor
  int Tr, Hr, Wr, Cr;      // row tensor
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int M, KG, cgs, ntaps;
  int lsT, lsH, lsW;
  int mode;                // 0 forward gather, 1 dgrad stride-1 (linear), 2 dgrad strided (parity classes)
  int mtiles, ntiles, ksplit, nclass;
  int grp_rows;            // BatchNorm statistics groups (forward only): rows [k*grp_rows, (k+1)*grp_rows) feed group k; 0 = one group
  FastDiv dW, dH, dT;      // dense launches: division by Wr, Hr, Tr
  FastDiv dKW, dKH;        // tap index -> (kt, kh, kw) (uniform-tap kernel)
  ClassInfo cls[8];
};

__device__ __forceinline__ auto make_uniform_rsrc(const void* p, unsigned bytes) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
  void* q = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// host-side entry points shared between the translation units
#define MSCL_PP_SKIP (-100)
int mscl_launch_splitk_finalize(const float* partial, bf16_t* out, const float* bias, const bf16_t* addend, int relu, float* ssum,
                                float* ssq, long rows, int C, int nslab, long grp_rows, hipStream_t st);
int mscl_launch_conv_pp(IGemmGeom g, const bf16_t* src, const bf16_t* wgt, bf16_t* out, const float* bias, const bf16_t* addend,
                        float* ssum, float* ssq, int relu, float* ws, long ws_floats, hipStream_t st);

template <int BK> __device__ __forceinline__ int swz(int row, int g) {
  if constexpr (BK == 64) return g ^ ((row >> 1) & 7);
  else { const int q = (row >> 2) & 3; return g ^ ((0x78 >> (q * 2)) & 3); }   // q -> {0,2,3,1}
}

// Shared epilogue: split-K slab store, or BatchNorm statistics + (+bias)(+addend)(relu) -> bf16.
// orow[i] = element offset of the output row of this lane's position fragment i (channel 0), or -1 for a row that is not stored.
// MASKROWS: rows with orow < 0 may hold non-zero accumulators (the padding columns of the shared-tap kernel, conv_pp.hip) and are
// kept out of the statistics; the kernels that zero-fill whole A rows beyond M do not need the test.
template <int BM, int BN, int IM, int JN, bool MASKROWS>
__device__ __forceinline__ void igemm_epilogue_rows(const IGemmGeom& g, f32x4_t (&acc)[JN][IM], unsigned char* smem, int tid, int fr, int fq,
                                                    int m0, int n0, int wm0, int wn0, int split, int Mc, const long (&orow)[IM],
                                                    bf16_t* __restrict__ out, const float* __restrict__ bias,
                                                    const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                                    float* __restrict__ stat_sq, int relu, float* __restrict__ partial) {
  if (partial != nullptr) {                   // split-K: this split's fp32 slab (plain 16-byte stores); the
    float* slab = partial + (long)split * ((long)g.N * g.Tr * g.Hr * g.Wr * g.Cr);   // epilogue runs in splitk_finalize_kernel
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      if (orow[i] < 0) continue;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int n = n0 + wn0 + j * 16 + fq * 4;
        if (n >= g.Cr) continue;
        *reinterpret_cast<float4*>(slab + orow[i] + n) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
      }
    }
    return;
  }

  // ---- epilogue: BatchNorm statistics of the raw fp32 result ----
  if (stat_sum != nullptr) {
    float* red = reinterpret_cast<float*>(smem);      // [waves][2][JN * 16]
    // statistics groups (a batch that holds two BatchNorm calls of the reference, e.g. base || rotated flow clips): rows below
    // `bound` feed group g_lo, the rest (a tile that straddles the boundary; there are at most two groups) group g_lo + 1
    int g_lo = 0, bound = 0x7fffffff, npass = 1;
    if (g.grp_rows > 0) {
      g_lo = m0 / g.grp_rows;
      bound = (g_lo + 1) * g.grp_rows;
      npass = (min(m0 + BM, Mc) > bound) ? 2 : 1;
    }
    // Every wave plain-stores the sums of its JN * 16 channels as one LDS row ([wave row][wave column][2][JN * 16]); the block's
    // first BN threads add the rows of their column (no LDS atomics, no zero-fill pass; cost of this section: conv_halo.hip).
    constexpr int WCH = JN * 16, NWM = BM / (IM * 16), NWN = BN / WCH;
    const int wrow = (wm0 / (IM * 16)) * NWN + wn0 / WCH;
    for (int pass = 0; pass < npass; ++pass) {
      __syncthreads();                                // the tiles (or the previous pass's rows) are dead
      // value 8 j + 4 sq + r = (channel tile j, sum / sum of squares, channel r of this lane row's quad); the reduce-scatter
      // (common.h) leaves quad q of a lane row with the totals of values [q * 2 JN, (q + 1) * 2 JN): half the DPP moves of one
      // row sum per value, and the LDS stores spread over four lanes of a row
      static_assert(JN == 1 || JN % 2 == 0, "a quad's share must be whole (j, sq) groups");
      float sv[JN * 8];
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < IM; ++i) {
          const bool mine = (((m0 + wm0 + i * 16 + fr) < bound) == (pass == 0)) && (!MASKROWS || orow[i] >= 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = mine ? acc[j][i][r] : 0.f; s[r] += v; q[r] += v * v; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv[j * 8 + r] = s[r]; sv[j * 8 + 4 + r] = q[r]; }
      }
      row16_reduce_scatter<JN * 8>(sv);
      if ((fr & 3) == 0) {
        const int v0 = (fr >> 2) * (2 * JN);           // first value of this quad
#pragma unroll
        for (int k = 0; k < 2 * JN; k += 4) {          // groups of 4 = the r of one (j, sq); JN = 1: two values (half a group) per quad
          if constexpr (JN >= 2) {
            const int v = v0 + k, j = v >> 3, sq = (v >> 2) & 1;
            *reinterpret_cast<float4*>(&red[(wrow * 2 + sq) * WCH + j * 16 + fq * 4]) = make_float4(sv[k], sv[k + 1], sv[k + 2], sv[k + 3]);
          } else {
            const int sq = v0 >> 2, r0 = v0 & 3;
            *reinterpret_cast<float2*>(&red[(wrow * 2 + sq) * WCH + fq * 4 + r0]) = make_float2(sv[0], sv[1]);
          }
        }
      }
      __syncthreads();
      for (int i = tid; i < BN; i += (int)blockDim.x) {
        if (n0 + i < g.Cr) {
          const int wn = i / WCH, c = i - wn * WCH;
          float ts = 0.f, tq = 0.f;
#pragma unroll
          for (int m = 0; m < NWM; ++m) { ts += red[((m * NWN + wn) * 2 + 0) * WCH + c]; tq += red[((m * NWN + wn) * 2 + 1) * WCH + c]; }
          const int so = ((g_lo + pass) * MSCL_STAT_SLOTS + (int)(blockIdx.x % MSCL_STAT_ACTIVE)) * 2 * g.Cr;
          atomicAdd(&stat_sum[so + n0 + i], ts); atomicAdd(&stat_sq[so + n0 + i], tq);
        }
      }
    }
  }

  // ---- epilogue: (+bias) (+addend) (relu) -> bf16 ----
  // A lane holds 4 consecutive channels of one position per 16-channel tile: 8 bytes.  Stores of 8 bytes per lane are issue-bound
  // on write-heavy layers (1-tap convs that widen the map, e.g. 64 -> 256 on a 205-MB map: 2.3 TB/s against 4.4-5.0 TB/s for the
  // read-heavy direction), so two channel tiles are paired: v_permlane16_swap exchanges the quads between lane rows fq and fq ^ 1
  // (same position, neighbouring channel quads), after which an even row holds 8 consecutive channels of tile j and an odd row
  // 8 consecutive channels of tile j + 1 -- one 16-byte store per lane instead of two 8-byte ones, same bytes, same values.
  auto quad = [&](int i, int j, bool ok) -> uint2 {
    const int n = n0 + wn0 + j * 16 + fq * 4;
    float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
    if (ok && n < g.Cr) {
      if (bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (addend != nullptr) {
        const uint2 av = *reinterpret_cast<const uint2*>(addend + orow[i] + n);
        v[0] += __uint_as_float(av.x << 16); v[1] += __uint_as_float(av.x & 0xFFFF0000u);
        v[2] += __uint_as_float(av.y << 16); v[3] += __uint_as_float(av.y & 0xFFFF0000u);
      }
    }
    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    uint2 pv; pv.x = pack2bf(v[0], v[1]); pv.y = pack2bf(v[2], v[3]);
    return pv;
  };
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const bool rowok = orow[i] >= 0;
    if constexpr (JN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < JN; j += 2) {
        const uint2 p0 = quad(i, j, rowok), p1 = quad(i, j + 1, rowok);
        // every lane takes part in the swaps (EXEC full here: no divergent branch encloses them)
        const auto sx = __builtin_amdgcn_permlane16_swap(p0.x, p1.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(p0.y, p1.y, false, false);
        // even rows: {own tile-j quad, next row's tile-j quad}; odd rows: {previous row's tile-(j+1) quad, own tile-(j+1) quad}
        const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        const int n = n0 + wn0 + (j + (fq & 1)) * 16 + (fq & 2) * 4;
        if (rowok && n < g.Cr) *reinterpret_cast<uint4*>(out + orow[i] + n) = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int n = n0 + wn0 + j * 16 + fq * 4;
        const uint2 pv = quad(i, j, rowok);
        if (rowok && n < g.Cr) *reinterpret_cast<uint2*>(out + orow[i] + n) = pv;
      }
    }
  }
}

template <int BM, int BN, int IM, int JN>
__device__ __forceinline__ void igemm_epilogue(const IGemmGeom& g, f32x4_t (&acc)[JN][IM], unsigned char* smem, int tid, int fr, int fq,
                                               int m0, int n0, int wm0, int wn0, int split, int Mc, FastDiv dW, FastDiv dH, FastDiv dT,
                                               int TrS, int HrS, int WrS, int rsT, int rsH, int rsW, int roT, int roH, int roW,
                                               bf16_t* __restrict__ out, const float* __restrict__ bias,
                                               const bf16_t* __restrict__ addend, float* __restrict__ stat_sum,
                                               float* __restrict__ stat_sq, int relu, float* __restrict__ partial) {
  // output position of this lane's rows (class-strided for the parity-split input gradient)
  long orow[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int m = m0 + wm0 + i * 16 + fr;
    if (m < Mc) {
      const int q1 = fdiv(m, dW), ws_ = m - q1 * WrS;
      const int q2 = fdiv(q1, dH), hs_ = q1 - q2 * HrS;
      const int n = fdiv(q2, dT), ts_ = q2 - n * TrS;
      orow[i] = ((((long)n * g.Tr + ts_ * rsT + roT) * g.Hr + hs_ * rsH + roH) * g.Wr + ws_ * rsW + roW) * g.Cr;
    } else orow[i] = -1;
  }
  igemm_epilogue_rows<BM, BN, IM, JN, false>(g, acc, smem, tid, fr, fq, m0, n0, wm0, wn0, split, Mc, orow, out, bias, addend, stat_sum,
                                             stat_sq, relu, partial);
}
