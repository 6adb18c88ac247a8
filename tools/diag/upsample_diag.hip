// Diagnostic twin of upsample_add_kernel (mscl_amd/csrc/elementwise.hip), trilinear path only: test infrastructure, never linked
// into libmscl_hip.so.  Built by tools/diag/build.sh into tools/diag/libups_diag.so; tools/flake_det.py (FLAKE_DIAG=1) and
// tools/diag/flake_repro.cpp route the step's / the reproducer's up-sampling launches through it.
//
// What it is for (profiles/r05_flake_det.md, r06_flake.md): inside the three-stream step one output row of this kernel came out as
// "right value minus ONE corner's contribution" for the sixteen lanes 48-63 of a wave, about once in 4000 launches.  Three things
// can produce that -- the corner's WEIGHT (VALU), its OFFSET (VALU) or the LOAD -- and the round-5 probes could not tell them apart
// (a NaN-preset destination with a valid load and a zero weight also gives a finite row).  Here every stage is done TWICE from
// inputs the compiler cannot see to be equal, with different instructions where there is a choice, and compared per lane:
//   index decode -> offsets off1 / off2 and weights wt1 / wt2;
//   loads A = global_load_dwordx4 (the form of the original kernel), loads B = buffer_load_dwordx4 ... sc1 (past L1);
//   accumulation fA (wt1, A) and fB (wt2, B).
// Any disagreement, and any corner that arrives as sixteen zero bytes, is written to a device record with the hardware ids of
// the wave (HW_ID, XCC_ID), the values of both sides and a third load C issued after a pause.  The OUTPUT is the A side's, as the
// original kernel would have produced it, so that the step-level comparison of tools/flake_det.py still sees the event.
#include "../../mscl_amd/csrc/common.h"

#define UPS_REC_WORDS 48
#define UPS_MAX_REC 2048
__device__ unsigned g_ups_cnt[8];                    // [0] records taken, [1] launches, [2] elements processed (low word)
__device__ unsigned g_ups_rec[UPS_MAX_REC * UPS_REC_WORDS];

__device__ __forceinline__ void lin_coord_d(int d, int in, int outn, int& i0, int& i1, float& w1) {
  const float sc = (float)in / (float)outn;
  float s = ((float)d + 0.5f) * sc - 0.5f; s = s < 0.f ? 0.f : s;
  i0 = (int)s; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 > in - 1 ? in - 1 : i0 + 1; w1 = s - (float)i0;
}
struct UpDivD { FastDiv G, Wd, Hd, Td; };

__device__ __forceinline__ void decode(int e, int G, int Ts, int Hs, int Ws, int Td, int Hd, int Wd, int C, const UpDivD& dv,
                                       int* off, float* wt) {
  int r = fdiv(e, dv.G); const int gq = e - r * G;
  int q = fdiv(r, dv.Wd); const int w = r - q * Wd; r = q;
  q = fdiv(r, dv.Hd); const int h = r - q * Hd; r = q;
  const int n = fdiv(r, dv.Td), t = r - n * Td;
  int t0, t1, h0, h1, w0, w1; float a, b, c;
  lin_coord_d(t, Ts, Td, t0, t1, a); lin_coord_d(h, Hs, Hd, h0, h1, b); lin_coord_d(w, Ws, Wd, w0, w1, c);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int tt = (k & 4) ? t1 : t0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
    wt[k] = ((k & 4) ? a : 1.f - a) * ((k & 2) ? b : 1.f - b) * ((k & 1) ? c : 1.f - c);
    off[k] = (((((n * Ts + tt) * Hs + hh) * Ws + ww) * C) + gq * 8) * 2;          // byte offset, < 2^31 (launcher)
  }
}

__global__ __launch_bounds__(256) void upsample_diag_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N,
                                                            int Ts, int Hs, int Ws, int Td, int Hd, int Wd, int C,
                                                            int accumulate, UpDivD dv, int use_b) {
  const int G = C >> 3;
  const int total = N * Td * Hd * Wd * G;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_ups_cnt[1], 1u);
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const uint64_t sa = reinterpret_cast<uint64_t>(src);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sa), hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int off1[8], off2[8]; float wt1[8], wt2[8];
    uint4 vA[8]; u32x4_t vB[8];
    float fA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fB[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
      // side A: the ORIGINAL kernel's statement sequence, literally (index decode, a global load per corner inside the loop with 64-bit
      // address arithmetic between the loads, the weight formed next to its load), keeping what it loaded and multiplied by
      int r = fdiv(e, dv.G); const int gq = e - r * G;
      int q = fdiv(r, dv.Wd); const int w = r - q * Wd; r = q;
      q = fdiv(r, dv.Hd); const int h = r - q * Hd; r = q;
      const int n = fdiv(r, dv.Td), t = r - n * Td;
      int t0, t1, h0, h1, w0, w1; float a, b, c;
      lin_coord_d(t, Ts, Td, t0, t1, a); lin_coord_d(h, Hs, Hd, h0, h1, b); lin_coord_d(w, Ws, Wd, w0, w1, c);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tt = (k & 4) ? t1 : t0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
        const float wt = ((k & 4) ? a : 1.f - a) * ((k & 2) ? b : 1.f - b) * ((k & 1) ? c : 1.f - c);
        const bf16_t* ap = src + ((((long)n * Ts + tt) * Hs + hh) * Ws + ww) * C + gq * 8;
        float g8[8];
        vA[k] = *reinterpret_cast<const uint4*>(ap);
        unpack8(vA[k], g8);
#pragma unroll
        for (int i = 0; i < 8; ++i) fA[i] += wt * g8[i];
        wt1[k] = wt; off1[k] = (int)((ap - src) * 2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int e2 = e; asm volatile("" : "+v"(e2));               // the compiler may not merge side B's decode with side A's
    decode(e2, G, Ts, Hs, Ws, Td, Hd, Wd, C, dv, off2, wt2);
#pragma unroll
    for (int k = 0; k < 8; ++k) vB[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off2[k], 0, 16 /* sc1 (gfx940+: aux bit 4) */);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float g8[8]; unpack8(make_uint4(vB[k][0], vB[k][1], vB[k][2], vB[k][3]), g8);
#pragma unroll
      for (int i = 0; i < 8; ++i) fB[i] += wt2[k] * g8[i];
    }
    unsigned m_off = 0, m_wt = 0, m_ld = 0, m_zA = 0, m_zB = 0, m_res = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      m_off |= (off1[k] != off2[k]) << k;
      m_wt |= (__float_as_uint(wt1[k]) != __float_as_uint(wt2[k])) << k;
      m_ld |= (vA[k].x != vB[k][0] || vA[k].y != vB[k][1] || vA[k].z != vB[k][2] || vA[k].w != vB[k][3]) << k;
      m_zA |= ((vA[k].x | vA[k].y | vA[k].z | vA[k].w) == 0u) << k;
      m_zB |= ((vB[k][0] | vB[k][1] | vB[k][2] | vB[k][3]) == 0u) << k;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) m_res |= (__float_as_uint(fA[i]) != __float_as_uint(fB[i])) << i;
    if (m_off | m_wt | m_ld | m_zA | m_zB | m_res) {
      const unsigned slot = atomicAdd(&g_ups_cnt[0], 1u);
      if (slot < UPS_MAX_REC) {
        unsigned* R = g_ups_rec + slot * UPS_REC_WORDS;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned long long tm = __builtin_amdgcn_s_memtime();
        const unsigned bad = m_off | m_wt | m_ld | m_zA | m_zB;
        const int k = bad ? __builtin_ctz(bad) : 0;
        __builtin_amdgcn_s_sleep(64);
        u32x4_t vC = __builtin_amdgcn_raw_buffer_load_b128(rs, off2[0], 0, 17);    // (re-indexed below: k is a run-time value)
        int offk1 = 0, offk2 = 0; unsigned w1b = 0, w2b = 0; uint4 a = make_uint4(0, 0, 0, 0); u32x4_t b = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == k) { offk1 = off1[j]; offk2 = off2[j]; w1b = __float_as_uint(wt1[j]); w2b = __float_as_uint(wt2[j]); a = vA[j]; b = vB[j]; }
        vC = __builtin_amdgcn_raw_buffer_load_b128(rs, offk2, 0, 17);      // sc0 sc1
        R[0] = m_off | (m_wt << 8) | (m_ld << 16) | (m_zA << 24);
        R[1] = m_zB | (m_res << 8);
        R[2] = g_ups_cnt[1]; R[3] = blockIdx.x; R[4] = threadIdx.x; R[5] = (unsigned)e; R[6] = hwid; R[7] = xcc;
        R[8] = (unsigned)tm; R[9] = (unsigned)(tm >> 32); R[10] = (unsigned)k; R[11] = (unsigned)offk1; R[12] = (unsigned)offk2;
        R[13] = w1b; R[14] = w2b;
        R[15] = a.x; R[16] = a.y; R[17] = a.z; R[18] = a.w;
        R[19] = b[0]; R[20] = b[1]; R[21] = b[2]; R[22] = b[3];
        R[23] = vC[0]; R[24] = vC[1]; R[25] = vC[2]; R[26] = vC[3];
        R[27] = __float_as_uint(fA[0]); R[28] = __float_as_uint(fB[0]);
        R[29] = (unsigned)total; R[30] = (unsigned)(Td | (Hd << 8) | (Wd << 16)); R[31] = (unsigned)C;
#pragma unroll
        for (int j = 0; j < 8; ++j) { R[32 + j] = __float_as_uint(wt1[j]); R[40 + j] = __float_as_uint(wt2[j]); }
      }
    }
    float* f = use_b ? fB : fA;
    if (accumulate) {
      float d8[8]; unpack8(*reinterpret_cast<const uint4*>(dst + (long)e * 8), d8);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += d8[i];
    }
    *reinterpret_cast<uint4*>(dst + (long)e * 8) = pack8(f);
  }
}

static int g_use_b = 0;
extern "C" void ups_diag_use_b(int v) { g_use_b = v; }

// same signature as mscl_upsample_add (include/mscl_hip.h); trilinear launches only -- the caller keeps nearest launches on the product
extern "C" int ups_diag_upsample_add(const uint16_t* src, uint16_t* dst, int N, int Ts, int Hs, int Ws, int Td, int Hd, int Wd,
                                     int C, int trilinear, int accumulate, void* stream) {
  if (!src || !dst || !trilinear || C % 8) return -1;
  const long total = (long)N * Td * Hd * Wd * (C / 8);
  if (total >= (1L << 31) || (long)N * Ts * Hs * Ws * C * 2 >= (1L << 31)) return -2;
  long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
  UpDivD d; d.G = make_fastdiv(C / 8); d.Wd = make_fastdiv(Wd); d.Hd = make_fastdiv(Hd); d.Td = make_fastdiv(Td);
  hipLaunchKernelGGL(upsample_diag_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, N, Ts, Hs, Ws, Td, Hd, Wd,
                     C, accumulate, d, g_use_b);
  return (int)hipGetLastError();
}

// copies [counters (8 words)][records] to host memory; returns the number of records taken so far (synchronises the device)
extern "C" int ups_diag_read(unsigned* counters8, unsigned* records, int max_records) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  unsigned c[8];
  if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_ups_cnt), sizeof(c)) != hipSuccess) return -2;
  for (int i = 0; i < 8; ++i) counters8[i] = c[i];
  int n = (int)(c[0] < UPS_MAX_REC ? c[0] : UPS_MAX_REC); if (n > max_records) n = max_records;
  if (n > 0 && hipMemcpyFromSymbol(records, HIP_SYMBOL(g_ups_rec), (size_t)n * UPS_REC_WORDS * 4) != hipSuccess) return -3;
  return n;
}
extern "C" int ups_diag_reset(void) {
  unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ups_cnt), c, sizeof(c));
}
extern "C" int ups_diag_rec_words(void) { return UPS_REC_WORDS; }
