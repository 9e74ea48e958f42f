// bz_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the decode hot path.
//
// Design (DESIGN.md has the full story):
//   * Every linear layer at batch 1 is an HBM-bound GEMV.  INT4 group-quantised weights (AWQ/GPTQ) are repacked
//     once at load into [N/64 tiles][K/32 chunks][64 lanes][16 B] so that one wave-wide `global_load_dwordx4`
//     reads 1 KiB contiguous = 32 k-values for 64 output columns; lane == output column, so there is no
//     cross-lane reduction, and the activation slice is wave-uniform (broadcast reads from LDS).
//   * The dot products run on V_DOT4_I32_I8: the f16 activation slice is split, per group of 128, into three
//     int8 planes (x ~= sx * (65536*hi + 256*mid + lo)), i.e. 24-bit fixed point relative to the group maximum --
//     the integer path reproduces f32 arithmetic on f16/bf16 activations.  Low nibbles are used as stored
//     (q, 0..15), high nibbles are stored as (q-8) in two's complement so that `w & 0xF0F0F0F0` IS the signed
//     byte 16*(q-8): one V_AND per 4 weights, no shifts.  All group arithmetic is exact in int32.
//   * Split-K partial sums are added with 64-bit INTEGER atomics in 2^-32 fixed point: integer addition is
//     associative, so the result is bit-reproducible regardless of block scheduling.  The consumer kernel
//     converts and rounds in its prologue -- there are no separate reduce / norm / activation launches.
//   * Each GEMV's prologue rebuilds its activation slice from the previous kernel's output (residual add,
//     RMSNorm, SiLU*up, rounding to the activation dtype), issues its first weight loads BEFORE doing so, and
//     stages the group scales/zeros for its tile in LDS.
#include "bz_internal.h"
#include <math.h>

#include <hip/hip_ext.h>

// ---------------------------------------------------------------------------------------------------------
// launch helper: plain launch, or -- while a profile is being taken -- hipExtLaunchKernelGGL with start/stop events
// bound to the dispatch itself (pure kernel time on the launch stream, no inter-kernel gap)
// ---------------------------------------------------------------------------------------------------------
static thread_local BzTimingSink* g_sink = nullptr;
void bzk_set_timing_sink(BzTimingSink* s) { g_sink = s; }

#define BZ_LAUNCH(label, bytes, kernel, grid, block, smem, stream, ...)                                              \
  do {                                                                                                               \
    if (g_sink) {                                                                                                    \
      hipEvent_t e0__, e1__;                                                                                         \
      BZ_HIP(hipEventCreate(&e0__));                                                                                 \
      BZ_HIP(hipEventCreate(&e1__));                                                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, e0__, e1__, 0, __VA_ARGS__);                          \
      g_sink->recs.push_back(BzTimingRec{label, (double)(bytes), e0__, e1__});                                       \
    } else {                                                                                                         \
      hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                            \
    }                                                                                                                \
  } while (0)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// streamed-once weights: non-temporal loads (guide: nt on weights that one CU reads once)
__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ldnt(const float4* p) {
  const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_round(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x;  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float round_act(float x, int act) {
  if (act == BZ_F16) return __half2float(__float2half_rn(x));
  if (act == BZ_BF16) return bf16_round(x);
  return x;
}
// 2^-32 fixed point
__device__ __forceinline__ float fix2f(long long a) {
  // sign-magnitude: (float)hi + (float)lo*2^-32 on the two's-complement halves cancels catastrophically for small
  // negative values (hi = -1, lo ~ 2^32)
  const unsigned long long m = a < 0 ? (unsigned long long)(-a) : (unsigned long long)a;
  const float r = (float)(unsigned)(m >> 32) + (float)(unsigned)(m & 0xffffffffull) * 2.3283064365386963e-10f;
  return a < 0 ? -r : r;
}
__device__ __forceinline__ long long f2fix(float p) { return __float2ll_rn(p * 4294967296.0f); }
__device__ __forceinline__ float vsrc_get(const VSrc& s, int i, int act) {
  if (s.fix) return round_act(fix2f(((const long long*)s.p)[i]), act);
  return ((const float*)s.p)[i];
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
// deterministic block sum over 256 threads; red: LDS float[4]
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return t;
}

__device__ __forceinline__ void zero_duty(long long* zb, int zn) {
  if (zb)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < zn; i += gridDim.x * blockDim.x) zb[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// prologue: build the activation slice x[k0, k0+KR) as f32 in LDS
// ---------------------------------------------------------------------------------------------------------
// Simple form (loads inside): used by the ROWS kernels, whose slice is all of K.
// pos(i): LDS position of slice element i (identity, or the ROWS swizzle).
template <bool SWZ>
__device__ __forceinline__ int xs_pos(int i) {
  if (!SWZ) return i;
  return (i & ~511) + ((i & 4) << 6) + ((i & 511) >> 3 << 2) + (i & 3);
}

template <bool SWZ>
__device__ void build_x_simple(const Pro& p, int k0, int KR, float* xs, float* red, bool writer) {
  const int tid = threadIdx.x;
  if (p.mode == PRO_NORM) {
    const bool hasprev = p.src.p != nullptr;
    float ss = 0.f;
    for (int base = 0; base < p.H; base += 4096) {
      float4 hv[4];
      float pv[4][4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int i = base + j * 1024 + tid * 4;
        hv[j] = (i < p.H) ? *(const float4*)(p.h_in + i) : make_float4(0, 0, 0, 0);
        if (hasprev) {
#pragma unroll
          for (int e = 0; e < 4; e++) pv[j][e] = (i < p.H) ? vsrc_get(p.src, i + e, p.act) : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int i = base + j * 1024 + tid * 4;
        float v[4] = {hv[j].x, hv[j].y, hv[j].z, hv[j].w};
        if (hasprev) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = round_act(v[e] + pv[j][e], p.act);
        }
        if (i < p.H) {
          ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
          if (writer && p.h_out) *(float4*)(p.h_out + i) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    ss = block_sum256(ss, red);
    const float rs = 1.0f / sqrtf(ss / (float)p.H + p.eps);
    for (int base = 0; base < KR; base += 2048) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        int i = base + e * 256 + tid;
        if (i < KR) {
          int kk = p.perm ? p.perm[k0 + i] : (k0 + i);
          float v = p.h_in[kk];
          if (hasprev) v = round_act(v + vsrc_get(p.src, kk, p.act), p.act);
          xs[xs_pos<SWZ>(i)] = round_act(p.norm_w[kk] * round_act(v * rs, p.act), p.act);
        }
      }
    }
  } else {
    for (int base = 0; base < KR; base += 2048) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        int i = base + e * 256 + tid;
        if (i < KR) {
          int kk = p.perm ? p.perm[k0 + i] : (k0 + i);
          float v;
          if (p.mode == PRO_SILU) {
            float g = vsrc_get(p.src, kk, p.act), u = vsrc_get(p.src, p.H + kk, p.act);
            v = round_act(round_act(silu_f(g), p.act) * u, p.act);
          } else {
            v = vsrc_get(p.src, kk, p.act);
          }
          xs[xs_pos<SWZ>(i)] = v;
        }
      }
    }
  }
  __syncthreads();
}

// Split form for the int4 GEMV: `xload` only ISSUES the prologue's global loads into registers, then the caller
// issues its first weight loads, then `xfinish` computes.  vmcnt is in-order, so this order lets the HBM weight
// loads fly while the (L2-resident) prologue data is consumed.  MODE / FIX are compile-time so that every array
// below is statically indexed and stays in registers.
template <int FIX>
__device__ __forceinline__ float vget(const void* p, int i, int act) {
  if (FIX) return round_act(fix2f(((const long long*)p)[i]), act);
  return ((const float*)p)[i];
}

template <int MODE, int MAXJ, int E>
struct XRegs {
  float4 h[MODE == PRO_NORM ? MAXJ : 1];     // NORM: full-H pass, element i = j*1024 + tid*4
  float pf[MODE == PRO_NORM ? MAXJ : 1][4];  // NORM: prev as f32 (converted at load)
  float sa[E];                               // slice element e: NORM h / PLAIN x / SILU gate
  float sb[MODE == PRO_PLAIN ? 1 : E];       // NORM prev / SILU up
  float sw[MODE == PRO_NORM ? E : 1];        // NORM weight
};

template <int MODE, int FIX, int MAXJ, int E>
__device__ __forceinline__ void xload(const Pro& p, int k0, int KR, XRegs<MODE, MAXJ, E>& r) {
  const int tid = threadIdx.x;
  const bool hasprev = p.src.p != nullptr;
  if (MODE == PRO_NORM) {
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
      const int i = j * 1024 + tid * 4;
      const bool on = i < p.H;
      r.h[j] = on ? *(const float4*)(p.h_in + i) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; e++) r.pf[j][e] = (on && hasprev) ? vget<FIX>(p.src.p, i + e, p.act) : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = e * 256 + tid;
    const bool on = i < KR;
    const int kk = on ? (p.perm ? p.perm[k0 + i] : (k0 + i)) : 0;
    if (MODE == PRO_NORM) {
      r.sa[e] = on ? p.h_in[kk] : 0.f;
      r.sb[e] = (on && hasprev) ? vget<FIX>(p.src.p, kk, p.act) : 0.f;
      r.sw[e] = on ? p.norm_w[kk] : 0.f;
    } else if (MODE == PRO_SILU) {
      r.sa[e] = on ? vget<FIX>(p.src.p, kk, p.act) : 0.f;
      r.sb[e] = on ? vget<FIX>(p.src.p, p.H + kk, p.act) : 0.f;
    } else {
      r.sa[e] = on ? vget<FIX>(p.src.p, kk, p.act) : 0.f;
    }
  }
}

template <int MODE, int MAXJ, int E>
__device__ __forceinline__ void xfinish(const Pro& p, int KR, const XRegs<MODE, MAXJ, E>& r, float* xs, float* red, bool writer) {
  const int tid = threadIdx.x;
  const bool hasprev = p.src.p != nullptr;
  if (MODE == PRO_NORM) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
      const int i = j * 1024 + tid * 4;
      float v[4] = {r.h[j].x, r.h[j].y, r.h[j].z, r.h[j].w};
      if (hasprev) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = round_act(v[e] + r.pf[j][e], p.act);
      }
      if (i < p.H) {
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (writer && p.h_out) *(float4*)(p.h_out + i) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    ss = block_sum256(ss, red);
    const float rs = 1.0f / sqrtf(ss / (float)p.H + p.eps);
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = e * 256 + tid;
      float v = r.sa[e];
      if (hasprev) v = round_act(v + r.sb[e], p.act);
      v = round_act(r.sw[e] * round_act(v * rs, p.act), p.act);
      if (i < KR) xs[i] = v;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = e * 256 + tid;
      float v = r.sa[e];
      if (MODE == PRO_SILU) v = round_act(round_act(silu_f(v), p.act) * r.sb[e], p.act);
      if (i < KR) xs[i] = v;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// activation slice -> three int8 planes + per-group parameters   (QG = 128 k per group, 16 lanes x 8 each)
//   x ~= sx * xi,  xi = 65536*hi + 256*mid + lo  (24-bit fixed point relative to the group maximum: exact for f16/bf16
//   activations down to 2^-13 of the group max, i.e. the int path reproduces f32 arithmetic)
//   gpar[2g]   = { sx/16 (float bits), 128*SB_hi, 128*SB_mid, 128*SB_lo }   SB = sum over the k with (k%8) >= 4
//   gpar[2g+1] = { 16*S_hi, 16*S_mid, 16*S_lo, 0 }                          S  = sum over the whole group
// ---------------------------------------------------------------------------------------------------------
#define XQ_MAX 8355000.0f  // < 127*65536 + 127*256 + 127

__device__ __forceinline__ void quant_x128(const float* xs, int KR, unsigned* xh, unsigned* xm, unsigned* xl, int4* gpar) {
  for (int base = 0; base < KR; base += 2048) {
    const int e0 = base + threadIdx.x * 8;
    const bool on = e0 < KR;
    float v[8];
    if (on) {
      float4 a = *(const float4*)(xs + e0), b = *(const float4*)(xs + e0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = 0.f;
    }
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) am = fmaxf(am, fabsf(v[i]));
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) am = fmaxf(am, __shfl_xor(am, m, 64));
    const float inv = am > 0.f ? XQ_MAX / am : 0.f;
    unsigned wh[2] = {0, 0}, wm[2] = {0, 0}, wl[2] = {0, 0};
    int s_hi = 0, s_mid = 0, s_lo = 0, b_hi = 0, b_mid = 0, b_lo = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int xi = (int)rintf(v[i] * inv);
      const int lo = ((xi + 128) & 255) - 128;
      const int r1 = (xi - lo) >> 8;
      const int mid = ((r1 + 128) & 255) - 128;
      const int hi = (r1 - mid) >> 8;
      wh[i >> 2] |= ((unsigned)hi & 255u) << (8 * (i & 3));
      wm[i >> 2] |= ((unsigned)mid & 255u) << (8 * (i & 3));
      wl[i >> 2] |= ((unsigned)lo & 255u) << (8 * (i & 3));
      s_hi += hi; s_mid += mid; s_lo += lo;
      if (i >= 4) { b_hi += hi; b_mid += mid; b_lo += lo; }
    }
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) {
      s_hi += __shfl_xor(s_hi, m, 64); s_mid += __shfl_xor(s_mid, m, 64); s_lo += __shfl_xor(s_lo, m, 64);
      b_hi += __shfl_xor(b_hi, m, 64); b_mid += __shfl_xor(b_mid, m, 64); b_lo += __shfl_xor(b_lo, m, 64);
    }
    if (on) {
      *(uint2*)(xh + e0 / 4) = make_uint2(wh[0], wh[1]);
      *(uint2*)(xm + e0 / 4) = make_uint2(wm[0], wm[1]);
      *(uint2*)(xl + e0 / 4) = make_uint2(wl[0], wl[1]);
      if ((threadIdx.x & 15) == 0) {
        gpar[2 * (e0 >> 7)] = make_int4(__float_as_int(am * (1.0f / (XQ_MAX * 16.0f))), 128 * b_hi, 128 * b_mid, 128 * b_lo);
        gpar[2 * (e0 >> 7) + 1] = make_int4(16 * s_hi, 16 * s_mid, 16 * s_lo, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// INT4 group-quantised GEMV (AWQ / GPTQ after repack).  grid = nst * (G/GW) blocks of 256 threads.
// ---------------------------------------------------------------------------------------------------------
#define Q4G_E 8  // slice elements per thread (KR <= 2048)

__device__ __forceinline__ void q4g_consume(const uint4 (&w)[4], int g, const uint4* xh4, const uint4* xm4, const uint4* xl4,
                                            const int4* gpar, float s, int z, float& y) {
  int Ah = 0, Am = 0, Al = 0, Bh = 0, Bm = 0, Bl = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint4 h0 = xh4[g * 8 + c * 2], h1 = xh4[g * 8 + c * 2 + 1];
    const uint4 m0 = xm4[g * 8 + c * 2], m1 = xm4[g * 8 + c * 2 + 1];
    const uint4 l0 = xl4[g * 8 + c * 2], l1 = xl4[g * 8 + c * 2 + 1];
    const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const unsigned W[4] = {w[c].x, w[c].y, w[c].z, w[c].w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int a = (int)(W[j] & 0x0F0F0F0Fu), b = (int)(W[j] & 0xF0F0F0F0u);
      Ah = __builtin_amdgcn_sdot4(a, (int)Xh[2 * j], Ah, false);
      Am = __builtin_amdgcn_sdot4(a, (int)Xm[2 * j], Am, false);
      Al = __builtin_amdgcn_sdot4(a, (int)Xl[2 * j], Al, false);
      Bh = __builtin_amdgcn_sdot4(b, (int)Xh[2 * j + 1], Bh, false);
      Bm = __builtin_amdgcn_sdot4(b, (int)Xm[2 * j + 1], Bm, false);
      Bl = __builtin_amdgcn_sdot4(b, (int)Xl[2 * j + 1], Bl, false);
    }
  }
  const int4 g1 = gpar[2 * g], g2 = gpar[2 * g + 1];
  // per plane: 16 * sum_k (q_k - z) * plane_k, exact in int32 (|.| < 2^23)
  const int Uh = (Ah << 4) + Bh + g1.y - z * g2.x;
  const int Um = (Am << 4) + Bm + g1.z - z * g2.y;
  const int Ul = (Al << 4) + Bl + g1.w - z * g2.z;
  const float f = fmaf((float)Uh, 65536.0f, fmaf((float)Um, 256.0f, (float)Ul));
  y += (s * __int_as_float(g1.x)) * f;
}

template <int MODE, int FIX, int MAXJ, int NPF>
__global__ __launch_bounds__(256) void k_gemv_q4g(const uint4* __restrict__ W, const __half* __restrict__ S,
                                                  const unsigned char* __restrict__ Z, const float* __restrict__ bias, int N, int K,
                                                  int GW, int nst, Pro pro, long long* acc, long long* zero_buf, int zero_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KR = GW * 128;
  float* xs = (float*)smem;                               // [KR]
  unsigned* xh = (unsigned*)(xs + KR);                    // [KR/4]
  unsigned* xm = xh + KR / 4;                             // [KR/4]
  unsigned* xl = xm + KR / 4;                             // [KR/4]
  int4* gpar = (int4*)(xl + KR / 4);                      // [2*GW]
  __half* sS = (__half*)(gpar + 2 * GW);                  // [4][GW][64]
  unsigned char* sZ = (unsigned char*)(sS + 4 * GW * 64); // [4][GW][64]
  float* red = (float*)(sZ + 4 * GW * 64);                // [4]

  const int st = blockIdx.x % nst, ks = blockIdx.x / nst;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nt = st * 4 + wave;
  const bool wave_on = nt * 64 < N;
  const int G = K >> 7;
  const int k0 = ks * KR, g0 = ks * GW;

  zero_duty(zero_buf, zero_n);

  // (1) group scales / zero points of this wave's tile: GW*128 B + GW*64 B, contiguous -> wide loads now, LDS later
  unsigned sreg[8], zreg[4];
  {
    const unsigned* Sg = (const unsigned*)(S + ((size_t)nt * G + g0) * 64);
    const unsigned* Zg = (const unsigned*)(Z + ((size_t)nt * G + g0) * 64);
#pragma unroll
    for (int j = 0; j < 8; j++) sreg[j] = (wave_on && lane + 64 * j < GW * 32) ? Sg[lane + 64 * j] : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) zreg[j] = (wave_on && lane + 64 * j < GW * 16) ? Zg[lane + 64 * j] : 0u;
  }

  // (2) issue the prologue's loads (L2-resident data)
  XRegs<MODE, MAXJ, Q4G_E> xr;
  xload<MODE, FIX, MAXJ, Q4G_E>(pro, k0, KR, xr);

  // (3) issue the first NPF groups of weight loads (HBM): they fly while the prologue computes
  const uint4* wp = W + ((size_t)nt * (K >> 5) + (k0 >> 5)) * 64 + lane;
  uint4 Wb[NPF][4];
#pragma unroll
  for (int b = 0; b < NPF; b++) {
#pragma unroll
    for (int c = 0; c < 4; c++) Wb[b][c] = (wave_on && b < GW) ? ldnt(wp + (b * 4 + c) * 64) : make_uint4(0, 0, 0, 0);
  }

  // (4) finish the prologue
  xfinish<MODE, MAXJ, Q4G_E>(pro, KR, xr, xs, red, blockIdx.x == 0);
  {
    unsigned* sSw = (unsigned*)(sS + wave * GW * 64);
    unsigned* sZw = (unsigned*)(sZ + wave * GW * 64);
#pragma unroll
    for (int j = 0; j < 8; j++) if (lane + 64 * j < GW * 32) sSw[lane + 64 * j] = sreg[j];
#pragma unroll
    for (int j = 0; j < 4; j++) if (lane + 64 * j < GW * 16) sZw[lane + 64 * j] = zreg[j];
  }
  if (!(pro.dbg & 4)) quant_x128(xs, KR, xh, xm, xl, gpar);
  __syncthreads();
  if (!wave_on) return;

  // (5) stream the k-range with NPF groups (NPF * 4 KiB per wave) in flight
  const uint4* xh4 = (const uint4*)xh;
  const uint4* xm4 = (const uint4*)xm;
  const uint4* xl4 = (const uint4*)xl;
  float y = 0.f;
  for (int gb = 0; gb < GW; gb += NPF) {
#pragma unroll
    for (int b = 0; b < NPF; b++) {
      const int g = gb + b;
      if (g < GW) {
        const float s = __half2float(sS[(wave * GW + g) * 64 + lane]);
        const int z = sZ[(wave * GW + g) * 64 + lane];
        if (pro.dbg & 2) y += __uint_as_float((Wb[b][0].x ^ Wb[b][1].y ^ Wb[b][2].z ^ Wb[b][3].w) & 0x007fffffu);
        else q4g_consume(Wb[b], g, xh4, xm4, xl4, gpar, s, z, y);
        if (g + NPF < GW) {
#pragma unroll
          for (int c = 0; c < 4; c++) Wb[b][c] = ldnt(wp + ((g + NPF) * 4 + c) * 64);
        }
      }
    }
  }
  const int n = nt * 64 + lane;
  if (bias != nullptr && ks == 0) y += bias[n];
  if (pro.dbg & 1) acc[n] = f2fix(y);
  else atomicAdd((unsigned long long*)(acc + n), (unsigned long long)f2fix(y));
}

static size_t q4g_smem(int GW) {
  size_t KR = (size_t)GW * 128;
  return KR * 4 + KR / 4 * 4 * 3 + (size_t)GW * 32 + (size_t)4 * GW * 64 * 2 + (size_t)4 * GW * 64 + 16;
}

// ---------------------------------------------------------------------------------------------------------
// dense row-major GEMV  W[N][K] (f16 / bf16 / f32), one wave per 4 rows at a time, x in LDS.
// Direct store of the rounded result (+ optional fused argmax partials for lm_head).
// ---------------------------------------------------------------------------------------------------------
template <int WDT>
__device__ __forceinline__ void load8(const void* W, size_t elem_off, bool on, float (&w)[8]) {
  if (WDT == BZ_F32) {
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (on) { a = ldnt((const float4*)((const float*)W + elem_off)); b = ldnt((const float4*)((const float*)W + elem_off + 4)); }
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (on) r = ldnt((const uint4*)((const unsigned short*)W + elem_off));
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (WDT == BZ_F16) {
        w[2 * i] = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu)));
        w[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16)));
      } else {
        w[2 * i] = __uint_as_float(u[i] << 16);
        w[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
      }
    }
  }
}

template <int WDT>
__global__ __launch_bounds__(256) void k_gemv_rows(const void* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                                   int rows_per_wg, Pro pro, float* out, int act, float* pval, int* pidx,
                                                   long long* zero_buf, int zero_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KP = (K + 511) & ~511;
  float* xs = (float*)smem;     // [KP] swizzled
  float* red = xs + KP;         // [4] + argmax scratch [4] + [4]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  zero_duty(zero_buf, zero_n);
  for (int i = K + threadIdx.x; i < KP; i += 256) xs[xs_pos<true>(i)] = 0.f;
  build_x_simple<true>(pro, 0, K, xs, red, blockIdx.x == 0);

  const int rpw = rows_per_wg >> 2;
  const int rbeg = blockIdx.x * rows_per_wg + wave * rpw;
  const int rend = min(rbeg + rpw, N);
  const int KC = KP >> 9;
  const float4* xs4 = (const float4*)xs;
  float bestv = -INFINITY; int besti = 0x7fffffff;
  for (int r = rbeg; r < rend; r += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kc = 0; kc < KC; kc++) {
      const int k = kc * 512 + lane * 8;
      const bool kon = k < K;
      float w[4][8];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) load8<WDT>(W, (size_t)(r + rr) * K + k, kon && (r + rr < rend), w[rr]);
      const float4 xa = xs4[kc * 128 + lane], xb = xs4[kc * 128 + 64 + lane];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        acc[rr] += w[rr][0] * xa.x + w[rr][1] * xa.y + w[rr][2] * xa.z + w[rr][3] * xa.w + w[rr][4] * xb.x + w[rr][5] * xb.y +
                   w[rr][6] * xb.z + w[rr][7] * xb.w;
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      float v = wave_sum(acc[rr]);
      if (r + rr < rend) {
        if (bias) v += bias[r + rr];
        v = round_act(v, act);
        if (lane == 0) out[r + rr] = v;
        if (v > bestv) { bestv = v; besti = r + rr; }
      }
    }
  }
  if (pval) {
    float* bv = red + 4; int* bi = (int*)(red + 8);
    if (lane == 0) { bv[wave] = bestv; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = bv[0]; int ix = bi[0];
      for (int w2 = 1; w2 < 4; w2++) if (bv[w2] > v) { v = bv[w2]; ix = bi[w2]; }
      pval[blockIdx.x] = v; pidx[blockIdx.x] = ix;
    }
  }
}

static int rows_per_wg_for(int N) {
  // ~1000 workgroups when N is large, at least 16 rows (4 per wave) per workgroup
  int r = 16;
  while (r < 256 && (N + r - 1) / r > 1024) r <<= 1;
  return r;
}
int bzk_gemv_rows_blocks(const LinearDev& L) { int r = rows_per_wg_for(L.N); return (L.N + r - 1) / r; }

int bzk_gemv(hipStream_t s, const LinearDev& L, const Pro& pro, const GemvOut& out, int act) {
  if (L.kind == LK_Q4G) {
    if (!out.acc) BZ_FAIL(BZ_E_INVALID, "q4g gemv needs a fixed-point accumulator");
    const int G = L.K / 128, GW = L.gw;
    const int nst = (L.N + 255) / 256;
    const int grid = nst * (G / GW);
    const size_t smem = q4g_smem(GW);
    const int maxj = pro.mode == PRO_NORM ? (pro.H + 1023) / 1024 : 1;
    if (GW > 16) BZ_FAIL(BZ_E_INVALID, "q4g gemv: %d groups per workgroup (max 16)", GW);
#define LAUNCH_Q4G(MODE, FIX, MJ, NPF) BZ_LAUNCH(MODE == PRO_NORM ? "gemv_q4g<norm>" : (MODE == PRO_SILU ? "gemv_q4g<silu>" : "gemv_q4g<plain>"), \
    L.algo_bytes, (k_gemv_q4g<MODE, FIX, MJ, NPF>), dim3(grid), dim3(256), smem, s, (const uint4*)L.w,                      \
    (const __half*)L.scales, (const unsigned char*)L.zeros, L.bias, L.N, L.K, GW, nst, pro, out.acc, out.zero_buf, out.zero_n)
#define LAUNCH_Q4G_N(MODE, FIX, MJ) do { if (L.npf >= 4) LAUNCH_Q4G(MODE, FIX, MJ, 4); else LAUNCH_Q4G(MODE, FIX, MJ, 2); } while (0)
#define LAUNCH_Q4G_F(MODE, MJ) do { if (pro.src.fix) LAUNCH_Q4G_N(MODE, 1, MJ); else LAUNCH_Q4G_N(MODE, 0, MJ); } while (0)
    if (pro.mode == PRO_PLAIN) LAUNCH_Q4G_F(PRO_PLAIN, 1);
    else if (pro.mode == PRO_SILU) LAUNCH_Q4G_F(PRO_SILU, 1);
    else if (maxj <= 1) LAUNCH_Q4G_F(PRO_NORM, 1);
    else if (maxj <= 2) LAUNCH_Q4G_F(PRO_NORM, 2);
    else if (maxj <= 4) LAUNCH_Q4G_F(PRO_NORM, 4);
    else if (maxj <= 8) LAUNCH_Q4G_F(PRO_NORM, 8);
    else BZ_FAIL(BZ_E_UNSUPPORTED, "hidden size %d too large for the fused norm prologue", pro.H);
#undef LAUNCH_Q4G_N
#undef LAUNCH_Q4G_F
#undef LAUNCH_Q4G
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  if (L.kind == LK_ROWS) {
    if (!out.direct) BZ_FAIL(BZ_E_INVALID, "rows gemv needs a direct output");
    const int rpw = rows_per_wg_for(L.N);
    const int grid = (L.N + rpw - 1) / rpw;
    const int KP = (L.K + 511) & ~511;
    const size_t smem = (size_t)KP * 4 + 64;
    if (smem > 160 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "K=%d too large for the rows GEMV", L.K);
#define LAUNCH_ROWS(DT) BZ_LAUNCH(out.amax_val ? "gemv_rows<lm_head+argmax>" : "gemv_rows", L.algo_bytes, (k_gemv_rows<DT>), dim3(grid), \
    dim3(256), smem, s, (const void*)L.w, L.bias, L.N, L.K, rpw, pro, out.direct, act, out.amax_val, out.amax_idx, out.zero_buf, out.zero_n)
    if (L.wdt == BZ_F16) LAUNCH_ROWS(BZ_F16); else if (L.wdt == BZ_BF16) LAUNCH_ROWS(BZ_BF16); else LAUNCH_ROWS(BZ_F32);
#undef LAUNCH_ROWS
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  BZ_FAIL(BZ_E_UNSUPPORTED, "gemv: linear kind %d not implemented", L.kind);
}

// ---------------------------------------------------------------------------------------------------------
// load-time repack (runs once, on the GPU)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned awq_nib(const uint32_t* qw, int N, int k, int n) {
  // /root/reference/src/loader/safetensors/awq.rs:29-32 : column j of a word sits at shift [0,16,4,20,8,24,12,28][j]
  const unsigned sh = ((n & 1) << 4) | ((n & 7) >> 1 << 2);
  return (qw[(size_t)k * (N >> 3) + (n >> 3)] >> sh) & 15u;
}

__global__ void k_repack_awq(const uint32_t* qw, const float* sc, const float* zf, int N, int K, int gs, uint32_t* wout, __half* sout,
                             unsigned char* zout) {
  const size_t total = (size_t)N * (K >> 3);  // output words
  const int G = K / gs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kc = (int)(t % (size_t)(K >> 5));
    const int nt = (int)(t / (size_t)(K >> 5));
    const int n = nt * 64 + lane;
    unsigned word = 0;
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const int k1 = kc * 32 + j * 8 + bb;
      const unsigned q1 = awq_nib(qw, N, k1, n), q2 = awq_nib(qw, N, k1 + 4, n);
      word |= (q1 | ((q2 ^ 8u) << 4)) << (8 * bb);
    }
    wout[idx] = word;
  }
  const size_t gt = (size_t)N * G;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < gt; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    const size_t t = idx >> 6;
    const int g = (int)(t % (size_t)G);
    const int nt = (int)(t / (size_t)G);
    const int n = nt * 64 + lane;
    sout[idx] = __float2half_rn(sc[(size_t)g * N + n]);
    zout[idx] = (unsigned char)(int)zf[(size_t)g * N + n];
  }
}

__global__ void k_repack_gptq(const uint32_t* qw, const float* sc, const uint32_t* qz, const int* perm, const int* gidx, int N, int K,
                              int gs, uint32_t* wout, __half* sout, unsigned char* zout) {
  // /root/reference/src/loader/safetensors/gptq.rs:3-8 : qweight [K/8][N] sequential nibbles, qzeros [G][N/8] packed
  const size_t total = (size_t)N * (K >> 3);
  const int G = K / gs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kc = (int)(t % (size_t)(K >> 5));
    const int nt = (int)(t / (size_t)(K >> 5));
    const int n = nt * 64 + lane;
    unsigned word = 0;
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      int k1 = kc * 32 + j * 8 + bb, k2 = k1 + 4;
      if (perm) { k1 = perm[k1]; k2 = perm[k2]; }
      const unsigned q1 = (qw[(size_t)(k1 >> 3) * N + n] >> (4 * (k1 & 7))) & 15u;
      const unsigned q2 = (qw[(size_t)(k2 >> 3) * N + n] >> (4 * (k2 & 7))) & 15u;
      word |= (q1 | ((q2 ^ 8u) << 4)) << (8 * bb);
    }
    wout[idx] = word;
  }
  const size_t gt = (size_t)N * G;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < gt; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    const size_t t = idx >> 6;
    const int gp = (int)(t % (size_t)G);      // group index in the (sorted) kernel order
    const int nt = (int)(t / (size_t)G);
    const int n = nt * 64 + lane;
    // original group id of the sorted group: g_idx of its first member (all members share it)
    const int g = (perm && gidx) ? gidx[perm[gp * gs]] : gp;
    sout[idx] = __float2half_rn(sc[(size_t)g * N + n]);
    zout[idx] = (unsigned char)(((qz[(size_t)g * (N >> 3) + (n >> 3)] >> (4 * (n & 7))) & 15u) + 1u);  // ASSUMPTION: AutoGPTQ v1 (+1)
  }
}

int bzk_repack_awq(hipStream_t s, const uint32_t* q, const float* sc, const float* z, int N, int K, int gs, void* w, void* so, void* zo) {
  hipLaunchKernelGGL(k_repack_awq, dim3(2048), dim3(256), 0, s, q, sc, z, N, K, gs, (uint32_t*)w, (__half*)so, (unsigned char*)zo);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_repack_gptq(hipStream_t s, const uint32_t* q, const float* sc, const uint32_t* qz, const int* perm, const int* gidx, int N, int K,
                    int gs, void* w, void* so, void* zo) {
  hipLaunchKernelGGL(k_repack_gptq, dim3(2048), dim3(256), 0, s, q, sc, qz, perm, gidx, N, K, gs, (uint32_t*)w, (__half*)so,
                     (unsigned char*)zo);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// dequantise the REPACKED layout back to f32 [N][K'] (K' in kernel order, i.e. perm applied) -- validates the repack
__global__ void k_dequant_q4g(const uint32_t* W, const __half* S, const unsigned char* Z, int N, int K, float* out) {
  const size_t total = (size_t)N * K;
  const int G = K >> 7;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % (size_t)K);
    const int n = (int)(idx / (size_t)K);
    const int nt = n >> 6, lane = n & 63, kc = k >> 5, j = (k & 31) >> 3, r = k & 7;
    const unsigned word = W[(((size_t)nt * (K >> 5) + kc) * 64 + lane) * 4 + j];
    const unsigned byte = (word >> (8 * (r & 3))) & 255u;
    const float q = (r < 4) ? (float)(byte & 15u) : (float)((byte >> 4) ^ 8u);
    const size_t gi = ((size_t)nt * G + (k >> 7)) * 64 + lane;
    out[idx] = (q - (float)Z[gi]) * __half2float(S[gi]);
  }
}
int bzk_dequant_q4g(hipStream_t s, const LinearDev& L, float* out) {
  hipLaunchKernelGGL(k_dequant_q4g, dim3(2048), dim3(256), 0, s, (const uint32_t*)L.w, (const __half*)L.scales,
                     (const unsigned char*)L.zeros, L.N, L.K, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
__global__ void k_dequant_rows(const void* W, int wdt, size_t total, float* out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v;
    if (wdt == BZ_F32) v = ((const float*)W)[i];
    else if (wdt == BZ_F16) v = __half2float(((const __half*)W)[i]);
    else v = __uint_as_float((unsigned)((const unsigned short*)W)[i] << 16);
    out[i] = v;
  }
}
int bzk_dequant_rows(hipStream_t s, const LinearDev& L, float* out) {
  hipLaunchKernelGGL(k_dequant_rows, dim3(2048), dim3(256), 0, s, (const void*)L.w, L.wdt, (size_t)L.N * L.K, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------
__global__ void k_embed(const void* table, int tdt, const long long* tok, int H, int act, float* h) {
  const long long t = tok[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) {
    const size_t idx = (size_t)t * H + i;
    float v;
    if (tdt == BZ_F32) v = ((const float*)table)[idx];
    else if (tdt == BZ_F16) v = __half2float(((const __half*)table)[idx]);
    else v = __uint_as_float((unsigned)((const unsigned short*)table)[idx] << 16);
    h[i] = round_act(v, act);
  }
}
int bzk_embed(hipStream_t s, const void* table, int tdt, const long long* tok, int H, int act, float* h) {
  BZ_LAUNCH("embed", (double)H * (tdt == BZ_F32 ? 4 : 2), k_embed, dim3((H + 255) / 256), dim3(256), 0, s, table, tdt, tok, H, act, h);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

__global__ void k_fix_to_f32(const long long* acc, int n, int act, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = round_act(fix2f(acc[i]), act);
}
int bzk_fix_to_f32(hipStream_t s, const long long* acc, int n, int act, float* out) {
  hipLaunchKernelGGL(k_fix_to_f32, dim3((n + 255) / 256), dim3(256), 0, s, acc, n, act, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
__global__ void k_zero64(long long* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
int bzk_zero64(hipStream_t s, long long* p, int n) {
  hipLaunchKernelGGL(k_zero64, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// KV cache access
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t kv_row_off(const KvView& kv, int layer, int kvh, int p) {
  if (kv.paged) {
    const int blk = kv.block_table[p / kv.bs];
    return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + (p % kv.bs)) * kv.hd;
  }
  return (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd;
}
__device__ __forceinline__ size_t kv_slot_off(const KvView& kv, int layer, int kvh, int slot) {
  const int blk = slot / kv.bs, o = slot % kv.bs;
  return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + o) * kv.hd;
}
__device__ __forceinline__ float kv_ld(const void* base, size_t off, int dt) {
  if (dt == BZ_F16) return __half2float(((const __half*)base)[off]);
  if (dt == BZ_BF16) return __uint_as_float((unsigned)((const unsigned short*)base)[off] << 16);
  return ((const float*)base)[off];
}
__device__ __forceinline__ void kv_st(void* base, size_t off, int dt, float v) {
  if (dt == BZ_F16) ((__half*)base)[off] = __float2half_rn(v);
  else if (dt == BZ_BF16) ((unsigned short*)base)[off] = (unsigned short)(__float_as_uint(bf16_round(v)) >> 16);
  else ((float*)base)[off] = v;
}
// 8 consecutive elements of a row -> f32
__device__ __forceinline__ void kv_ld8(const void* base, size_t off, int dt, float (&o)[8]) {
  if (dt == BZ_F32) {
    const float4 a = *(const float4*)((const float*)base + off), b = *(const float4*)((const float*)base + off + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    const uint4 r = *(const uint4*)((const unsigned short*)base + off);
    const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (dt == BZ_F16) {
        o[2 * i] = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu)));
        o[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16)));
      } else {
        o[2 * i] = __uint_as_float(u[i] << 16);
        o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// attention decode: one workgroup per kv head; fuses q/k/v finishing (fixed-point -> f32, rounding), RoPE, KV append
// and the single-query attention of the `rep` query heads that share the kv head.
//
// Wave w serves query head w (+4, +8 ...).  LANE == POSITION: lane l walks positions l, l+64, ... as its own online-
// softmax stream (m, l, o[HD] in registers; K/V rows straight to VGPRs, all loads of a row issued together), so there
// is no per-position cross-lane traffic; the 64 streams are merged once at the end (log-sum-exp weights, o reduced
// through LDS in 32-column rounds).
// ---------------------------------------------------------------------------------------------------------
#define ATT_MAXHD 256

// one cache row as raw 16-byte pieces (kept packed in registers; converted at use)
template <int HD, int KVDT>
struct KvRow {
  static constexpr int NV = (KVDT == BZ_F32) ? HD / 4 : HD / 8;
  uint4 raw[NV];
  __device__ __forceinline__ void load(const void* base, size_t off) {
    const uint4* p = (KVDT == BZ_F32) ? (const uint4*)((const float*)base + off) : (const uint4*)((const unsigned short*)base + off);
#pragma unroll
    for (int i = 0; i < NV; i++) raw[i] = p[i];
  }
  __device__ __forceinline__ void from_f32(const float* src) {   // LDS row (the token being appended)
#pragma unroll
    for (int i = 0; i < NV; i++) {
      if (KVDT == BZ_F32) {
        const float4 v = *(const float4*)(src + i * 4);
        raw[i] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
      } else {
        unsigned u[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float x0 = src[i * 8 + 2 * j], x1 = src[i * 8 + 2 * j + 1];
          if (KVDT == BZ_F16) u[j] = (unsigned)__half_as_ushort(__float2half_rn(x0)) | ((unsigned)__half_as_ushort(__float2half_rn(x1)) << 16);
          else u[j] = (__float_as_uint(bf16_round(x0)) >> 16) | (__float_as_uint(bf16_round(x1)) & 0xffff0000u);
        }
        raw[i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
    }
  }
  // element e of the row as f32 (e is a compile-time constant after unrolling)
  __device__ __forceinline__ float get(int e) const {
    if (KVDT == BZ_F32) {
      const uint4 v = raw[e >> 2];
      const unsigned u = (e & 3) == 0 ? v.x : ((e & 3) == 1 ? v.y : ((e & 3) == 2 ? v.z : v.w));
      return __uint_as_float(u);
    }
    const uint4 v = raw[e >> 3];
    const int w = (e >> 1) & 3;
    const unsigned u = w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w));
    if (KVDT == BZ_F16) return __half2float(__ushort_as_half((unsigned short)((e & 1) ? (u >> 16) : (u & 0xffffu))));
    return __uint_as_float((e & 1) ? (u & 0xffff0000u) : (u << 16));
  }
};

template <int HD, int KVDT>
__global__ __launch_bounds__(256) void k_attn_decode(AttnArgs a) {
  // grid = nq query heads.  All 256 lanes of the workgroup are position streams of ONE query head (position p -> lane
  // p % 256), merged once through LDS.  Workgroups of one GQA group recompute the (tiny) k/v finishing redundantly; the
  // first head of the group appends the new row to the cache.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int half = HD / 2;
  constexpr int LDR = HD + 4;             // padded row: 16-B aligned, conflict-free column walks
  const int rep = a.nq / a.nkv;
  float* qs = (float*)smem;               // [HD]
  float* knew = qs + HD;                  // [HD]
  float* vnew = knew + HD;                // [HD]
  float* wred = vnew + HD;                // [8]: per-wave max / sum
  float* ored = wred + 8;                 // [256][LDR]
  const int hq = blockIdx.x, kvh = hq / rep;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int pos = a.pos[0];
  const int len = a.q_only ? pos : pos + 1;
  const int ncache = pos;                 // positions [0, ncache) come from the cache; `pos` itself (if any) from LDS
  const KvView& kv = a.kv;
  zero_duty(a.zero_buf, a.zero_n);

  if (!a.q_only) {
    const float* cr = a.cos_t + (size_t)pos * half;
    const float* sr = a.sin_t + (size_t)pos * half;
    for (int idx = tid; idx < 2 * half; idx += 256) {
      const int hh = idx / half, i = idx % half;       // hh 0: this query head, 1: the kv head's new key
      const int base = hh == 0 ? hq * HD : a.nq * HD + kvh * HD;
      const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
      const float x0 = vsrc_get(a.qkv, base + ia, a.act), x1 = vsrc_get(a.qkv, base + ib, a.act);
      const float c = cr[i], s = sr[i];
      const float y0 = round_act(x0 * c - x1 * s, a.act), y1 = round_act(x1 * c + x0 * s, a.act);
      if (hh == 0) { qs[ia] = y0; qs[ib] = y1; }
      else { knew[ia] = y0; knew[ib] = y1; }
    }
    for (int i = tid; i < HD; i += 256) vnew[i] = vsrc_get(a.qkv, a.nq * HD + a.nkv * HD + kvh * HD + i, a.act);
    __syncthreads();
    if (hq % rep == 0) {   // KV append (kv_insert), once per kv head
      size_t woff;
      if (kv.paged) woff = kv_slot_off(kv, a.layer, kvh, kv.slot ? kv.slot[0] : (kv.block_table[pos / kv.bs] * kv.bs + pos % kv.bs));
      else woff = kv_row_off(kv, a.layer, kvh, pos);
      for (int i = tid; i < HD; i += 256) { kv_st(kv.k, woff + i, kv.dtype, knew[i]); kv_st(kv.v, woff + i, kv.dtype, vnew[i]); }
    }
  } else {
    for (int i = tid; i < HD; i += 256) qs[i] = ((const float*)a.qkv.p)[hq * HD + i];
    __syncthreads();
  }

  const float scale = 1.0f / sqrtf((float)HD);
  float m = -INFINITY, l = 0.f;
  float o[HD];
#pragma unroll
  for (int i = 0; i < HD; i++) o[i] = 0.f;
  for (int p = tid; p < len; p += 256) {
    KvRow<HD, KVDT> kr, vr;
    const size_t ro = p < ncache ? kv_row_off(kv, a.layer, kvh, p) : 0;
    if (p < ncache) kr.load(kv.k, ro); else kr.from_f32(knew);
    // 16-bit caches: K and V rows in flight together (128 VGPRs); f32 cache: V after the dot (register budget)
    if (KVDT != BZ_F32) { if (p < ncache) vr.load(kv.v, ro); else vr.from_f32(vnew); }
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < HD; i += 4) {
      const float4 qa = *(const float4*)(qs + i);
      d += kr.get(i) * qa.x + kr.get(i + 1) * qa.y + kr.get(i + 2) * qa.z + kr.get(i + 3) * qa.w;
    }
    if (KVDT == BZ_F32) { if (p < ncache) vr.load(kv.v, ro); else vr.from_f32(vnew); }
    const float s = d * scale;
    if (m == -INFINITY) {          // first position of this stream: e = exp(0) = 1
      m = s; l = 1.f;
#pragma unroll
      for (int i = 0; i < HD; i++) o[i] = vr.get(i);
    } else {
      const float mn = fmaxf(m, s);
      const float alpha = expf(m - mn), e = expf(s - mn);
      l = l * alpha + e;
      m = mn;
#pragma unroll
      for (int i = 0; i < HD; i++) o[i] = o[i] * alpha + e * vr.get(i);
    }
  }
  // merge the (up to) 256 streams: global max / sum through LDS, o through a padded [rows][HD] image
  const float wm = wave_max(m);
  if (lane == 0) wred[wave] = wm;
  __syncthreads();
  const float M = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  const float w = (m == -INFINITY) ? 0.f : expf(m - M);
  const float ws = wave_sum(l * w);
  if (lane == 0) wred[4 + wave] = ws;
  const int nrows = min(len, 256);
  if (tid < nrows) {
#pragma unroll
    for (int i = 0; i < HD; i += 4) *(float4*)(ored + tid * LDR + i) = make_float4(o[i] * w, o[i + 1] * w, o[i + 2] * w, o[i + 3] * w);
  }
  __syncthreads();
  const float inv = 1.0f / ((wred[4] + wred[5]) + (wred[6] + wred[7]));
  // thread t: column t % HD, rows t / HD, t / HD + 256 / HD, ...   (256 / HD row-phases)
  constexpr int PH = 256 / HD;            // 2 for HD 128, 4 for HD 64
  const int col = tid % HD, ph = tid / HD;
  float acc = 0.f;
  for (int r = ph; r < nrows; r += PH) acc += ored[r * LDR + col];
  __syncthreads();
  ored[ph * LDR + col] = acc;             // reuse the first PH rows for the phase partials
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < PH; q++) t += ored[q * LDR + tid];
    a.out[(size_t)hq * HD + tid] = round_act(t * inv, a.act);
  }
}

int bzk_attn_decode(hipStream_t s, const AttnArgs& a) {
  const size_t smem = (size_t)(3 * a.hd + 8 + 256 * (a.hd + 4)) * 4;
#define LAUNCH_ATT(HD, DT) BZ_LAUNCH("attn_decode", 0.0, (k_attn_decode<HD, DT>), dim3(a.nq), dim3(256), smem, s, a)
#define LAUNCH_ATT_DT(HD) do { if (a.kv.dtype == BZ_F16) LAUNCH_ATT(HD, BZ_F16); else if (a.kv.dtype == BZ_BF16) LAUNCH_ATT(HD, BZ_BF16); \
                               else LAUNCH_ATT(HD, BZ_F32); } while (0)
  if (a.hd == 64) LAUNCH_ATT_DT(64);
  else if (a.hd == 128) LAUNCH_ATT_DT(128);
  else BZ_FAIL(BZ_E_UNSUPPORTED, "head_dim %d unsupported (64 and 128 are built)", a.hd);
#undef LAUNCH_ATT_DT
#undef LAUNCH_ATT
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

__global__ void k_kv_insert(KvView kv, int layer, const float* k, const float* v, const int* pos, int nkv, int hd) {
  const int p = pos[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nkv * hd; i += gridDim.x * blockDim.x) {
    const int h = i / hd, d = i % hd;
    const size_t off = kv.paged ? kv_slot_off(kv, layer, h, kv.slot ? kv.slot[0] : (kv.block_table[p / kv.bs] * kv.bs + p % kv.bs))
                                : kv_row_off(kv, layer, h, p);
    kv_st(kv.k, off + d, kv.dtype, k[i]);
    kv_st(kv.v, off + d, kv.dtype, v[i]);
  }
}
int bzk_kv_insert(hipStream_t s, const KvView& kv, int layer, const float* k, const float* v, const int* pos, int nkv, int hd) {
  hipLaunchKernelGGL(k_kv_insert, dim3((nkv * hd + 255) / 256), dim3(256), 0, s, kv, layer, k, v, pos, nkv, hd);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
__global__ void k_kv_read(KvView kv, int layer, int kvh, int which, int len, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len * kv.hd; i += gridDim.x * blockDim.x) {
    const int p = i / kv.hd, d = i % kv.hd;
    out[i] = kv_ld(which ? kv.v : kv.k, kv_row_off(kv, layer, kvh, p) + d, kv.dtype);
  }
}
int bzk_kv_read(hipStream_t s, const KvView& kv, int layer, int kvh, int which, int len, float* out) {
  hipLaunchKernelGGL(k_kv_read, dim3((len * kv.hd + 255) / 256), dim3(256), 0, s, kv, layer, kvh, which, len, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// sampling
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void k_argmax_final(FinalArgs a) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < a.nparts; i += 256) {
    const float v = a.pval[i]; const int ix = a.pidx[i];
    if (better(v, ix, bv, bi)) { bv = v; bi = ix; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s && better(sv[threadIdx.x + s], si[threadIdx.x + s], sv[threadIdx.x], si[threadIdx.x])) {
      sv[threadIdx.x] = sv[threadIdx.x + s]; si[threadIdx.x] = si[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long tok = si[0];
    a.tok_out[0] = tok;
    if (a.tok_log) { const int st = a.step[0]; a.tok_log[st % a.logcap] = tok; a.step[0] = st + 1; }
    if (a.pos) a.pos[0] = a.pos[0] + 1;
  }
  if (a.zero_buf) for (int i = threadIdx.x; i < a.zero_n; i += 256) a.zero_buf[i] = 0;
}
int bzk_argmax_final(hipStream_t s, const FinalArgs& a) {
  BZ_LAUNCH("argmax_final", 0.0, k_argmax_final, dim3(1), dim3(256), 0, s, a);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// SamplingOps::logits_to_token (/root/reference/src/engine/sampling.rs:445-460): penalties over <= a few hundred
// ids, then argmax (temperature == 0).  Two launches: per-block partial argmax over penalised logits, then final.
__global__ __launch_bounds__(256) void k_penalised_argmax(const float* logits, long long V, const long long* ids, const int* cnts, int n,
                                                          float rp, float fp, float pp, float* pval, int* pidx) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
    float x = logits[i];
    for (int j = 0; j < n; j++) {
      if (ids[j] == i) {
        if (rp != 1.0f) x = (x > 0.f) ? x / rp : x * rp;  // ASSUMPTION: llama.cpp sign rule (oracle/orc_ops.c)
        x -= fp * (float)cnts[j] + pp;
      }
    }
    if (better(x, (int)i, bv, bi)) { bv = x; bi = (int)i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s && better(sv[threadIdx.x + s], si[threadIdx.x + s], sv[threadIdx.x], si[threadIdx.x])) {
      sv[threadIdx.x] = sv[threadIdx.x + s]; si[threadIdx.x] = si[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { pval[blockIdx.x] = sv[0]; pidx[blockIdx.x] = si[0]; }
}

int bzk_logits_to_token(hipStream_t s, const float* logits, long long V, const long long* ids, const int* cnts, int n, float rp, float fp,
                        float pp, float temperature, int top_k, float top_p, float min_p, unsigned long long seed, float* scratch,
                        long long* tok_out) {
  (void)top_k; (void)top_p; (void)min_p; (void)seed;
  if (temperature != 0.0f)
    BZ_FAIL(BZ_E_UNSUPPORTED, "logits_to_token: temperature > 0 sampling is not implemented yet (greedy + penalties only)");
  const int nb = 64;
  float* pval = scratch; int* pidx = (int*)(scratch + nb);
  hipLaunchKernelGGL(k_penalised_argmax, dim3(nb), dim3(256), 0, s, logits, V, ids, cnts, n, rp, fp, pp, pval, pidx);
  BZ_HIP(hipGetLastError());
  FinalArgs fa{};
  fa.pval = pval; fa.pidx = pidx; fa.nparts = nb; fa.tok_out = tok_out;
  return bzk_argmax_final(s, fa);
}

// ---------------------------------------------------------------------------------------------------------
// op-level kernels (tests; the forward path uses the fused forms above)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rms_norm(const float* x, const float* prev, const float* w, int n, float eps, int act, float* y,
                                                  float* h_out) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * n;
  const float* pr = prev ? prev + (size_t)blockIdx.x * n : nullptr;
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v = xr[i];
    if (pr) v = round_act(v + pr[i], act);
    ss += v * v;
  }
  ss = block_sum256(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)n + eps);
  for (int i = threadIdx.x; i < n; i += 256) {
    float v = xr[i];
    if (pr) v = round_act(v + pr[i], act);
    if (h_out) h_out[(size_t)blockIdx.x * n + i] = v;
    y[(size_t)blockIdx.x * n + i] = round_act(w[i] * round_act(v * rs, act), act);
  }
}
int bzk_rms_norm(hipStream_t s, const float* x, const float* prev, const float* w, int rows, int n, float eps, int act, float* y,
                 float* h_out) {
  hipLaunchKernelGGL(k_rms_norm, dim3(rows), dim3(256), 0, s, x, prev, w, n, eps, act, y, h_out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

__global__ void k_rope(float* x, int S, int nh, int hd, int position, const float* cos_t, const float* sin_t, int interleaved, int act) {
  const int half = hd >> 1;
  const size_t total = (size_t)S * nh * half;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % half);
    const size_t hv = idx / half;
    const int sidx = (int)(hv / nh);
    float* v = x + hv * hd;
    const int ia = interleaved ? 2 * i : i, ib = interleaved ? 2 * i + 1 : i + half;
    const float c = cos_t[(size_t)(position + sidx) * half + i], sn = sin_t[(size_t)(position + sidx) * half + i];
    const float x0 = v[ia], x1 = v[ib];
    v[ia] = round_act(x0 * c - x1 * sn, act);
    v[ib] = round_act(x1 * c + x0 * sn, act);
  }
}
int bzk_rope(hipStream_t s, float* x, int S, int nh, int hd, int position, const float* cos_t, const float* sin_t, int interleaved,
             int act) {
  const size_t total = (size_t)S * nh * (hd / 2);
  hipLaunchKernelGGL(k_rope, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, S, nh, hd, position, cos_t, sin_t, interleaved, act);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

__global__ void k_silu_mul(const float* g, const float* u, long long n, int act, float* y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = round_act(round_act(silu_f(g[i]), act) * u[i], act);
}
int bzk_silu_mul(hipStream_t s, const float* g, const float* u, long long n, int act, float* y) {
  hipLaunchKernelGGL(k_silu_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, u, n, act, y);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
