// bz_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the decode hot path.
//
// Design (DESIGN.md has the full story):
//   * Every linear layer at batch 1 is an HBM-bound GEMV.  INT4 group-quantised weights (AWQ/GPTQ) are repacked
//     once at load into [N/64 tiles][K/32 chunks][64 lanes][16 B] so that one wave-wide `global_load_dwordx4`
//     reads 1 KiB contiguous = 32 k-values for 64 output columns; lane == output column, so there is no
//     cross-lane reduction, and the activation slice is wave-uniform (broadcast reads from LDS).
//   * The int4 (AWQ / GPTQ) dot products run on V_DOT8_I32_I4: weights are stored as signed nibbles (q - 8), the f16
//     activation slice is split, per group of 128, into six balanced signed-nibble planes under a power-of-two scale
//     (x = 2^-e * sum_p 16^p n_p, exact for f16 data), so one 32-bit weight word meets one 32-bit plane word per
//     instruction: six dots per eight weights, no unpacking.  All group arithmetic is exact integer arithmetic and
//     the groups are summed in double: the kernel computes the exactly rounded dot product (as the oracle defines it).
//     The GGUF block formats (f32 activations) run on V_DOT4_I32_I8 over three int8 planes per 32-k chunk.
//   * Split-K partial sums are added with 64-bit INTEGER atomics in 2^-44 fixed point: integer addition is
//     associative, so the result is bit-reproducible regardless of block scheduling.  The consumer kernel
//     converts and rounds in its prologue -- there are no separate reduce / norm / activation launches.
//   * Each GEMV's prologue rebuilds its activation slice from the previous kernel's output (residual add,
//     RMSNorm, SiLU*up, rounding to the activation dtype), issues its first weight loads BEFORE doing so, and
//     stages the group scales/zeros for its tile in LDS.
#include "bz_internal.h"
#include "bz_dev.h"
#include <math.h>

#include <hip/hip_ext.h>

// ---------------------------------------------------------------------------------------------------------
// launch helper: plain launch, or -- while a profile is being taken -- hipExtLaunchKernelGGL with start/stop events
// bound to the dispatch itself (pure kernel time on the launch stream, no inter-kernel gap)
// ---------------------------------------------------------------------------------------------------------
static thread_local BzTimingSink* g_sink = nullptr;
void bzk_set_timing_sink(BzTimingSink* s) { g_sink = s; }
BzTimingSink* bzk_timing_sink() { return g_sink; }

// deterministic block sum over 256 threads; red: LDS float[4]
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return t;
}

// NTH = threads per block as a compile-time constant: blockDim.x is a 16-bit field of the dispatch packet, i.e. a VECTOR load plus a full
// s_waitcnt vmcnt(0) in front of everything that follows (gridDim.x is a scalar load batched with the kernel arguments)
template <int NTH>
__device__ __forceinline__ void zero_duty(long long* zb, int zn) {
  if (zb)
    for (int i = blockIdx.x * NTH + threadIdx.x; i < zn; i += gridDim.x * NTH) zb[i] = 0;
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

// All global loads of a batch are issued first, unconditionally (clamped index), and converted afterwards: a load under
// a condition, or a conversion right after its load, makes hipcc wait vmcnt(0) per element and serialises the batch.
template <bool FIX> struct SrcRaw { typedef float T; };
template <> struct SrcRaw<true> { typedef long long T; };
template <bool FIX>
__device__ __forceinline__ typename SrcRaw<FIX>::T src_raw(const void* p, int i) {
  if constexpr (FIX) return ((const long long*)p)[i]; else return ((const float*)p)[i];
}
template <bool FIX>
__device__ __forceinline__ float src_cvt(typename SrcRaw<FIX>::T r, int act) {
  if constexpr (FIX) return round_act(fix2f(r, act), act); else return r;
}

// NORM pass 1: v = R(h + prev) for the whole row -> sum of squares; slice elements [k0, k0+KR) are parked in xs
template <bool SWZ, bool FIX, bool PREV>
__device__ __forceinline__ double norm_pass1(const Pro& p, int k0, int KR, float* xs, bool writer) {
  const int tid = threadIdx.x;
  double ss = 0.0;
  for (int base = 0; base < p.H; base += 4096) {
    float4 hv[4];
    typename SrcRaw<FIX>::T pr[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = min(base + j * 1024 + tid * 4, p.H - 4);
      hv[j] = *(const float4*)(p.h_in + i);
      if (PREV) {
#pragma unroll
        for (int e = 0; e < 4; e++) pr[j][e] = src_raw<FIX>(p.src.p, i + e);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = base + j * 1024 + tid * 4;
      float v[4] = {hv[j].x, hv[j].y, hv[j].z, hv[j].w};
      if (PREV) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = round_act(v[e] + src_cvt<FIX>(pr[j][e], p.act), p.act);
      }
      if (i < p.H) {
        ss += ((double)(v[0] * v[0]) + (double)(v[1] * v[1])) + ((double)(v[2] * v[2]) + (double)(v[3] * v[3]));
        if (writer && p.h_out) *(float4*)(p.h_out + i) = make_float4(v[0], v[1], v[2], v[3]);
        if (i >= k0 && i < k0 + KR) {       // k0, KR and i are multiples of 4
#pragma unroll
          for (int e = 0; e < 4; e++) xs[xs_pos<SWZ>(i - k0 + e)] = v[e];
        }
      }
    }
  }
  return ss;
}

template <bool SWZ, bool FIX, bool SILU>
__device__ __forceinline__ void plain_fill(const Pro& p, int k0, int KR, float* xs) {
  const int tid = threadIdx.x;
  for (int base = 0; base < KR; base += 2048) {
    typename SrcRaw<FIX>::T a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int kk = k0 + min(base + e * 256 + tid, KR - 1);
      a[e] = src_raw<FIX>(p.src.p, kk);
      if (SILU) b[e] = src_raw<FIX>(p.src.p, p.H + kk);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int i = base + e * 256 + tid;
      float v = src_cvt<FIX>(a[e], p.act);
      if (SILU) v = round_act(round_act(silu_f(v), p.act) * src_cvt<FIX>(b[e], p.act), p.act);
      if (i < KR) xs[xs_pos<SWZ>(i)] = v;
    }
  }
}

// Simple form (loads inside): used by the ROWS kernels (no act-order permutation there).  Slice [k0, k0+KR), KR > 0.
template <bool SWZ>
__device__ void build_x_simple(const Pro& p, int k0, int KR, float* xs, float* red, bool writer) {
  const int tid = threadIdx.x;
  if (KR <= 0) { __syncthreads(); return; }
  if (p.mode == PRO_NORM) {
    double ssd;
    if (p.src.p == nullptr) ssd = norm_pass1<SWZ, false, false>(p, k0, KR, xs, writer);
    else if (p.src.fix) ssd = norm_pass1<SWZ, true, true>(p, k0, KR, xs, writer);
    else ssd = norm_pass1<SWZ, false, true>(p, k0, KR, xs, writer);
    const float ss = (float)block_sum_d<4>(ssd);   // the rounded exact sum (also orders the xs writes of pass 1 before the reads below)
    const float rs = rms_scale(ss, (float)p.H, p.eps);
    for (int base = 0; base < KR; base += 2048) {
      float ww[8];
#pragma unroll
      for (int e = 0; e < 8; e++) ww[e] = p.norm_w[k0 + min(base + e * 256 + tid, KR - 1)];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int i = base + e * 256 + tid;
        if (i < KR) xs[xs_pos<SWZ>(i)] = round_act(ww[e] * round_act(xs[xs_pos<SWZ>(i)] * rs, p.act), p.act);
      }
    }
  } else if (p.mode == PRO_GATED) {
    // Mamba2 gated RMSNorm: v = R(y * R(silu(z))) ; x = R(w * R(v * rsqrt(mean_group(v^2) + eps)))   (src = y, h_in = z, H = d_inner)
    // whole-vector form (k0 == 0, KR == H); the decode step uses GATED2, where the SSM kernel has done the gate and the sums
    const int G = p.aux > 0 ? p.aux : 1, gsz = p.H / G;
    for (int g = 0; g < G; g++) {
      float ss = 0.f;
      for (int i = tid; i < gsz; i += 256) {
        const int kk = g * gsz + i;
        const float v = round_act(vsrc_get(p.src, kk, p.act) * round_act(silu_f(p.h_in[kk]), p.act), p.act);
        xs[xs_pos<SWZ>(kk)] = v;
        ss += v * v;
      }
      ss = block_sum256(ss, red);
      const float rs = rms_scale(ss, (float)gsz, p.eps);
      for (int i = tid; i < gsz; i += 256) {
        const int kk = g * gsz + i;
        xs[xs_pos<SWZ>(kk)] = round_act(p.norm_w[kk] * round_act(xs[xs_pos<SWZ>(kk)] * rs, p.act), p.act);
      }
    }
  } else if (p.mode == PRO_GATED2) {
    // gated RMSNorm with the gate and the per-head sums of squares already applied / reduced by the SSM kernel
    const int G = p.aux > 0 ? p.aux : 1, gsz = p.H / G, hpg = p.aux2 / G;
    if (tid < G) {
      double ss = 0.0;                                  // per-head exact sums arrive as hi + lo floats (k_ssm_step)
      for (int h = 0; h < hpg; h++) ss += (double)p.h_in[2 * (tid * hpg + h)] + (double)p.h_in[2 * (tid * hpg + h) + 1];
      red[4 + tid] = rms_scale((float)ss, (float)gsz, p.eps);
    }
    __syncthreads();
    for (int base = 0; base < KR; base += 2048) {
      float vv[8], ww[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int kk = k0 + min(base + e * 256 + tid, KR - 1);
        vv[e] = ((const float*)p.src.p)[kk];
        ww[e] = p.norm_w[kk];
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int i = base + e * 256 + tid;
        if (i < KR) xs[xs_pos<SWZ>(i)] = round_act(ww[e] * round_act(vv[e] * red[4 + (k0 + i) / gsz], p.act), p.act);
      }
    }
  } else if (p.mode == PRO_SILU) {
    if (p.src.fix) plain_fill<SWZ, true, true>(p, k0, KR, xs); else plain_fill<SWZ, false, true>(p, k0, KR, xs);
  } else {
    if (p.src.fix) plain_fill<SWZ, true, false>(p, k0, KR, xs); else plain_fill<SWZ, false, false>(p, k0, KR, xs);
  }
  __syncthreads();
}

// Split form for the int4 GEMV: `xload` only ISSUES the prologue's global loads into registers, then the caller
// issues its first weight loads, then `xfinish` computes.  vmcnt is in-order, so this order lets the HBM weight
// loads fly while the (L2-resident) prologue data is consumed.  MODE / FIX are compile-time so that every array
// below is statically indexed and stays in registers.
template <int FIX>
__device__ __forceinline__ float vget(const void* p, int i, int act) {
  if (FIX) return round_act(fix2f(((const long long*)p)[i], act), act);
  return ((const float*)p)[i];
}

// raw value as loaded (f32, or the 64-bit fixed-point accumulator); converted only in xfinish so that the loads of the
// whole prologue are issued back to back (a conversion right after its load makes hipcc wait vmcnt(0) per element)
template <int FIX> struct RawT { typedef float T; };
template <> struct RawT<1> { typedef long long T; };
template <int FIX>
__device__ __forceinline__ typename RawT<FIX>::T vraw(const void* p, int i) {
  if (FIX) return (typename RawT<FIX>::T)((const long long*)p)[i];
  return (typename RawT<FIX>::T)((const float*)p)[i];
}
template <int FIX>
__device__ __forceinline__ float vcvt(typename RawT<FIX>::T r, int act) {
  if (FIX) return round_act(fix2f((long long)r, act), act);
  return (float)r;
}

template <int MODE, int FIX, int MAXJ, int E>
struct XRegs {
  typedef typename RawT<FIX>::T R;
  float4 h[MODE == PRO_NORM ? MAXJ : 1];     // NORM: full-H pass, element i = j*1024 + tid*4
  R pf[MODE == PRO_NORM ? MAXJ : 1][4];      // NORM: deferred residual (raw)
  float sa_f[MODE == PRO_NORM ? E : 1];      // NORM: h of the slice
  R sa[MODE == PRO_NORM ? 1 : E];            // PLAIN x / SILU gate (raw)
  R sb[MODE == PRO_PLAIN ? 1 : E];           // NORM prev / SILU up (raw)
  float sw[MODE == PRO_NORM ? E : 1];        // NORM weight
};

template <int MODE, int FIX, int MAXJ, int E>
__device__ __forceinline__ void xload(const Pro& p, int k0, int KR, XRegs<MODE, FIX, MAXJ, E>& r) {
  // NOTE: every load is unconditional at a clamped (always valid) index.  `cond ? load : 0` makes hipcc branch around
  // the load and wait vmcnt(0) after it, which serialises the whole prologue (and everything issued before it).
  const int tid = threadIdx.x;
  const bool hasprev = p.src.p != nullptr;
  const void* prevp = hasprev ? p.src.p : (const void*)p.h_in;   // any valid address when there is no residual
  if (MODE == PRO_NORM) {
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
      int i = j * 1024 + tid * 4;
      i = i < p.H ? i : 0;
      r.h[j] = *(const float4*)(p.h_in + i);
#pragma unroll
      for (int e = 0; e < 4; e++) r.pf[j][e] = vraw<FIX>(prevp, (FIX || hasprev) ? i + e : 0);
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    int i = e * 256 + tid;
    i = i < KR ? i : 0;
    const int kk = p.perm ? p.perm[k0 + i] : (k0 + i);
    if (MODE == PRO_NORM) {
      r.sa_f[e] = p.h_in[kk];
      r.sb[e] = vraw<FIX>(prevp, kk);
      r.sw[e] = p.norm_w[kk];
    } else if (MODE == PRO_SILU) {
      r.sa[e] = vraw<FIX>(p.src.p, kk);
      r.sb[e] = vraw<FIX>(p.src.p, p.H + kk);
    } else {
      r.sa[e] = vraw<FIX>(p.src.p, kk);
    }
  }
}

template <int MODE, int FIX, int MAXJ, int E>
__device__ __forceinline__ void xfinish(const Pro& p, int KR, const XRegs<MODE, FIX, MAXJ, E>& r, float* xs, float* red, bool writer) {
  const int tid = threadIdx.x;
  const bool hasprev = p.src.p != nullptr;
  if (MODE == PRO_NORM) {
    double ssd = 0.0;
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
      const int i = j * 1024 + tid * 4;
      float v[4] = {r.h[j].x, r.h[j].y, r.h[j].z, r.h[j].w};
      if (hasprev) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = round_act(v[e] + vcvt<FIX>(r.pf[j][e], p.act), p.act);
      }
      if (i < p.H) {
        ssd += ((double)(v[0] * v[0]) + (double)(v[1] * v[1])) + ((double)(v[2] * v[2]) + (double)(v[3] * v[3]));
        if (writer && p.h_out) *(float4*)(p.h_out + i) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    const float ss = (float)block_sum_d<4>(ssd);
    const float rs = rms_scale(ss, (float)p.H, p.eps);
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = e * 256 + tid;
      float v = r.sa_f[e];
      if (hasprev) v = round_act(v + vcvt<FIX>(r.sb[e], p.act), p.act);
      v = round_act(r.sw[e] * round_act(v * rs, p.act), p.act);
      if (i < KR) xs[i] = v;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = e * 256 + tid;
      float v = vcvt<FIX>(r.sa[e], p.act);
      if (MODE == PRO_SILU) v = round_act(round_act(silu_f(v), p.act) * vcvt<FIX>(r.sb[e], p.act), p.act);
      if (i < KR) xs[i] = v;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// INT4 group-quantised GEMV (AWQ / GPTQ after repack).  grid = nst * (G/GW) blocks of 256 threads.
// ---------------------------------------------------------------------------------------------------------
#define Q4G_E 8  // slice elements per thread (KR <= 2048)

// ---------------------------------------------------------------------------------------------------------
// Fused MLP (INT4):  acc_down += Wd[:, slice] . R(R(silu(R(Wg[slice] x))) * R(Wu[slice] x)),  x = RMSNorm(R(h + prev))
// grid = I/64 blocks of 512 threads.  A block owns 64 intermediate columns: gate and up over the FULL K (8 waves split K,
// no split-K across blocks, so SiLU*up is final inside the block) and then its 64-k slab of down_proj for all outputs.
// down_proj is a sum over the intermediate dimension, so the slabs combine through the fixed-point accumulator.
// One launch streams 81 % of a layer's bytes with one fused-norm prologue per block.
// ---------------------------------------------------------------------------------------------------------
// Structure of the launch (per-wave stamps of the diagnostic build, profiles/r02_mlp_stamps_*.txt):
//  * a CU keeps ~40 KiB of HBM loads in flight; a wave that asks for more STALLS AT ISSUE -- in program order, so whatever it would do
//    next waits too.  Round 1 issued 8 KiB of weights per wave ahead of the norm: waves 10..15 sat in the issue queue until 4.4 us, the
//    block-wide sum of squares (every wave held 1/16 of the row) was complete at 5.7 us and the planes at 7.8 us; asking for little
//    before the norm instead (one chunk per wave) finished the planes at 4 us but left HBM idle from 1.5 to 4 us -- same total.
//    Now the waves have ROLES: the first half ("streamers") request their whole gate / up range at entry and sit in the issue queue,
//    which is exactly what keeps HBM busy from 0.2 us on; the second half holds the row (an octet per thread), does the norm and builds
//    the planes undisturbed, and starts its own stream afterwards.  The two halves meet through two LDS counters (sum of squares
//    complete; planes published) instead of workgroup barriers, which a streamer could only reach after its issue stall.
//  * the row never goes through LDS as f32: a thread normalises its eight elements in registers and builds the six plane words.
//  * the tail (sum of the waves' partials, SiLU * up, its 64-value quantisation) runs in ONE wave between two barriers instead of four.
template <int FIX, int GPW, int TPW, int NW, int ACT, int DIAG = 0>   // NW waves per block; GPW k-groups per wave (gate/up), TPW output tiles per wave (down); DIAG: stamp build
__global__ __launch_bounds__(NW * 64) void k_mlp_q4g(const uint4* __restrict__ Wgu, const __half* __restrict__ Sgu, const unsigned char* __restrict__ Zgu,
                                                 const float* __restrict__ bgu, const uint4* __restrict__ Wd, const __half* __restrict__ Sd,
                                                 const unsigned char* __restrict__ Zd, const float* __restrict__ bd, int H, int I, Pro pro,
                                                 long long* acc, long long* zero_buf, int zero_n) {
  static_assert(GPW == 2, "a wave owns two 128-k groups of gate / up");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* xpl = (uint4*)smem;                  // [H/32 chunks][6 planes]: 3 H bytes
  const int G = H >> 7;
  int4* gpar = (int4*)(xpl + (H >> 5) * XQ_NP);   // [2G]
  constexpr int NTH = NW * 64, NP = NW / 2;   // NP prologue waves (the second half)
  double* part = (double*)(gpar + 2 * G);     // [NW][128]
  uint4* apl = (uint4*)(part + NW * 128);     // [2 chunks][6 planes]
  int4* apar = (int4*)(apl + 2 * XQ_NP);      // [2]
  double* dred = (double*)(apar + 2);         // [NP]
  volatile unsigned* cnt = (volatile unsigned*)(dred + NW);   // [0]: waves whose sum of squares is stored, [1]: waves whose planes are stored

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool prolog = wave >= NP;             // wave-uniform role
  const int sl = blockIdx.x;                  // intermediate slice: columns [64 sl, 64 sl + 64)
  const int NTI = I >> 6;                     // gate tile = sl, up tile = NTI + sl
  const int gbeg = wave * GPW;                // G == NW * GPW
  const int tbeg = wave * TPW;                // H / 64 == NW * TPW
  const int GD = I >> 7, gd = sl >> 1;        // down's quantisation group of these 64 k
  // all kernel arguments in ONE scalar-load batch (the compiler otherwise fetches late-used ones lazily: a ~0.3 us round trip each time)
  asm volatile("" :: "s"(zero_buf), "s"(zero_n), "s"(acc), "s"(pro.h_in), "s"(pro.src.p), "s"(pro.norm_w), "s"(pro.h_out), "s"(pro.H),
               "s"(H), "s"(I), "s"(Wgu), "s"(Sgu), "s"(Zgu), "s"(Wd), "s"(Sd), "s"(Zd));
  // diagnostic instantiation only (bz_tune_mlp): s_memrealtime (100 MHz) per wave at phase boundaries, kept in scalar registers and stored
  // once at the end (a store per stamp would sit in every later vmcnt wait)
  unsigned long long T[15];
#define MSTAMP(i) do { if (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T[i]) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } } while (0)
#pragma unroll
  for (int i = 0; i < 15; i++) T[i] = 0;
  MSTAMP(0);
  static_assert(GPW * NW * 128 == NP * 64 * 8, "the prologue half covers the row with eight elements per thread");
  const uint4* wg = Wgu + ((size_t)sl * (H >> 5) + gbeg * 4) * 64 + lane;
  const uint4* wu = Wgu + ((size_t)(NTI + sl) * (H >> 5) + gbeg * 4) * 64 + lane;
  uint4 Ag[8], Au[8];                  // chunk c of this wave's 256-k range: group c >> 2
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  if (prolog) {
    // (1p) prologue loads: an octet per thread -- h, deferred residual, norm weight; then the head of this wave's stream (2 KiB: no stall)
    const bool hasprev = pro.src.p != nullptr;
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
    const int i0 = (tid - NP * 64) * 8;
    const float4 ha = *(const float4*)(pro.h_in + i0), hb = *(const float4*)(pro.h_in + i0 + 4);
    typename RawT<FIX>::T pv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) pv[e] = vraw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
    const float4 na = *(const float4*)(pro.norm_w + i0), nb = *(const float4*)(pro.norm_w + i0 + 4);
    __builtin_amdgcn_sched_barrier(0);   // the prologue's loads go AHEAD of the weight stream (vmcnt is in-order)
    Ag[0] = ldnt(wg); Au[0] = ldnt(wu);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // the counters are zero (nobody waits for data here)
    MSTAMP(1);
    // (2p) h' = R(h + R(prev)), sum of squares (exact: double), 1 / rms
    float v[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    if (hasprev) {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = round_t<ACT>(v[e] + round_t<ACT>(FIX ? fix2f((long long)pv[e], ACT) : (float)pv[e]));
    }
    if (blockIdx.x == 0 && pro.h_out) { *(float4*)(pro.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
    double ssd = 0.0;
#pragma unroll
    for (int e = 0; e < 8; e += 2) ssd += (double)(v[e] * v[e]) + (double)(v[e + 1] * v[e + 1]);
    ssd = wave_sum_d(ssd);
    if (lane == 0) { dred[wave - NP] = ssd; __builtin_amdgcn_s_waitcnt(0xc07f); atomicAdd((unsigned*)&cnt[0], 1u); }
    lds_wait_count(&cnt[0], NP);
    ssd = ((dred[0] + dred[1]) + (dred[2] + dred[3]));
    if (NP == 8) ssd += ((dred[4] + dred[5]) + (dred[6] + dred[7]));
    const float ss = (float)ssd;                  // the rounded exact sum of squares (oracle: orc_rms_norm)
    const float rs = rms_scale(ss, (float)H, pro.eps);
    MSTAMP(2);
    // (3p) x = R(w R(h' rs)) in registers -> the octet's six plane words, the group's sums (16 threads)
    const float nwv[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
    float x[8];
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { x[e] = round_t<ACT>(nwv[e] * round_t<ACT>(v[e] * rs)); am = fmaxf(am, fabsf(x[e])); }
    am = grp_reduce<16, OpMax>(am);
    unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
    xq_split8(x, am, w, sp, cs);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<16, OpAdd>(sp[p]);
    const int4 g2w = xq_gpar_hi<16>(w, sp);
    const int oct = tid - NP * 64;                                        // octet index in the row: chunk oct >> 2, weight word oct & 3
    unsigned* plw = (unsigned*)xpl + ((oct >> 2) * XQ_NP) * 4 + (oct & 3);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w[p];
    if ((lane & 15) == 0) {
      gpar[2 * (oct >> 4)] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
      gpar[2 * (oct >> 4) + 1] = g2w;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                   // lgkmcnt(0): this wave's LDS stores are done
    if (lane == 0) atomicAdd((unsigned*)&cnt[1], 1u);
    MSTAMP(4);
    // (4p) the rest of this wave's stream
#pragma unroll
    for (int c = 1; c < 8; c++) { Ag[c] = ldnt(wg + c * 64); Au[c] = ldnt(wu + c * 64); }
  } else {
    // (1s) streamers: the whole gate / up range of the wave goes out now (16 KiB; the wave sits in the issue queue, HBM stays busy)
    __syncthreads();                     // the counters are zero
#pragma unroll
    for (int c = 0; c < 8; c++) { Ag[c] = ldnt(wg + c * 64); Au[c] = ldnt(wu + c * 64); }
    MSTAMP(1);
    zero_duty<NP * 64>(zero_buf, zero_n);   // (threads 0 .. NP*64-1 are exactly the streamers)
    MSTAMP(2);
    MSTAMP(4);
  }
  float sg[2], su[2]; int zg[2], zu[2];
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const size_t ig = ((size_t)sl * G + gbeg + b) * 64 + lane, iu = ((size_t)(NTI + sl) * G + gbeg + b) * 64 + lane;
    sg[b] = __half2float(Sgu[ig]); zg[b] = Zgu[ig]; su[b] = __half2float(Sgu[iu]); zu[b] = Zgu[iu];
  }
  lds_wait_count(&cnt[1], NP);           // the planes of the whole row are in LDS
  // (5) gate / up dots chunk by chunk (the compiler counts the waits); the down slab goes out behind the first group
  double yg = 0.0, yu = 0.0;
  uint4 D[TPW][2];
  float sd[TPW]; int zd[TPW];
#pragma unroll
  for (int b = 0; b < 2; b++) {
    if (b == 0) MSTAMP(5);
    if (b == 1) MSTAMP(7);
    if (b == 0) q4g_consume2_at<0>(Ag, Au, gbeg, xpl, gpar, sg[0], zg[0], su[0], zu[0], yg, yu);
    else q4g_consume2_at<4>(Ag, Au, gbeg + 1, xpl, gpar, sg[1], zg[1], su[1], zu[1], yg, yu);
    if (b == 0) MSTAMP(6);
    if (b == 1) MSTAMP(8);
    if (b == 0) {
#pragma unroll
      for (int q = 0; q < TPW; q++) {
        const uint4* wp = Wd + ((size_t)(tbeg + q) * (I >> 5) + 2 * sl) * 64 + lane;
        D[q][0] = ldnt(wp); D[q][1] = ldnt(wp + 64);
      }
    }
  }
  MSTAMP(9);
  part[wave * 128 + lane] = yg;
  part[wave * 128 + 64 + lane] = yu;
  // the slab's scales / zero points (L2-resident): requested here, under the partial-sum barrier and the tail -- eight registers less across the second group's dots
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    const size_t si = ((size_t)(tbeg + q) * GD + gd) * 64 + lane;
    sd[q] = __half2float(Sd[si]); zd[q] = Zd[si];
  }
  __syncthreads();
  MSTAMP(10);
  // (6) wave 0: sum of the waves' k-range partials (exact, fixed order), ONE rounding to f32 (the oracle's definition), + bias, R, SiLU * up,
  //     and the 64 values' planes -- one value per lane, an octet = 8 lanes
  if (wave == 0) {
    double tg = 0.0, tu = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) { tg += part[w2 * 128 + lane]; tu += part[w2 * 128 + 64 + lane]; }
    float fg = (float)tg, fu = (float)tu;
    if (bgu) { fg += bgu[sl * 64 + lane]; fu += bgu[I + sl * 64 + lane]; }
    const float a = round_t<ACT>(round_t<ACT>(silu_f(round_t<ACT>(fg))) * round_t<ACT>(fu));
    const float am = wave_max(fabsf(a));
    const unsigned eb = (__float_as_uint(am) >> 23) & 255u;
    const bool live = eb >= 32u && eb < 255u;
    const float inv = live ? __uint_as_float((283u - eb) << 23) : 0.f;
    const float cs = live ? __uint_as_float((eb - 29u) << 23) : 0.f;
    const unsigned code = ((unsigned)(int)rintf(a * inv) + 0x88888888u) ^ 0x88888888u;
    const int lowfl = __builtin_amdgcn_ballot_w64((code & 0xFFu) != 0u) != 0ull;                  // some value of the 64 has bits in the low planes
    const int o = lane & 7, sh = 4 * (2 * (o & 3) + (o >> 2));                // k offset o -> nibble 2 (o & 3) + (o >> 2)
    int sp[XQ_NP];
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) {
      const int nib = p < XQ_NM ? p + 2 : p - XQ_NM;                                // plane index -> nibble of the code (xq_split8's order)
      const int wv = grp_reduce<8, OpOr>((int)(((code >> (4 * nib)) & 15u) << sh));   // the octet's word of plane p, in all of its lanes
      if (o == 0) ((unsigned*)apl)[((lane >> 5) * XQ_NP + p) * 4 + ((lane >> 3) & 3)] = (unsigned)wv;
      int t = __builtin_amdgcn_sdot8(wv, 0x11111111, 0, false);                    // the octet's sum (the same in its 8 lanes)
      t += dpp_get<DPP_ROR8>(t);                                                   // + the other octet of the 16-lane row
      const bz_u2_t r1 = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
      t = (int)r1.x + (int)r1.y;
      const bz_u2_t r2 = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
      sp[p] = (int)r2.x + (int)r2.y;
    }
    if (lane == 0) { apar[0] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]); apar[1] = make_int4(sp[3], sp[4], sp[5], xq_pack_low(sp[6], sp[7], lowfl)); }
  }
  __syncthreads();
  MSTAMP(11);
  // (7) this wave's down slab
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    double y = 0.0;
    q4g_consume_n<2>(D[q], 0, 0, apl, apar, sd[q], zd[q], y);
    const int n = (tbeg + q) * 64 + lane;
    if (bd != nullptr && sl == 0) y += (double)bd[n];
    if (q == 0) MSTAMP(12);
    atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(y, ACT));
  }
  MSTAMP(13);
  if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  MSTAMP(14);
  if (DIAG && pro.stamps && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 113)) {
#pragma unroll
    for (int i = 0; i < 15; i++) pro.stamps[((blockIdx.x ? 1 : 0) * 16 + wave) * 16 + i] = (long long)((i == 3) ? T[2] : T[i]);
  }
#undef MSTAMP
}

static size_t mlp_smem(int H) { return (size_t)(H >> 5) * XQ_NP * 16 + (size_t)(H >> 7) * 32 + 16 * 128 * 8 + 2 * XQ_NP * 16 + 32 + 16 * 8 + 64; }   // planes, gpar, part, apl, apar, dred, counters

bool bzk_mlp_fusable(const LinearDev& gu, const LinearDev& dn, int H, int I) {
  return gu.kind == LK_Q4G && dn.kind == LK_Q4G && !gu.perm && !dn.perm && gu.N == 2 * I && gu.K == H && dn.N == H && dn.K == I &&
         (H == 2048 || H == 4096) && I % 128 == 0 && mlp_smem(H) <= 64 * 1024;   // (+ f16 activations: checked by the launcher)
}

int bzk_mlp_q4g(hipStream_t s, const LinearDev& gu, const LinearDev& dn, int H, int I, const Pro& pro, long long* acc, long long* zero_buf, int zero_n) {
  if (!bzk_mlp_fusable(gu, dn, H, I) || pro.act != BZ_F16) BZ_FAIL(BZ_E_INVALID, "fused MLP does not apply to this shape / activation dtype");
  const size_t smem = mlp_smem(H);
  const double bytes = (double)gu.algo_bytes + (double)dn.algo_bytes;
#define LAUNCH_MLP(FIX, GP, TP, W_, DG) BZ_LAUNCH("mlp_q4g<norm+gate/up+silu+down>", bytes, (k_mlp_q4g<FIX, GP, TP, W_, BZ_F16, DG>), dim3(I / 64), dim3(W_ * 64), smem, s, \
    (const uint4*)gu.w, (const __half*)gu.scales, (const unsigned char*)gu.zeros, gu.bias, (const uint4*)dn.w, (const __half*)dn.scales,            \
    (const unsigned char*)dn.zeros, dn.bias, H, I, pro, acc, zero_buf, zero_n)
  if (H == 4096 && pro.stamps) LAUNCH_MLP(1, 2, 4, 16, 1);     // diagnostic build (bz_tune_mlp)
  else if (H == 4096) { if (pro.src.fix) LAUNCH_MLP(1, 2, 4, 16, 0); else LAUNCH_MLP(0, 2, 4, 16, 0); }
  else { if (pro.src.fix) LAUNCH_MLP(1, 2, 4, 8, 0); else LAUNCH_MLP(0, 2, 4, 8, 0); }
#undef LAUNCH_MLP
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

template <int MODE, int FIX, int MAXJ, int NPF>
__global__ __launch_bounds__(256) void k_gemv_q4g(const uint4* __restrict__ W, const __half* __restrict__ S,
                                                  const unsigned char* __restrict__ Z, const float* __restrict__ bias, int N, int K,
                                                  int GW, int nst, Pro pro, long long* acc, long long* zero_buf, int zero_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KR = GW * 128;
  float* xs = (float*)smem;                               // [KR]
  uint4* xpl = (uint4*)(xs + KR);                         // [KR/32 chunks][6 planes]: 3 KR bytes
  int4* gpar = (int4*)(xpl + (KR >> 5) * XQ_NP);          // [2*GW]
  __half* sS = (__half*)(gpar + 2 * GW);                  // [4][GW][64]
  unsigned char* sZ = (unsigned char*)(sS + 4 * GW * 64); // [4][GW][64]
  float* red = (float*)(sZ + 4 * GW * 64);                // [4]

  const int st = blockIdx.x % nst, ks = blockIdx.x / nst;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nt = st * 4 + wave;
  const bool wave_on = nt * 64 < N;
  const int G = K >> 7;
  const int k0 = ks * KR, g0 = ks * GW;
  // diagnostic build only (BZ_QKV_STAMPS): s_memrealtime (100 MHz) at phase boundaries of the first and the last block
#define QSTAMP(i) do { if (pro.stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    pro.stamps[(blockIdx.x == 0 ? 0 : 8) + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
  QSTAMP(0);

  zero_duty<256>(zero_buf, zero_n);

  // (1) group scales / zero points of this wave's tile: GW*128 B + GW*64 B, contiguous -> wide loads now, LDS later
  const int ntc = wave_on ? nt : 0;          // clamped tile for addressing: loads are unconditional (see xload)
  unsigned sreg[8], zreg[4];
  {
    const unsigned* Sg = (const unsigned*)(S + ((size_t)ntc * G + g0) * 64);
    const unsigned* Zg = (const unsigned*)(Z + ((size_t)ntc * G + g0) * 64);
#pragma unroll
    for (int j = 0; j < 8; j++) sreg[j] = Sg[min(lane + 64 * j, GW * 32 - 1)];
#pragma unroll
    for (int j = 0; j < 4; j++) zreg[j] = Zg[min(lane + 64 * j, GW * 16 - 1)];
  }

  // (2) issue the prologue's loads (L2-resident data)
  XRegs<MODE, FIX, MAXJ, Q4G_E> xr;
  xload<MODE, FIX, MAXJ, Q4G_E>(pro, k0, KR, xr);
  __builtin_amdgcn_sched_barrier(0);   // keep the prologue's loads AHEAD of the weight stream (vmcnt is in-order)

  // (3) issue the first NPF groups of weight loads (HBM): they fly while the prologue computes
  const uint4* wp = W + ((size_t)ntc * (K >> 5) + (k0 >> 5)) * 64 + lane;
  uint4 Wb[NPF][4];
#pragma unroll
  for (int b = 0; b < NPF; b++) {
    const int bc = min(b, GW - 1);           // NPF <= GW by construction of the launcher; clamp keeps the address valid anyway
#pragma unroll
    for (int c = 0; c < 4; c++) Wb[b][c] = ldnt(wp + (bc * 4 + c) * 64);
  }

  QSTAMP(1);
  // (4) finish the prologue
  xfinish<MODE, FIX, MAXJ, Q4G_E>(pro, KR, xr, xs, red, blockIdx.x == 0);
  QSTAMP(2);
  {
    unsigned* sSw = (unsigned*)(sS + wave * GW * 64);
    unsigned* sZw = (unsigned*)(sZ + wave * GW * 64);
#pragma unroll
    for (int j = 0; j < 8; j++) if (lane + 64 * j < GW * 32) sSw[lane + 64 * j] = sreg[j];
#pragma unroll
    for (int j = 0; j < 4; j++) if (lane + 64 * j < GW * 16) sZw[lane + 64 * j] = zreg[j];
  }
  if (!(pro.dbg & 4)) quant_x128<256>(xs, KR, xpl, gpar);
  __syncthreads();
  QSTAMP(3);
  if (!wave_on) return;

  // (5) stream the k-range with NPF groups (NPF * 4 KiB per wave) in flight
  double y = 0.0;
  for (int gb = 0; gb < GW; gb += NPF) {
#pragma unroll
    for (int b = 0; b < NPF; b++) {
      const int g = gb + b;
      if (g < GW) {
        const float s = __half2float(sS[(wave * GW + g) * 64 + lane]);
        const int z = sZ[(wave * GW + g) * 64 + lane];
        if (pro.dbg & 2) y += (double)__uint_as_float((Wb[b][0].x ^ Wb[b][1].y ^ Wb[b][2].z ^ Wb[b][3].w) & 0x007fffffu);
        else q4g_consume(Wb[b], g, xpl, gpar, s, z, y);
        if (g + NPF < GW) {
#pragma unroll
          for (int c = 0; c < 4; c++) Wb[b][c] = ldnt(wp + ((g + NPF) * 4 + c) * 64);
        }
      }
    }
  }
  QSTAMP(4);
  const int n = nt * 64 + lane;
  if (bias != nullptr && ks == 0) y += (double)bias[n];
  if (pro.dbg & 1) acc[n] = d2fix(y, pro.act);
  else atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(y, pro.act));
  QSTAMP(5);
#undef QSTAMP
}

// ---------------------------------------------------------------------------------------------------------
// "Slim" int4 GEMV with the fused residual + RMSNorm prologue, for the q/k/v projection (N ~ 6K, K = H).
// The stamp timeline of k_gemv_q4g on this shape (scripts/qkv_stamps.py) shows where its 12.7 us go: 3 us to reach the first
// instruction past the load issue, 4.3 us waiting for the prologue's 48 KB per block (h + the 64-bit fixed-point sums of the
// previous launch: 600 blocks x 48 KB = 29 MB through ~10 B/clk/CU), 1.4 us quantising, 0.6 us of dot products.  The weights
// were never the problem.  Here a block is 512 threads = 8 waves that share one 256-k slice (so a block quantises 256 activations)
// and take one 64-column tile each (8 KiB of weights per wave, all in flight before the prologue); grid = (K/256) x ceil(N/512):
// 192 blocks for Llama-3-8B, i.e. 9 MB of prologue traffic instead of 29 MB.
// ---------------------------------------------------------------------------------------------------------
// Wave roles (round 2, as in the fused MLP), 12 waves: waves 0-3 hold the row (8 NJ elements per thread), do the norm and build the slice's
// planes; waves 4-11 own one 64-column tile each (8 KiB of weights, requested at entry) and do the dots once the planes are published.  The
// halves meet through LDS counters.  Per-wave stamps (scripts/tune_qkv.py, profiles/r02_qkv_stamps.txt): a CU sustains ~27 GB/s, so the
// block's 64 KiB need ~2.4 us from the moment they are requested -- the launch is as long as [request -> data] + dots + atomic drain, and
// everything else (the prologue's dependent loads, norm, planes: done at ~2.8 us) hides under it.
template <int FIX, int NJ, int ACT, int DIAG = 0>     // NJ = H / 2048
__global__ __launch_bounds__(768) void k_gemv_q4g_slim(const uint4* __restrict__ W, const __half* __restrict__ S, const unsigned char* __restrict__ Z,
                                                      const float* __restrict__ bias, int N, int H, Pro pro, long long* acc, long long* zero_buf, int zero_n) {
  __shared__ uint4 xpl[8 * XQ_NP];
  __shared__ int4 gpar[4];
  __shared__ double red[4];
  __shared__ unsigned cntv[2];
  volatile unsigned* cnt = cntv;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool prolog = wave < 4;
  const int NKS = H >> 8, G = H >> 7;
  const int ksl = blockIdx.x % NKS, tg = blockIdx.x / NKS;
  asm volatile("" :: "s"(zero_buf), "s"(zero_n), "s"(acc), "s"(pro.h_in), "s"(pro.src.p), "s"(pro.norm_w), "s"(pro.h_out), "s"(pro.H),
               "s"(H), "s"(N), "s"(W), "s"(S), "s"(Z), "s"(bias));   // one scalar-load batch for all arguments
  unsigned long long T[8];
#define SSTAMP(i) do { if (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T[i]) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } } while (0)
#define SFLUSH() do { if (DIAG && pro.stamps && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 97)) { \
    _Pragma("unroll") for (int i = 0; i < 8; i++) pro.stamps[((blockIdx.x ? 1 : 0) * 12 + wave) * 8 + i] = (long long)T[i]; } } while (0)
#pragma unroll
  for (int i = 0; i < 8; i++) T[i] = 0;
  SSTAMP(0);
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  if (prolog) {
    // (1p) the row: octet j * 256 + pt of thread pt
    const int pt = tid;
    const bool hasprev = pro.src.p != nullptr;
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
    float4 ha[NJ], hb[NJ];
    typename RawT<FIX>::T pv[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int i0 = (j * 256 + pt) * 8;
      ha[j] = *(const float4*)(pro.h_in + i0); hb[j] = *(const float4*)(pro.h_in + i0 + 4);
#pragma unroll
      for (int e = 0; e < 8; e++) pv[j][e] = vraw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
    }
    const int osl = 32 * ksl, jsel = osl >> 8, p0 = osl & 255;              // the slice's 32 octets: register jsel of threads p0 .. p0 + 31
    const bool mine = pt >= p0 && pt < p0 + 32;
    const int so = min(max(pt - p0, 0), 31);                                // this thread's octet of the slice (clamped: loads are unconditional)
    const float4 na = *(const float4*)(pro.norm_w + (osl + so) * 8), nb = *(const float4*)(pro.norm_w + (osl + so) * 8 + 4);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // counters zeroed; the row's loads are all in the CU's queue AHEAD of the tile waves' weight loads
    __builtin_amdgcn_sched_barrier(0);
    SSTAMP(1);
    // (2p) h' = R(h + R(prev)), exact sum of squares
    float v[NJ][8];
    double ssd = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const float t[8] = {ha[j].x, ha[j].y, ha[j].z, ha[j].w, hb[j].x, hb[j].y, hb[j].z, hb[j].w};
#pragma unroll
      for (int e = 0; e < 8; e++) v[j][e] = hasprev ? round_t<ACT>(t[e] + round_t<ACT>(FIX ? fix2f((long long)pv[j][e], ACT) : (float)pv[j][e])) : t[e];
#pragma unroll
      for (int e = 0; e < 8; e += 2) ssd += (double)(v[j][e] * v[j][e]) + (double)(v[j][e + 1] * v[j][e + 1]);
      if (blockIdx.x == 0 && pro.h_out) {
        const int i0 = (j * 256 + pt) * 8;
        *(float4*)(pro.h_out + i0) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[j][4], v[j][5], v[j][6], v[j][7]);
      }
    }
    ssd = wave_sum_d(ssd);
    SSTAMP(2);
    if (lane == 0) { red[wave] = ssd; __builtin_amdgcn_s_waitcnt(0xc07f); atomicAdd((unsigned*)&cnt[0], 1u); }
    if (p0 / 64 == wave) {               // only the wave that holds the slice goes on (wave-uniform)
      lds_wait_count(&cnt[0], 4);
      SSTAMP(3);
      const float ss = (float)((red[0] + red[1]) + (red[2] + red[3]));
      const float rs = rms_scale(ss, (float)H, pro.eps);
      float sv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        sv[e] = v[0][e];
#pragma unroll
        for (int j = 1; j < NJ; j++) sv[e] = (j == jsel) ? v[j][e] : sv[e];
      }
      const float nwv[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
      float x[8];
      float am = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) { x[e] = mine ? round_t<ACT>(nwv[e] * round_t<ACT>(sv[e] * rs)) : 0.f; am = fmaxf(am, fabsf(x[e])); }
      am = grp_reduce<16, OpMax>(am);    // p0 is a multiple of 32: a 128-k group is an aligned run of 16 lanes, the other half wave carries zeros
      unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
      xq_split8(x, am, w, sp, cs);
#pragma unroll
      for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<16, OpAdd>(sp[p]);
      const int4 g2w = xq_gpar_hi<16>(w, sp);
      if (mine) {
        unsigned* plw = (unsigned*)xpl + ((so >> 2) * XQ_NP) * 4 + (so & 3);
#pragma unroll
        for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w[p];
        if ((so & 15) == 0) {
          gpar[2 * (so >> 4)] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
          gpar[2 * (so >> 4) + 1] = g2w;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (lane == 0) atomicAdd((unsigned*)&cnt[1], 1u);
      SSTAMP(4);
    }
    SFLUSH();
    return;
  }
  // (1s) tile waves: the rendezvous, then the tile's weights
  const int tq = tg * 8 + (wave - 4);
  const bool q_on = tq * 64 < N;
  const int tqc = q_on ? tq : 0;
  const uint4* wq = W + ((size_t)tqc * (H >> 5) + ksl * 8) * 64 + lane;
  uint4 Q[8]; float sq[2]; int zq[2];
  __syncthreads();                       // (the row waves' loads go first: the CU issues ~64 B per clock, 120 KiB of requests take ~0.9 us)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 8; c++) Q[c] = ldnt(wq + c * 64);
#pragma unroll
  for (int b = 0; b < 2; b++) { const size_t ix = ((size_t)tqc * G + ksl * 2 + b) * 64 + lane; sq[b] = __half2float(S[ix]); zq[b] = Z[ix]; }
  SSTAMP(1);
  if (zero_buf)
    for (int i = blockIdx.x * 512 + (tid - 256); i < zero_n; i += gridDim.x * 512) zero_buf[i] = 0;
  lds_wait_count(&cnt[1], 1);
  SSTAMP(4);
  if (q_on) {
    double y = 0.0;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      if (b == 0) q4g_consume_at<0, 4>(Q, 0, 0, xpl, gpar, sq[0], zq[0], y);
      else q4g_consume_at<4, 4>(Q, 4, 2, xpl, gpar, sq[1], zq[1], y);
      if (b == 0) SSTAMP(5);
    }
    const int n = tq * 64 + lane;
    if (bias != nullptr && ksl == 0) y += (double)bias[n];
    atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(y, ACT));
  }
  SSTAMP(6);
  if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  SSTAMP(7);
  SFLUSH();
#undef SSTAMP
#undef SFLUSH
}

// ---------------------------------------------------------------------------------------------------------
// int4 GEMV with the fused residual + RMSNorm prologue, FULL K per workgroup, direct output (the q/k/v projection of the decode step).
// The slim kernel above splits K over 16 workgroups per tile group: 49 k 64-bit atomics whose drain the launch waits for (~1.5 us), and a
// consumer that reads 8-byte fixed point.  Here a workgroup owns ONE 64-column tile over the whole K: its NW waves take 256 k each
// (8 KiB of weights per wave), meet through LDS, and the block stores 64 finished values -- no atomics, nothing to zero, the attention
// kernel reads plain f32.  N / 64 workgroups (96 for Llama-3-8B) leave most CUs idle, but the launch is a latency chain, not a stream:
// 13 MB over 96 CUs is ~2 us, hidden under the prologue's dependent loads.  Same wave roles as the fused MLP: the first half requests its
// weights at entry, the second half holds the row (an octet per thread), does the norm and publishes the nibble planes.
// ---------------------------------------------------------------------------------------------------------
template <int FIX, int NW, int ACT>     // H = NW * 256
__global__ __launch_bounds__(NW * 64) void k_gemv_q4g_cols(const uint4* __restrict__ W, const __half* __restrict__ S, const unsigned char* __restrict__ Z,
                                                          const float* __restrict__ bias, int N, int H, Pro pro, float* __restrict__ out, long long* zero_buf, int zero_n) {
  constexpr int NP = NW / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* xpl = (uint4*)smem;                       // [H/32 chunks][6 planes]
  const int G = H >> 7;
  int4* gpar = (int4*)(xpl + (H >> 5) * XQ_NP);    // [2G]
  double* part = (double*)(gpar + 2 * G);          // [NW][64]
  double* dred = part + NW * 64;                   // [NP]
  volatile unsigned* cnt = (volatile unsigned*)(dred + NW);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool prolog = wave >= NP;
  const int tile = blockIdx.x;
  asm volatile("" :: "s"(zero_buf), "s"(zero_n), "s"(out), "s"(pro.h_in), "s"(pro.src.p), "s"(pro.norm_w), "s"(pro.h_out), "s"(H), "s"(N), "s"(W), "s"(S), "s"(Z), "s"(bias));
  const uint4* wq = W + ((size_t)tile * (H >> 5) + wave * 8) * 64 + lane;
  uint4 Q[8];
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  if (prolog) {
    const bool hasprev = pro.src.p != nullptr;
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
    const int i0 = (tid - NP * 64) * 8;
    const float4 ha = *(const float4*)(pro.h_in + i0), hb = *(const float4*)(pro.h_in + i0 + 4);
    typename RawT<FIX>::T pv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) pv[e] = vraw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
    const float4 na = *(const float4*)(pro.norm_w + i0), nb = *(const float4*)(pro.norm_w + i0 + 4);
    __builtin_amdgcn_sched_barrier(0);
    Q[0] = ldnt(wq); Q[1] = ldnt(wq + 64);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // the counters are zero (nobody waits for data here)
    float v[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    if (hasprev) {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = round_t<ACT>(v[e] + round_t<ACT>(FIX ? fix2f((long long)pv[e], ACT) : (float)pv[e]));
    }
    if (blockIdx.x == 0 && pro.h_out) { *(float4*)(pro.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
    double ssd = 0.0;
#pragma unroll
    for (int e = 0; e < 8; e += 2) ssd += (double)(v[e] * v[e]) + (double)(v[e + 1] * v[e + 1]);
    ssd = wave_sum_d(ssd);
    if (lane == 0) { dred[wave - NP] = ssd; __builtin_amdgcn_s_waitcnt(0xc07f); atomicAdd((unsigned*)&cnt[0], 1u); }
    lds_wait_count(&cnt[0], NP);
    ssd = ((dred[0] + dred[1]) + (dred[2] + dred[3]));
    if (NP == 8) ssd += ((dred[4] + dred[5]) + (dred[6] + dred[7]));
    const float rs = rms_scale((float)ssd, (float)H, pro.eps);
    const float nwv[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
    float x[8];
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { x[e] = round_t<ACT>(nwv[e] * round_t<ACT>(v[e] * rs)); am = fmaxf(am, fabsf(x[e])); }
    am = grp_reduce<16, OpMax>(am);
    unsigned w[XQ_NP]; int sp[XQ_NP]; float cs;
    xq_split8(x, am, w, sp, cs);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) sp[p] = grp_reduce<16, OpAdd>(sp[p]);
    const int4 g2w = xq_gpar_hi<16>(w, sp);
    const int oct = tid - NP * 64;
    unsigned* plw = (unsigned*)xpl + ((oct >> 2) * XQ_NP) * 4 + (oct & 3);
#pragma unroll
    for (int p = 0; p < XQ_NP; p++) plw[p * 4] = w[p];
    if ((lane & 15) == 0) {
      gpar[2 * (oct >> 4)] = make_int4(__float_as_int(cs), sp[0], sp[1], sp[2]);
      gpar[2 * (oct >> 4) + 1] = g2w;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) atomicAdd((unsigned*)&cnt[1], 1u);
#pragma unroll
    for (int c = 2; c < 8; c++) Q[c] = ldnt(wq + c * 64);
  } else {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; c++) Q[c] = ldnt(wq + c * 64);
    if (zero_buf)
      for (int i = blockIdx.x * (NP * 64) + tid; i < zero_n; i += gridDim.x * (NP * 64)) zero_buf[i] = 0;
  }
  float sq[2]; int zq[2];
#pragma unroll
  for (int b = 0; b < 2; b++) { const size_t ix = ((size_t)tile * G + wave * 2 + b) * 64 + lane; sq[b] = __half2float(S[ix]); zq[b] = Z[ix]; }
  lds_wait_count(&cnt[1], NP);
  double y = 0.0;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    if (b == 0) q4g_consume_at<0, 4>(Q, (wave * 2) * 4, 2 * (wave * 2), xpl, gpar, sq[0], zq[0], y);
    else q4g_consume_at<4, 4>(Q, (wave * 2 + 1) * 4, 2 * (wave * 2 + 1), xpl, gpar, sq[1], zq[1], y);
  }
  part[wave * 64 + lane] = y;
  __syncthreads();
  if (tid < 64) {
    double t = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) t += part[w2 * 64 + tid];
    float f = (float)t;                      // ONE rounding of the exact dot product (the oracle's definition)
    const int n = tile * 64 + tid;
    if (bias) f += bias[n];
    out[n] = round_t<ACT>(f);
  }
}
static size_t q4g_cols_smem(int H) { return (size_t)(H >> 5) * XQ_NP * 16 + (size_t)(H >> 7) * 32 + 16 * 64 * 8 + 16 * 8 + 64; }
bool bzk_gemv_cols_ok(const LinearDev& L, const Pro& pro, int act) {
  static const bool off = getenv("BZ_COLS_QKV") == nullptr;   // opt-in: measured slower than the slim kernel (7.8 vs 6.8 us: a CU sustains ~27 GB/s, 96 CUs x 128 KiB take longer than 192 x 64 KiB)
  return !off && act == BZ_F16 && L.kind == LK_Q4G && !L.perm && pro.mode == PRO_NORM && pro.perm == nullptr && L.K == pro.H && (L.K == 2048 || L.K == 4096) && L.N % 64 == 0 &&
         L.N <= 16384 && !pro.dbg && !pro.stamps;
}
int bzk_gemv_cols(hipStream_t s, const LinearDev& L, const Pro& pro, float* out, long long* zero_buf, int zero_n) {
  const size_t smem = q4g_cols_smem(L.K);
#define LAUNCH_COLS(FIX, NW_) BZ_LAUNCH("gemv_q4g<norm,cols>", L.algo_bytes, (k_gemv_q4g_cols<FIX, NW_, BZ_F16>), dim3(L.N / 64), dim3(NW_ * 64), smem, s, (const uint4*)L.w, \
    (const __half*)L.scales, (const unsigned char*)L.zeros, L.bias, L.N, L.K, pro, out, zero_buf, zero_n)
  if (L.K == 4096) { if (pro.src.fix) LAUNCH_COLS(1, 16); else LAUNCH_COLS(0, 16); }
  else { if (pro.src.fix) LAUNCH_COLS(1, 8); else LAUNCH_COLS(0, 8); }
#undef LAUNCH_COLS
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Multi-row int4 GEMM (rows = prompt tokens of a prefill chunk, or the sequences of a decode batch): the weights of a (tile, 256-k slice)
// are loaded ONCE per wave and consumed for up to 8 activation rows, so a prompt costs one pass over the weights per 8 tokens instead of
// one per token.  Same arithmetic as the GEMV (three int8 planes per 128-group, exact int32 group sums, 64-bit fixed-point split-K), so a
// row's result is bit-identical to what the single-row kernels produce for it.
//   X: 16-bit rows [rows][K] (the prefill pipeline's GEMM input), acc: fixed point [rows][N], zero on entry.
// grid = (K/256) x ceil(N/512), 512 threads: 8 waves share the slice (8 x 256 activations quantised per block), one 64-column tile each.
// ---------------------------------------------------------------------------------------------------------
template <int XDT>
__global__ __launch_bounds__(512) void k_gemm_q4g_rows(const uint4* __restrict__ W, const __half* __restrict__ S, const unsigned char* __restrict__ Z,
                                                      const float* __restrict__ bias, int N, int K, const unsigned short* __restrict__ X, int rows,
                                                      long long* __restrict__ acc) {
  __shared__ __attribute__((aligned(16))) float xs[8 * 256];
  __shared__ uint4 xpl[8 * 8 * XQ_NP];
  __shared__ int4 gpar[8 * 4];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int NKS = K >> 8, G = K >> 7;
  const int ksl = blockIdx.x % NKS, tq = (blockIdx.x / NKS) * 8 + wave;
  const bool q_on = tq * 64 < N;
  const int tqc = q_on ? tq : 0;
  // activations of this slice: thread t -> row t / 64, 4 elements at (t % 64) * 4 (one 8-byte load)
  const int xr = tid >> 6;
  const uint2 xraw = *(const uint2*)(X + (size_t)min(xr, rows - 1) * K + ksl * 256 + lane * 4);
  __builtin_amdgcn_sched_barrier(0);
  uint4 Q[2][4]; float sq[2]; int zq[2];
  {
    const uint4* wq = W + ((size_t)tqc * (K >> 5) + ksl * 8) * 64 + lane;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int c = 0; c < 4; c++) Q[b][c] = ldnt(wq + (b * 4 + c) * 64);
#pragma unroll
    for (int b = 0; b < 2; b++) { const size_t ix = ((size_t)tqc * G + ksl * 2 + b) * 64 + lane; sq[b] = __half2float(S[ix]); zq[b] = Z[ix]; }
  }
  {
    const unsigned u[2] = {xraw.x, xraw.y};
    float v[4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (XDT == BZ_F16) { v[2 * i] = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu))); v[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16))); }
      else { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
    }
    const bool on = xr < rows;
    *(float4*)(xs + xr * 256 + lane * 4) = on ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  quant_x128<512>(xs, 8 * 256, xpl, gpar);              // 16 groups of 128: group index = row * 2 + (group of the slice)
  __syncthreads();
  if (!q_on) return;
  const int n = tq * 64 + lane;
  const double bv = (bias != nullptr && ksl == 0) ? (double)bias[n] : 0.0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    if (r < rows) {
      double y = bv;
      q4g_consume(Q[0], r * 2, xpl, gpar, sq[0], zq[0], y);
      q4g_consume(Q[1], r * 2 + 1, xpl, gpar, sq[1], zq[1], y);
      atomicAdd((unsigned long long*)(acc + (size_t)r * N + n), (unsigned long long)d2fix(y, XDT));
    }
  }
}

// out[i] = R(fix2f(acc[i])), acc[i] = 0  (the accumulator is ready for the next GEMM)
__global__ void k_fix_rows_finish(long long* acc, size_t n, int act, float* out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { out[i] = round_act(fix2f(acc[i], act), act); acc[i] = 0; }
}

bool bzk_gemm_q4g_rows_ok(const LinearDev& L) { return L.kind == LK_Q4G && !L.perm && L.K % 256 == 0 && L.N % 64 == 0; }

// Y[rows][N] (f32, rounded to act) = X16[rows][K] . W^T for any number of rows, 8 at a time; `acc` is scratch of 8 * N fixed-point values (zero on entry and on exit)
int bzk_gemm_q4g_rows(hipStream_t s, const LinearDev& L, int xdt, const void* x16, int rows, int act, long long* acc, float* y) {
  if (!bzk_gemm_q4g_rows_ok(L) || (xdt != BZ_F16 && xdt != BZ_BF16) || xdt != act) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_q4g_rows: unsupported weight / activation format");
  const int nks = L.K / 256, ntg = (L.N / 64 + 7) / 8;
  for (int r0 = 0; r0 < rows; r0 += 8) {
    const int nr = std::min(8, rows - r0);
    const unsigned short* xp = (const unsigned short*)x16 + (size_t)r0 * L.K;
    const double bytes = (double)L.algo_bytes;
    if (xdt == BZ_F16) BZ_LAUNCH("gemm_q4g_rows", bytes, k_gemm_q4g_rows<BZ_F16>, dim3(nks * ntg), dim3(512), 0, s, (const uint4*)L.w, (const __half*)L.scales,
                                 (const unsigned char*)L.zeros, L.bias, L.N, L.K, xp, nr, acc);
    else BZ_LAUNCH("gemm_q4g_rows", bytes, k_gemm_q4g_rows<BZ_BF16>, dim3(nks * ntg), dim3(512), 0, s, (const uint4*)L.w, (const __half*)L.scales,
                   (const unsigned char*)L.zeros, L.bias, L.N, L.K, xp, nr, acc);
    hipLaunchKernelGGL(k_fix_rows_finish, dim3(std::min<size_t>(1024, ((size_t)nr * L.N + 255) / 256)), dim3(256), 0, s, acc, (size_t)nr * L.N, act, y + (size_t)r0 * L.N);
  }
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

bool bzk_gemv_slim_ok(const LinearDev& L, const Pro& pro) {
  static const bool off = getenv("BZ_NO_SLIM_QKV") != nullptr;
  return !off && L.kind == LK_Q4G && !L.perm && pro.mode == PRO_NORM && pro.perm == nullptr && L.K == pro.H && (L.K == 2048 || L.K == 4096 || L.K == 8192) && L.N % 64 == 0 &&
         L.N <= 16384 && !pro.dbg;
}

// =========================================================================================================
// GGUF block-quantised GEMV: Q8_0 / Q4_K / Q6_K  (formats = public GGML block layouts, gguf.rs:33 hands
// the blocks over opaquely).  Same skeleton as k_gemv_q4g -- lane == output column, split-K over k-slices, fused
// prologues, 3-plane int8 activations on V_DOT4_I32_I8, fixed-point atomics -- with the activation slice scaled per
// 32-k chunk so that every format's sub-block structure (32 for Q8_0/Q4_K, 16 for Q6_K) falls on chunk boundaries.
//
// HBM layouts after the load-time repack (bytes per weight unchanged: 34/32, 144/256, 210/256):
//   Q8_0 : q  [N/64][K/16][64][16 B] int8,                      d  [N/64][K/32][64] f16
//   Q4_K : q  [N/64][K/32][64][16 B] nibbles (A/B order as Q4G), hdr [N/64][K/256][64][16 B] = {d, dmin, scales[12]} verbatim
//   Q6_K : ql [N/64][K/32][64][16 B] low nibbles (A/B order),    qh [N/64][K/32][64][8 B] 2-bit highs,
//          sc [N/64][K/256][64][16 B] int8 x16,                  d  [N/64][K/256][64] f16
// per-chunk activation parameters in LDS (2 x int4 per chunk):
//   P0 = { sx (f32 bits), Sa_h, Sa_m, Sa_l }   P1 = { Sb_h, Sb_m, Sb_l, 0 }
//   Q4_K: Sa = sum over the chunk, Sb = sum over k%8 >= 4 (for the signed high-nibble trick);  Q6_K: Sa / Sb = sums over the
//   first / second 16;  Q8_0: unused.
// =========================================================================================================
enum { GQ_Q80 = 0, GQ_Q4K = 1, GQ_Q6K = 2, GQ_MIX = 3 };   // MIX: Q4_K tiles followed by Q6_K tiles in one launch (the Q4_K_M q/k + v split)
#define XQ_MAX 8355000.0f  // < 127*65536 + 127*256 + 127: the largest magnitude of three balanced int8 planes

// one thread's octet (8 consecutive k of a 32-k chunk; the chunk's four octets sit in four consecutive lanes, quad-aligned) -> three int8 planes
// + the chunk parameters.  e0 = index of the octet's first element in the slice; `on`: this thread takes part (its lanes' reductions run anyway)
template <int FMT>
__device__ __forceinline__ void quant8_x32(const float (&v)[8], int e0, bool on, int lane, unsigned* xh, unsigned* xm, unsigned* xl, int4* cpar, int4* cpar6 = nullptr) {
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) am = fmaxf(am, fabsf(v[i]));
  am = grp_reduce<4, OpMax>(am);   // 4 lanes x 8 = one 32-k chunk
  const float inv = am > 0.f ? XQ_MAX / am : 0.f;
  unsigned wh[2] = {0, 0}, wm[2] = {0, 0}, wl[2] = {0, 0};
  int sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};     // Q4_K: sums over the chunk / over k % 8 >= 4 ; Q6_K (and the second set of MIX): first / second 16
  int ta[3] = {0, 0, 0}, tb[3] = {0, 0, 0};
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
    if (FMT == GQ_Q4K || FMT == GQ_MIX) { sa[0] += hi; sa[1] += mid; sa[2] += lo; if (i >= 4) { sb[0] += hi; sb[1] += mid; sb[2] += lo; } }
    if (FMT == GQ_Q6K) { sa[0] += hi; sa[1] += mid; sa[2] += lo; }   // per-thread sum; halves are separated below
  }
  if (FMT == GQ_Q6K || FMT == GQ_MIX) {
    // lanes 0,1 of the 4-lane group hold k 0..15 (first half), lanes 2,3 hold k 16..31
    const bool second = (lane & 2) != 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int mine = sa[q] + dpp_get<DPP_XOR1>(sa[q]);          // sum of my half (MIX: sa is still the per-thread sum here)
      const int other = dpp_get<DPP_XOR2>(mine);                  // the other half
      ta[q] = second ? other : mine;                              // first half
      tb[q] = second ? mine : other;                              // second half
    }
    if (FMT == GQ_Q6K) {
#pragma unroll
      for (int q = 0; q < 3; q++) { sa[q] = ta[q]; sb[q] = tb[q]; }
    }
  }
  if (FMT == GQ_Q4K || FMT == GQ_MIX) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      sa[q] = grp_reduce<4, OpAdd>(sa[q]);
      sb[q] = grp_reduce<4, OpAdd>(sb[q]);
    }
  }
  if (on) {
    *(uint2*)(xh + e0 / 4) = make_uint2(wh[0], wh[1]);
    *(uint2*)(xm + e0 / 4) = make_uint2(wm[0], wm[1]);
    *(uint2*)(xl + e0 / 4) = make_uint2(wl[0], wl[1]);
    if ((lane & 3) == 0) {
      cpar[2 * (e0 >> 5)] = make_int4(__float_as_int(am * (1.0f / XQ_MAX)), sa[0], sa[1], sa[2]);
      cpar[2 * (e0 >> 5) + 1] = make_int4(sb[0], sb[1], sb[2], 0);
      if (FMT == GQ_MIX) {
        cpar6[2 * (e0 >> 5)] = make_int4(__float_as_int(am * (1.0f / XQ_MAX)), ta[0], ta[1], ta[2]);
        cpar6[2 * (e0 >> 5) + 1] = make_int4(tb[0], tb[1], tb[2], 0);
      }
    }
  }
}
template <int FMT>
__device__ __forceinline__ void quant_x32(const float* xs, int KR, unsigned* xh, unsigned* xm, unsigned* xl, int4* cpar, int4* cpar6 = nullptr) {
  for (int base = 0; base < KR; base += 256 * 8) {
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
    quant8_x32<FMT>(v, e0, on, (int)threadIdx.x, xh, xm, xl, cpar, cpar6);
  }
}

__device__ __forceinline__ float planes_f(int uh, int um, int ul) { return fmaf((float)uh, 65536.0f, fmaf((float)um, 256.0f, (float)ul)); }

// one 32-k chunk of signed / unsigned int8 weights given as 8 words (4 weights each, k order), against the chunk's planes
__device__ __forceinline__ void dot_chunk8(const unsigned (&w)[8], const uint4* xh4, const uint4* xm4, const uint4* xl4, int c, int (&u)[3]) {
  const uint4 h0 = xh4[c * 2], h1 = xh4[c * 2 + 1], m0 = xm4[c * 2], m1 = xm4[c * 2 + 1], l0 = xl4[c * 2], l1 = xl4[c * 2 + 1];
  const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
  const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u[0] = __builtin_amdgcn_sdot4((int)w[j], (int)Xh[j], u[0], false);
    u[1] = __builtin_amdgcn_sdot4((int)w[j], (int)Xm[j], u[1], false);
    u[2] = __builtin_amdgcn_sdot4((int)w[j], (int)Xl[j], u[2], false);
  }
}

// Q4_K 6-bit scale / min of sub-block j from the 12 packed bytes (GGML get_scale_min_k4)
__device__ __forceinline__ void q4k_scale_min(const unsigned (&hw)[4], int j, int& sc, int& mn) {
  // bytes: hw[0] = d,dmin ; scales bytes s[0..11] = bytes 4..15 of the 16-byte header
  auto sbyte = [&](int i) -> unsigned { const int bi = 4 + i; return (hw[bi >> 2] >> (8 * (bi & 3))) & 255u; };
  if (j < 4) { sc = (int)(sbyte(j) & 63u); mn = (int)(sbyte(j + 4) & 63u); }
  else { sc = (int)((sbyte(j + 4) & 15u) | ((sbyte(j - 4) >> 6) << 4)); mn = (int)((sbyte(j + 4) >> 4) | ((sbyte(j) >> 6) << 4)); }
}

#define GQ_E 8   // slice elements per thread (KR <= 2048)

template <int FMT, int MODE, int FIX, int MAXJ>
__global__ __launch_bounds__(256) void k_gemv_gq(const uint4* __restrict__ Wq, const uint2* __restrict__ Wh, const uint4* __restrict__ Hd,
                                                 const __half* __restrict__ Dd, const float* __restrict__ bias, int N, int K, int SBW, int nst,
                                                 Pro pro, long long* acc, long long* zero_buf, int zero_n) {
  // SBW = 256-k superblocks per workgroup (k-slice of 256 * SBW)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KR = SBW * 256;
  float* xs = (float*)smem;                 // [KR]
  unsigned* xh = (unsigned*)(xs + KR);      // [KR/4] x3
  unsigned* xm = xh + KR / 4;
  unsigned* xl = xm + KR / 4;
  int4* cpar = (int4*)(xl + KR / 4);        // [2 * KR/32]
  float* red = (float*)(cpar + 2 * (KR / 32));
  const int st = blockIdx.x % nst, ks = blockIdx.x / nst;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nt = st * 4 + wave;
  const bool wave_on = nt * 64 < N;
  const int ntc = wave_on ? nt : 0;
  const int k0 = ks * KR;
  zero_duty<256>(zero_buf, zero_n);

  XRegs<MODE, FIX, MAXJ, GQ_E> xr;
  xload<MODE, FIX, MAXJ, GQ_E>(pro, k0, KR, xr);
  __builtin_amdgcn_sched_barrier(0);
  // first superblock's weights
  const int SB = K >> 8, C32 = K >> 5;
  const int sb0 = ks * SBW;
  uint4 q[16];          // Q8_0: 16 x 16 B (256 int8) ; Q4_K / Q6_K: 8 x 16 B nibbles
  uint2 qh[8];          // Q6_K highs
  uint4 hd = make_uint4(0, 0, 0, 0);
  __half dv[8];         // Q8_0: the 8 block scales of the superblock ; Q6_K: dv[0] = d
  auto load_sb = [&](int sb) {
    if (FMT == GQ_Q80) {
      const uint4* p = Wq + ((size_t)ntc * (K >> 4) + (size_t)sb * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; i++) q[i] = ldnt(p + i * 64);
#pragma unroll
      for (int c = 0; c < 8; c++) dv[c] = Dd[((size_t)ntc * C32 + (size_t)sb * 8 + c) * 64 + lane];
    } else {
      const uint4* p = Wq + ((size_t)ntc * C32 + (size_t)sb * 8) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = ldnt(p + i * 64);
      hd = Hd[((size_t)ntc * SB + sb) * 64 + lane];
      if (FMT == GQ_Q6K) {
        const uint2* ph = Wh + ((size_t)ntc * C32 + (size_t)sb * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; i++) qh[i] = ph[i * 64];
        dv[0] = Dd[((size_t)ntc * SB + sb) * 64 + lane];
      }
    }
  };
  load_sb(sb0);
  xfinish<MODE, FIX, MAXJ, GQ_E>(pro, KR, xr, xs, red, blockIdx.x == 0);
  quant_x32<FMT>(xs, KR, xh, xm, xl, cpar);
  __syncthreads();
  if (!wave_on) return;

  const uint4* xh4 = (const uint4*)xh;
  const uint4* xm4 = (const uint4*)xm;
  const uint4* xl4 = (const uint4*)xl;
  float y = 0.f;
  for (int s = 0; s < SBW; s++) {
    const int sb = sb0 + s;
    (void)sb;
    if (FMT == GQ_Q80) {
#pragma unroll
      for (int c = 0; c < 8; c++) {     // chunk = one Q8_0 block
        const unsigned w[8] = {q[2 * c].x, q[2 * c].y, q[2 * c].z, q[2 * c].w, q[2 * c + 1].x, q[2 * c + 1].y, q[2 * c + 1].z, q[2 * c + 1].w};
        int u[3] = {0, 0, 0};
        dot_chunk8(w, xh4, xm4, xl4, s * 8 + c, u);
        y += __half2float(dv[c]) * (__int_as_float(cpar[2 * (s * 8 + c)].x) * planes_f(u[0], u[1], u[2]));
      }
    } else if (FMT == GQ_Q4K) {
      const unsigned hw[4] = {hd.x, hd.y, hd.z, hd.w};
      const float d = __half2float(__ushort_as_half((unsigned short)(hw[0] & 0xffffu)));
      const float dmin = __half2float(__ushort_as_half((unsigned short)(hw[0] >> 16)));
#pragma unroll
      for (int c = 0; c < 8; c++) {     // chunk = one sub-block
        const unsigned ww[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
        // nibble trick of Q4G: low nibbles q (unsigned), high nibbles stored as (q - 8): words in k order are A0,B0,A1,B1,...
        unsigned w[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { w[2 * j] = ww[j] & 0x0F0F0F0Fu; w[2 * j + 1] = ww[j] & 0xF0F0F0F0u; }
        int ua[3] = {0, 0, 0}, ub[3] = {0, 0, 0};
        {
          const int cc = s * 8 + c;
          const uint4 h0 = xh4[cc * 2], h1 = xh4[cc * 2 + 1], m0 = xm4[cc * 2], m1 = xm4[cc * 2 + 1], l0 = xl4[cc * 2], l1 = xl4[cc * 2 + 1];
          const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
          const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
          const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            ua[0] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xh[2 * j], ua[0], false);
            ua[1] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xm[2 * j], ua[1], false);
            ua[2] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xl[2 * j], ua[2], false);
            ub[0] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xh[2 * j + 1], ub[0], false);
            ub[1] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xm[2 * j + 1], ub[1], false);
            ub[2] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xl[2 * j + 1], ub[2], false);
          }
        }
        const int4 p0 = cpar[2 * (s * 8 + c)], p1 = cpar[2 * (s * 8 + c) + 1];
        // 16 * sum q x  per plane = 16 A + B' + 128 Sb   (B' = sum 16 (q - 8) x)
        const float qx = planes_f((ua[0] << 4) + ub[0] + 128 * p1.x, (ua[1] << 4) + ub[1] + 128 * p1.y, (ua[2] << 4) + ub[2] + 128 * p1.z) * (1.0f / 16.0f);
        const float sx_ = planes_f(p0.y, p0.z, p0.w);
        int sc, mn;
        q4k_scale_min(hw, c, sc, mn);
        y += __int_as_float(p0.x) * ((d * (float)sc) * qx - (dmin * (float)mn) * sx_);
      }
    } else {  // Q6_K
      const unsigned sw[4] = {hd.x, hd.y, hd.z, hd.w};   // 16 int8 scales
      const float d = __half2float(dv[0]);
#pragma unroll
      for (int c = 0; c < 8; c++) {     // chunk = 32 k = two 16-blocks (words 0..3 and 4..7 in k order)
        const unsigned ww[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
        const unsigned hh[2] = {qh[c].x, qh[c].y};
        unsigned w[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned lo_a = ww[j] & 0x0F0F0F0Fu, lo_b = (ww[j] >> 4) & 0x0F0F0F0Fu;
          const int fa = 2 * j, fb = 2 * j + 1;      // 2-bit field index (k order words A_j, B_j)
          const unsigned hi_a = ((hh[fa >> 2] >> (2 * (fa & 3))) & 0x03030303u) << 4;
          const unsigned hi_b = ((hh[fb >> 2] >> (2 * (fb & 3))) & 0x03030303u) << 4;
          w[2 * j] = lo_a | hi_a;       // q in 0..63 (as positive int8); the -32 is applied through the plane sums
          w[2 * j + 1] = lo_b | hi_b;
        }
        const int cc = s * 8 + c;
        const uint4 h0 = xh4[cc * 2], h1 = xh4[cc * 2 + 1], m0 = xm4[cc * 2], m1 = xm4[cc * 2 + 1], l0 = xl4[cc * 2], l1 = xl4[cc * 2 + 1];
        const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        int u0[3] = {0, 0, 0}, u1[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          u0[0] = __builtin_amdgcn_sdot4((int)w[j], (int)Xh[j], u0[0], false);
          u0[1] = __builtin_amdgcn_sdot4((int)w[j], (int)Xm[j], u0[1], false);
          u0[2] = __builtin_amdgcn_sdot4((int)w[j], (int)Xl[j], u0[2], false);
          u1[0] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xh[4 + j], u1[0], false);
          u1[1] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xm[4 + j], u1[1], false);
          u1[2] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xl[4 + j], u1[2], false);
        }
        const int4 p0 = cpar[2 * cc], p1 = cpar[2 * cc + 1];
        const int s0 = (int)(signed char)((sw[(2 * c) >> 2] >> (8 * ((2 * c) & 3))) & 255u);
        const int s1 = (int)(signed char)((sw[(2 * c + 1) >> 2] >> (8 * ((2 * c + 1) & 3))) & 255u);
        const float f0 = planes_f(u0[0] - 32 * p0.y, u0[1] - 32 * p0.z, u0[2] - 32 * p0.w);
        const float f1 = planes_f(u1[0] - 32 * p1.x, u1[1] - 32 * p1.y, u1[2] - 32 * p1.z);
        y += __int_as_float(p0.x) * ((d * (float)s0) * f0 + (d * (float)s1) * f1);
      }
    }
    if (s + 1 < SBW) load_sb(sb + 1);
  }
  const int n = nt * 64 + lane;
  if (bias != nullptr && ks == 0) y += bias[n];
  atomicAdd((unsigned long long*)(acc + n), (unsigned long long)f2fix(y, pro.act));
}

// ---------------------------------------------------------------------------------------------------------
// Slim form of the GGUF block-quant GEMV (Q4_K / Q6_K): the structure of k_gemv_q4g_slim -- 8 waves share ONE 256-k slice, which here is
// exactly one superblock, each wave owns one 64-column tile (its 8 KiB of nibbles + header in flight before the prologue) -- with the
// per-32-k-chunk activation planes and the block arithmetic of k_gemv_gq.  The generic kernel holds > 256 registers per lane (one
// workgroup of 4 waves per CU) and runs q/k/v in two rounds: 18.2 us against 7.0 us for the same bytes in AWQ form.
//   MODE NORM: x = RMSNorm(h + prev) slice (full-H sum of squares per block)      MODE SILU: x = R(R(silu(gate)) * up) slice
// grid = (K / 256) x ceil(N / 512), 512 threads
// ---------------------------------------------------------------------------------------------------------
// (the balanced role form of k_gemv_rows2 on these layouts -- 256 workgroups of 4 row + 8 tile waves, units = (superblock slice, tile), 512 activations quantised
// once per workgroup, Q4_K one unit ahead -- was built, is parity-clean, and is slower on every Mistral-7B launch: gate/up / down 15.1 vs 13.9 us, Q6_K 15.6 vs 13.4,
// q/k/v mix 10.3 vs 9.5: eight tile waves per CU cannot keep the dot / epilogue chains of this arithmetic busy; the slim kernel's 16 computing waves per CU can.)
// (several tiles per wave -- one norm prologue per ~256 workgroups instead of 896 on gate/up, the next tile's superblock prefetched under the current dots --
// was built and measured too: 132 / 169 registers instead of 77-115, one workgroup per CU, 527 vs 539 tok/s.  The kernel is bound by its dot issue, not its prologues.)
// (a 12-wave role-split form of this kernel, as in k_gemv_q4g_slim, was built and measured on the Mistral-7B Q4_K_M shape: 539 vs 549 tok/s -- no gain; the f32
// activations of the GGUF path make its prologue lighter (no 64-bit residual reads), and its big launches want many small workgroups in flight)
template <int FMT, int MODE, int FIX, int NJ>     // NJ = H / 2048 (NORM only)
__global__ __launch_bounds__(512) void k_gemv_gq_slim(const uint4* __restrict__ Wq, const uint2* __restrict__ Wh, const uint4* __restrict__ Hd,
                                                     const __half* __restrict__ Dd, const float* __restrict__ bias, int N, int K, Pro pro, long long* acc,
                                                     long long* zero_buf, int zero_n, GqMix mix) {
  __shared__ __attribute__((aligned(16))) float xs[256];
  __shared__ __attribute__((aligned(16))) unsigned xh[64], xm[64], xl[64];
  __shared__ int4 cpar[16];
  __shared__ int4 cpar6[FMT == GQ_MIX ? 16 : 1];
  __shared__ float red[8];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int SB = K >> 8, C32 = K >> 5;
  const int ksl = blockIdx.x % SB, tq = (blockIdx.x / SB) * 8 + wave;
  const bool q_on = tq * 64 < N;
  // MIX: tiles [0, mix.nt4) are Q4_K (the kernel's primary pointers), the rest Q6_K (mix pointers, tile index restarting at 0); wave-uniform
  const bool is6 = FMT == GQ_Q6K || (FMT == GQ_MIX && q_on && tq >= mix.nt4);
  const int tqc = q_on ? (FMT == GQ_MIX && is6 ? tq - mix.nt4 : tq) : 0;
  const uint4* Wq_ = (FMT == GQ_MIX && is6) ? mix.Wq6 : Wq;
  const uint4* Hd_ = (FMT == GQ_MIX && is6) ? mix.Hd6 : Hd;
  const uint2* Wh_ = FMT == GQ_MIX ? mix.Wh6 : Wh;
  const __half* Dd_ = FMT == GQ_MIX ? mix.Dd6 : Dd;
  asm volatile("" :: "s"(zero_buf), "s"(zero_n), "s"(acc), "s"(pro.h_in), "s"(pro.src.p), "s"(pro.norm_w), "s"(pro.h_out), "s"(pro.H),
               "s"(pro.act), "s"(K), "s"(N), "s"(Wq), "s"(Wh), "s"(Hd), "s"(Dd), "s"(bias));   // one scalar-load batch for all arguments
  zero_duty<512>(zero_buf, zero_n);
  // (1) prologue loads
  const bool hasprev = pro.src.p != nullptr;
  float hv[NJ][4];
  typename RawT<FIX>::T pv[NJ][4];
  float4 nw = make_float4(0, 0, 0, 0);
  typename RawT<FIX>::T ga = 0, ua = 0;
  if (MODE == PRO_NORM) {
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int i = j * 2048 + tid * 4;
      const float4 h4 = *(const float4*)(pro.h_in + i);
      hv[j][0] = h4.x; hv[j][1] = h4.y; hv[j][2] = h4.z; hv[j][3] = h4.w;
#pragma unroll
      for (int e = 0; e < 4; e++) pv[j][e] = vraw<FIX>(prevp, (FIX || hasprev) ? i + e : 0);
    }
    nw = *(const float4*)(pro.norm_w + ksl * 256 + (tid & 63) * 4);
  } else {
    const int kk = ksl * 256 + (tid & 255);
    ga = vraw<FIX>(pro.src.p, kk); ua = vraw<FIX>(pro.src.p, pro.H + kk);
  }
  __builtin_amdgcn_sched_barrier(0);
  // (2) this wave's superblock: 8 x 16 B of nibbles (+ Q6_K: 8 x 8 B of high bits), header, all in flight now
  uint4 q[8]; uint2 qh[8]; uint4 hd; __half dsb = __float2half(0.f);
  {
    const uint4* p = Wq_ + ((size_t)tqc * C32 + (size_t)ksl * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = ldnt(p + i * 64);
    hd = Hd_[((size_t)tqc * SB + ksl) * 64 + lane];
    if (FMT == GQ_Q6K || FMT == GQ_MIX) {
      const int t6 = is6 ? tqc : 0;              // (MIX, Q4_K wave: harmless loads of tile 0)
      const uint2* ph = Wh_ + ((size_t)t6 * C32 + (size_t)ksl * 8) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 8; i++) qh[i] = ph[i * 64];
      dsb = Dd_[((size_t)t6 * SB + ksl) * 64 + lane];
    }
  }
  // (3) the activation slice
  if (MODE == PRO_NORM) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int i = j * 2048 + tid * 4;
      if (hasprev) {
#pragma unroll
        for (int e = 0; e < 4; e++) hv[j][e] = round_act(hv[j][e] + vcvt<FIX>(pv[j][e], pro.act), pro.act);
      }
      ss += hv[j][0] * hv[j][0] + hv[j][1] * hv[j][1] + hv[j][2] * hv[j][2] + hv[j][3] * hv[j][3];
      if (blockIdx.x == 0 && pro.h_out) *(float4*)(pro.h_out + i) = make_float4(hv[j][0], hv[j][1], hv[j][2], hv[j][3]);
      if ((i >> 8) == ksl) *(float4*)(xs + (i & 255)) = make_float4(hv[j][0], hv[j][1], hv[j][2], hv[j][3]);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float rs = rms_scale(ss, (float)pro.H, pro.eps);
    if (tid < 64) {
      const float4 v = *(const float4*)(xs + tid * 4);
      *(float4*)(xs + tid * 4) = make_float4(round_act(nw.x * round_act(v.x * rs, pro.act), pro.act), round_act(nw.y * round_act(v.y * rs, pro.act), pro.act),
                                             round_act(nw.z * round_act(v.z * rs, pro.act), pro.act), round_act(nw.w * round_act(v.w * rs, pro.act), pro.act));
    }
  } else {
    if (tid < 256) xs[tid] = round_act(round_act(silu_f(vcvt<FIX>(ga, pro.act)), pro.act) * vcvt<FIX>(ua, pro.act), pro.act);
  }
  __syncthreads();
  quant_x32<FMT>(xs, 256, xh, xm, xl, cpar, cpar6);
  __syncthreads();
  if (!q_on) return;
  const int4* cp6 = FMT == GQ_MIX ? cpar6 : cpar;   // the chunk parameters in the Q6_K convention
  const uint4* xh4 = (const uint4*)xh;
  const uint4* xm4 = (const uint4*)xm;
  const uint4* xl4 = (const uint4*)xl;
  float y = 0.f;
  if (!is6) {
    const unsigned hw[4] = {hd.x, hd.y, hd.z, hd.w};
    const float d = __half2float(__ushort_as_half((unsigned short)(hw[0] & 0xffffu)));
    const float dmin = __half2float(__ushort_as_half((unsigned short)(hw[0] >> 16)));
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const unsigned ww[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
      unsigned w[8];
#pragma unroll
      for (int j = 0; j < 4; j++) { w[2 * j] = ww[j] & 0x0F0F0F0Fu; w[2 * j + 1] = ww[j] & 0xF0F0F0F0u; }
      int ua_[3] = {0, 0, 0}, ub_[3] = {0, 0, 0};
      const uint4 h0 = xh4[c * 2], h1 = xh4[c * 2 + 1], m0 = xm4[c * 2], m1 = xm4[c * 2 + 1], l0 = xl4[c * 2], l1 = xl4[c * 2 + 1];
      const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        ua_[0] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xh[2 * j], ua_[0], false);
        ua_[1] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xm[2 * j], ua_[1], false);
        ua_[2] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xl[2 * j], ua_[2], false);
        ub_[0] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xh[2 * j + 1], ub_[0], false);
        ub_[1] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xm[2 * j + 1], ub_[1], false);
        ub_[2] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xl[2 * j + 1], ub_[2], false);
      }
      const int4 p0 = cpar[2 * c], p1 = cpar[2 * c + 1];
      const float qx = planes_f((ua_[0] << 4) + ub_[0] + 128 * p1.x, (ua_[1] << 4) + ub_[1] + 128 * p1.y, (ua_[2] << 4) + ub_[2] + 128 * p1.z) * (1.0f / 16.0f);
      const float sx_ = planes_f(p0.y, p0.z, p0.w);
      int sc, mn;
      q4k_scale_min(hw, c, sc, mn);
      y += __int_as_float(p0.x) * ((d * (float)sc) * qx - (dmin * (float)mn) * sx_);
    }
  } else {   // Q6_K
    const unsigned sw[4] = {hd.x, hd.y, hd.z, hd.w};   // 16 int8 scales
    const float d = __half2float(dsb);
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const unsigned ww[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
      const unsigned hh[2] = {qh[c].x, qh[c].y};
      unsigned w[8];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned lo_a = ww[j] & 0x0F0F0F0Fu, lo_b = (ww[j] >> 4) & 0x0F0F0F0Fu;
        const int fa = 2 * j, fb = 2 * j + 1;
        const unsigned hi_a = ((hh[fa >> 2] >> (2 * (fa & 3))) & 0x03030303u) << 4;
        const unsigned hi_b = ((hh[fb >> 2] >> (2 * (fb & 3))) & 0x03030303u) << 4;
        w[2 * j] = lo_a | hi_a;
        w[2 * j + 1] = lo_b | hi_b;
      }
      const uint4 h0 = xh4[c * 2], h1 = xh4[c * 2 + 1], m0 = xm4[c * 2], m1 = xm4[c * 2 + 1], l0 = xl4[c * 2], l1 = xl4[c * 2 + 1];
      const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
      int u0[3] = {0, 0, 0}, u1[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        u0[0] = __builtin_amdgcn_sdot4((int)w[j], (int)Xh[j], u0[0], false);
        u0[1] = __builtin_amdgcn_sdot4((int)w[j], (int)Xm[j], u0[1], false);
        u0[2] = __builtin_amdgcn_sdot4((int)w[j], (int)Xl[j], u0[2], false);
        u1[0] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xh[4 + j], u1[0], false);
        u1[1] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xm[4 + j], u1[1], false);
        u1[2] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xl[4 + j], u1[2], false);
      }
      const int4 p0 = cp6[2 * c], p1 = cp6[2 * c + 1];
      const int s0 = (int)(signed char)((sw[(2 * c) >> 2] >> (8 * ((2 * c) & 3))) & 255u);
      const int s1 = (int)(signed char)((sw[(2 * c + 1) >> 2] >> (8 * ((2 * c + 1) & 3))) & 255u);
      const float f0 = planes_f(u0[0] - 32 * p0.y, u0[1] - 32 * p0.z, u0[2] - 32 * p0.w);
      const float f1 = planes_f(u1[0] - 32 * p1.x, u1[1] - 32 * p1.y, u1[2] - 32 * p1.z);
      y += __int_as_float(p0.x) * ((d * (float)s0) * f0 + (d * (float)s1) * f1);
    }
  }
  const int n = tq * 64 + lane;
  if (bias != nullptr && ksl == 0) y += bias[n];
  atomicAdd((unsigned long long*)(acc + n), (unsigned long long)f2fix(y, pro.act));
}

// ---------------------------------------------------------------------------------------------------------
// Fused MLP for the GGUF block formats (Q4_K gate / up, Q4_K or Q6_K down: the Q4_K_M mix), the structure of k_mlp_q4g:
//   acc_down += Wd[:, slab] . (silu(Wg[slice] x) * (Wu[slice] x)),   x = RMSNorm(h + prev)        (f32 activations: gguf.rs:305)
// grid = I / 64 workgroups of NW = H / 256 waves: wave w owns superblock w (256 k) of the slice's gate tile and up tile, then 64 / NW ... output tiles
// of the workgroup's 64-k slab of down_proj (two 32-k chunks of one superblock, with that superblock's header).  Wave roles as in k_mlp_q4g: the
// first half requests its gate / up superblocks at entry, the second half holds the row, does the norm and publishes the int8 planes (three per
// 32-k chunk, quant8_x32); LDS counters instead of barriers on that path.  One launch instead of two slim GEMVs: the SiLU * up vector never
// leaves the CU, the down slab's loads fly under the gate / up dots.
// ---------------------------------------------------------------------------------------------------------
// one 32-k chunk of a Q4_K superblock: weight piece qc, chunk's planes at index pc, sub-block cs of the superblock header hw
__device__ __forceinline__ float gq_chunk_q4k(const uint4& qc, int pc, int cs, const unsigned (&hw)[4], float d, float dmin, const uint4* xh4, const uint4* xm4, const uint4* xl4,
                                               const int4* cpar) {
  const unsigned ww[4] = {qc.x, qc.y, qc.z, qc.w};
  unsigned w[8];
#pragma unroll
  for (int j = 0; j < 4; j++) { w[2 * j] = ww[j] & 0x0F0F0F0Fu; w[2 * j + 1] = ww[j] & 0xF0F0F0F0u; }
  int ua_[3] = {0, 0, 0}, ub_[3] = {0, 0, 0};
  const uint4 h0 = xh4[pc * 2], h1 = xh4[pc * 2 + 1], m0 = xm4[pc * 2], m1 = xm4[pc * 2 + 1], l0 = xl4[pc * 2], l1 = xl4[pc * 2 + 1];
  const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
  const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    ua_[0] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xh[2 * j], ua_[0], false);
    ua_[1] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xm[2 * j], ua_[1], false);
    ua_[2] = __builtin_amdgcn_sdot4((int)w[2 * j], (int)Xl[2 * j], ua_[2], false);
    ub_[0] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xh[2 * j + 1], ub_[0], false);
    ub_[1] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xm[2 * j + 1], ub_[1], false);
    ub_[2] = __builtin_amdgcn_sdot4((int)w[2 * j + 1], (int)Xl[2 * j + 1], ub_[2], false);
  }
  const int4 p0 = cpar[2 * pc], p1 = cpar[2 * pc + 1];
  const float qx = planes_f((ua_[0] << 4) + ub_[0] + 128 * p1.x, (ua_[1] << 4) + ub_[1] + 128 * p1.y, (ua_[2] << 4) + ub_[2] + 128 * p1.z) * (1.0f / 16.0f);
  const float sx_ = planes_f(p0.y, p0.z, p0.w);
  int sc, mn;
  q4k_scale_min(hw, cs, sc, mn);
  return __int_as_float(p0.x) * ((d * (float)sc) * qx - (dmin * (float)mn) * sx_);
}
// one 32-k chunk of a Q6_K superblock: low nibbles qc, 2-bit highs qhc, 16 int8 scales sw, chunk cs of the superblock
__device__ __forceinline__ float gq_chunk_q6k(const uint4& qc, const uint2& qhc, int pc, int cs, const unsigned (&sw)[4], float d, const uint4* xh4, const uint4* xm4,
                                               const uint4* xl4, const int4* cpar) {
  const unsigned ww[4] = {qc.x, qc.y, qc.z, qc.w};
  const unsigned hh[2] = {qhc.x, qhc.y};
  unsigned w[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned lo_a = ww[j] & 0x0F0F0F0Fu, lo_b = (ww[j] >> 4) & 0x0F0F0F0Fu;
    const int fa = 2 * j, fb = 2 * j + 1;
    const unsigned hi_a = ((hh[fa >> 2] >> (2 * (fa & 3))) & 0x03030303u) << 4;
    const unsigned hi_b = ((hh[fb >> 2] >> (2 * (fb & 3))) & 0x03030303u) << 4;
    w[2 * j] = lo_a | hi_a;
    w[2 * j + 1] = lo_b | hi_b;
  }
  const uint4 h0 = xh4[pc * 2], h1 = xh4[pc * 2 + 1], m0 = xm4[pc * 2], m1 = xm4[pc * 2 + 1], l0 = xl4[pc * 2], l1 = xl4[pc * 2 + 1];
  const unsigned Xh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  const unsigned Xm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
  const unsigned Xl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
  int u0[3] = {0, 0, 0}, u1[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    u0[0] = __builtin_amdgcn_sdot4((int)w[j], (int)Xh[j], u0[0], false);
    u0[1] = __builtin_amdgcn_sdot4((int)w[j], (int)Xm[j], u0[1], false);
    u0[2] = __builtin_amdgcn_sdot4((int)w[j], (int)Xl[j], u0[2], false);
    u1[0] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xh[4 + j], u1[0], false);
    u1[1] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xm[4 + j], u1[1], false);
    u1[2] = __builtin_amdgcn_sdot4((int)w[4 + j], (int)Xl[4 + j], u1[2], false);
  }
  const int4 p0 = cpar[2 * pc], p1 = cpar[2 * pc + 1];
  const int s0 = (int)(signed char)((sw[(2 * cs) >> 2] >> (8 * ((2 * cs) & 3))) & 255u);
  const int s1 = (int)(signed char)((sw[(2 * cs + 1) >> 2] >> (8 * ((2 * cs + 1) & 3))) & 255u);
  const float f0 = planes_f(u0[0] - 32 * p0.y, u0[1] - 32 * p0.z, u0[2] - 32 * p0.w);
  const float f1 = planes_f(u1[0] - 32 * p1.x, u1[1] - 32 * p1.y, u1[2] - 32 * p1.z);
  return __int_as_float(p0.x) * ((d * (float)s0) * f0 + (d * (float)s1) * f1);
}

template <int FMTD, int FIX, int NW, int TPW>   // FMTD: down_proj format (GQ_Q4K / GQ_Q6K); NW = H / 256 waves; TPW = H / 64 / NW down tiles per wave
__global__ __launch_bounds__(NW * 64) void k_mlp_gq(const uint4* __restrict__ Wgu, const uint4* __restrict__ Hgu, const float* __restrict__ bgu, const uint4* __restrict__ Wd,
                                                const uint2* __restrict__ Whd, const uint4* __restrict__ Hdd, const __half* __restrict__ Ddd, const float* __restrict__ bd,
                                                int H, int I, Pro pro, long long* acc, long long* zero_buf, int zero_n) {
  constexpr int NP = NW / 2, NTH = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* xh = (unsigned*)smem;             // [H/4] x 3
  unsigned* xm = xh + H / 4;
  unsigned* xl = xm + H / 4;
  int4* cpar = (int4*)(xl + H / 4);           // [2 H/32]
  float* part = (float*)(cpar + 2 * (H >> 5));  // [NW][128]
  float* av = part + NW * 128;                // [64]
  unsigned* ah = (unsigned*)(av + 64);        // [16] x 3
  unsigned* am_ = ah + 16;
  unsigned* al = am_ + 16;
  int4* apar = (int4*)(al + 16);              // [4]
  float* red = (float*)(apar + 4);            // [NP]
  volatile unsigned* cnt = (volatile unsigned*)(red + NW);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool prolog = wave >= NP;
  const int sl = blockIdx.x, NTI = I >> 6, C32 = H >> 5, SBH = H >> 8;
  asm volatile("" :: "s"(zero_buf), "s"(zero_n), "s"(acc), "s"(pro.h_in), "s"(pro.src.p), "s"(pro.norm_w), "s"(pro.h_out), "s"(pro.H), "s"(pro.act),
               "s"(H), "s"(I), "s"(Wgu), "s"(Hgu), "s"(Wd), "s"(Whd), "s"(Hdd), "s"(Ddd));
  const uint4* pg = Wgu + ((size_t)sl * C32 + (size_t)wave * 8) * 64 + lane;
  const uint4* pu = Wgu + ((size_t)(NTI + sl) * C32 + (size_t)wave * 8) * 64 + lane;
  uint4 qg[8], qu[8];
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  if (prolog) {
    const bool hasprev = pro.src.p != nullptr;
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
    const int oct = tid - NP * 64, i0 = oct * 8;
    const float4 ha = *(const float4*)(pro.h_in + i0), hb = *(const float4*)(pro.h_in + i0 + 4);
    typename RawT<FIX>::T pv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) pv[e] = vraw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
    const float4 na = *(const float4*)(pro.norm_w + i0), nb = *(const float4*)(pro.norm_w + i0 + 4);
    __builtin_amdgcn_sched_barrier(0);
    qg[0] = ldnt(pg); qu[0] = ldnt(pu);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // counters zeroed; nobody waits for data here
    float v[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { if (hasprev) v[e] = round_act(v[e] + vcvt<FIX>(pv[e], pro.act), pro.act); ss += v[e] * v[e]; }
    if (blockIdx.x == 0 && pro.h_out) { *(float4*)(pro.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
    ss = wave_sum(ss);
    if (lane == 0) { red[wave - NP] = ss; __builtin_amdgcn_s_waitcnt(0xc07f); atomicAdd((unsigned*)&cnt[0], 1u); }
    lds_wait_count(&cnt[0], NP);
    float tot = (red[0] + red[1]) + (red[2] + red[3]);
    if (NP == 8) tot += (red[4] + red[5]) + (red[6] + red[7]);
    const float rs = rms_scale(tot, (float)H, pro.eps);
    const float nwv[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = round_act(nwv[e] * round_act(v[e] * rs, pro.act), pro.act);
    quant8_x32<GQ_Q4K>(x, i0, true, lane, xh, xm, xl, cpar);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) atomicAdd((unsigned*)&cnt[1], 1u);
#pragma unroll
    for (int c = 1; c < 8; c++) { qg[c] = ldnt(pg + c * 64); qu[c] = ldnt(pu + c * 64); }
  } else {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; c++) { qg[c] = ldnt(pg + c * 64); qu[c] = ldnt(pu + c * 64); }
    zero_duty<NP * 64>(zero_buf, zero_n);
  }
  const uint4 hg = Hgu[((size_t)sl * SBH + wave) * 64 + lane], hu = Hgu[((size_t)(NTI + sl) * SBH + wave) * 64 + lane];
  lds_wait_count(&cnt[1], NP);
  const uint4* xh4 = (const uint4*)xh;
  const uint4* xm4 = (const uint4*)xm;
  const uint4* xl4 = (const uint4*)xl;
  float yg = 0.f, yu = 0.f;
  {
    const unsigned hwg[4] = {hg.x, hg.y, hg.z, hg.w}, hwu[4] = {hu.x, hu.y, hu.z, hu.w};
    const float dg = __half2float(__ushort_as_half((unsigned short)(hwg[0] & 0xffffu))), dming = __half2float(__ushort_as_half((unsigned short)(hwg[0] >> 16)));
    const float du = __half2float(__ushort_as_half((unsigned short)(hwu[0] & 0xffffu))), dminu = __half2float(__ushort_as_half((unsigned short)(hwu[0] >> 16)));
#pragma unroll
    for (int c = 0; c < 8; c++) {
      yg += gq_chunk_q4k(qg[c], wave * 8 + c, c, hwg, dg, dming, xh4, xm4, xl4, cpar);
      yu += gq_chunk_q4k(qu[c], wave * 8 + c, c, hwu, du, dminu, xh4, xm4, xl4, cpar);
    }
  }
  // the down slab: chunks 2 sl, 2 sl + 1 of down's k range = chunks (2 sl) % 8, +1 of superblock sl / 4; TPW tiles per wave
  const int C32D = I >> 5, SBD = I >> 8, sbd = sl >> 2, cs0 = (2 * sl) & 7;
  uint4 D[TPW][2]; uint2 Dh[TPW][2]; uint4 hdd[TPW]; __half ddd[TPW];
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    const int t = wave * TPW + q;
    const uint4* wp = Wd + ((size_t)t * C32D + 2 * sl) * 64 + lane;
    D[q][0] = ldnt(wp); D[q][1] = ldnt(wp + 64);
    hdd[q] = Hdd[((size_t)t * SBD + sbd) * 64 + lane];
    if (FMTD == GQ_Q6K) {
      const uint2* wh = Whd + ((size_t)t * C32D + 2 * sl) * 64 + lane;
      Dh[q][0] = wh[0]; Dh[q][1] = wh[64];
      ddd[q] = Ddd[((size_t)t * SBD + sbd) * 64 + lane];
    }
  }
  part[wave * 128 + lane] = yg;
  part[wave * 128 + 64 + lane] = yu;
  __syncthreads();
  if (tid < 64) {
    float tg = 0.f, tu = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) { tg += part[w2 * 128 + tid]; tu += part[w2 * 128 + 64 + tid]; }
    if (bgu) { tg += bgu[sl * 64 + tid]; tu += bgu[I + sl * 64 + tid]; }
    tg = round_act(tg, pro.act); tu = round_act(tu, pro.act);
    av[tid] = round_act(round_act(silu_f(tg), pro.act) * tu, pro.act);
  }
  __syncthreads();
  if (wave == 0) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] = lane < 8 ? av[lane * 8 + e] : 0.f;
    if (FMTD == GQ_Q6K) quant8_x32<GQ_Q6K>(x, lane * 8, lane < 8, lane, ah, am_, al, apar); else quant8_x32<GQ_Q4K>(x, lane * 8, lane < 8, lane, ah, am_, al, apar);
  }
  __syncthreads();
  const uint4* ah4 = (const uint4*)ah;
  const uint4* am4 = (const uint4*)am_;
  const uint4* al4 = (const uint4*)al;
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    const unsigned hw[4] = {hdd[q].x, hdd[q].y, hdd[q].z, hdd[q].w};
    float y = 0.f;
    if (FMTD == GQ_Q4K) {
      const float d = __half2float(__ushort_as_half((unsigned short)(hw[0] & 0xffffu))), dmin = __half2float(__ushort_as_half((unsigned short)(hw[0] >> 16)));
      y += gq_chunk_q4k(D[q][0], 0, cs0, hw, d, dmin, ah4, am4, al4, apar);
      y += gq_chunk_q4k(D[q][1], 1, cs0 + 1, hw, d, dmin, ah4, am4, al4, apar);
    } else {
      const float d = __half2float(ddd[q]);
      y += gq_chunk_q6k(D[q][0], Dh[q][0], 0, cs0, hw, d, ah4, am4, al4, apar);
      y += gq_chunk_q6k(D[q][1], Dh[q][1], 1, cs0 + 1, hw, d, ah4, am4, al4, apar);
    }
    const int n = (wave * TPW + q) * 64 + lane;
    if (bd != nullptr && sl == 0) y += bd[n];
    atomicAdd((unsigned long long*)(acc + n), (unsigned long long)f2fix(y, pro.act));
  }
}
static size_t mlp_gq_smem(int H) { return (size_t)H * 3 + (size_t)(H >> 5) * 32 + 16 * 128 * 4 + 64 * 4 + 48 * 4 + 64 + 16 * 4 + 16 + 64; }
bool bzk_mlp_gq_fusable(const LinearDev& gu, const LinearDev& dn, int H, int I, int act) {
  // opt-in: measured on the Mistral-7B Q4_K_M shape the fused launch takes 31.0 us against ~19 + ~11 us for the two slim launches (535.6 vs 549.0 tok/s, same
  // box): the block formats cost ~2.5 VALU operations per weight (AND-unpack + three-plane V_DOT4 + a per-32-k epilogue of scale / min arithmetic), ~19 us
  // of issue per CU when 224 workgroups carry all of it -- the fused form serialises that behind its stream, the slim launches spread it over 256 CUs
  static const bool on = getenv("BZ_GGUF_MLP_FUSION") != nullptr && getenv("BZ_NO_MLP_FUSION") == nullptr;
  return on && act == BZ_F32 && gu.kind == LK_Q4K && (dn.kind == LK_Q4K || dn.kind == LK_Q6K) && !gu.perm && !dn.perm && gu.N == 2 * I && gu.K == H && dn.N == H && dn.K == I &&
         (H == 2048 || H == 4096) && I % 256 == 0 && I % 64 == 0;
}
int bzk_mlp_gq(hipStream_t s, const LinearDev& gu, const LinearDev& dn, int H, int I, const Pro& pro, long long* acc, long long* zero_buf, int zero_n) {
  if (!bzk_mlp_gq_fusable(gu, dn, H, I, pro.act)) BZ_FAIL(BZ_E_INVALID, "fused GGUF MLP does not apply to this shape");
  const size_t smem = mlp_gq_smem(H);
  const double bytes = (double)gu.algo_bytes + (double)dn.algo_bytes;
#define LAUNCH_MGQ(FD, FIX, NW_, TP) BZ_LAUNCH("mlp_gguf<norm+gate/up+silu+down>", bytes, (k_mlp_gq<FD, FIX, NW_, TP>), dim3(I / 64), dim3(NW_ * 64), smem, s, (const uint4*)gu.w, \
    (const uint4*)gu.hdr, gu.bias, (const uint4*)dn.w, (const uint2*)dn.zeros, (const uint4*)dn.hdr, (const __half*)dn.scales, dn.bias, H, I, pro, acc, zero_buf, zero_n)
#define LAUNCH_MGQ_F(FD, NW_, TP) do { if (pro.src.fix) LAUNCH_MGQ(FD, 1, NW_, TP); else LAUNCH_MGQ(FD, 0, NW_, TP); } while (0)
  if (H == 4096) { if (dn.kind == LK_Q6K) LAUNCH_MGQ_F(GQ_Q6K, 16, 4); else LAUNCH_MGQ_F(GQ_Q4K, 16, 4); }
  else { if (dn.kind == LK_Q6K) LAUNCH_MGQ_F(GQ_Q6K, 8, 4); else LAUNCH_MGQ_F(GQ_Q4K, 8, 4); }
#undef LAUNCH_MGQ_F
#undef LAUNCH_MGQ
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

bool bzk_gq_slim_ok(const LinearDev& L, const Pro& pro) {
  static const bool off = getenv("BZ_NO_GQ_SLIM") != nullptr;
  if (off || (L.kind != LK_Q4K && L.kind != LK_Q6K) || pro.perm != nullptr || pro.dbg || pro.stamps || L.N % 64 || L.K % 256) return false;
  if (pro.mode == PRO_NORM) return L.K == pro.H && (L.K == 2048 || L.K == 4096 || L.K == 8192);
  if (pro.mode == PRO_SILU) return L.K == pro.H;
  return false;
}


// Q4_K tiles followed by Q6_K tiles (one K, one prologue) in ONE slim launch: the Q4_K_M rule stores q / k as Q4_K and v as Q6_K, which made q/k/v two launches
bool bzk_gq_mix_ok(const LinearDev& A, const LinearDev& B, const Pro& pro) {
  return A.kind == LK_Q4K && B.kind == LK_Q6K && A.K == B.K && !A.bias && !B.bias && pro.mode == PRO_NORM && bzk_gq_slim_ok(A, pro) && bzk_gq_slim_ok(B, pro);
}
int bzk_gemv_gq_mix(hipStream_t s, const LinearDev& A, const LinearDev& B, const Pro& pro, const GemvOut& out) {
  if (!bzk_gq_mix_ok(A, B, pro) || !out.acc) BZ_FAIL(BZ_E_INVALID, "mixed Q4_K / Q6_K slim launch does not apply");
  const int N = A.N + B.N, K = A.K, nsb = K / 256, ntg = (N / 64 + 7) / 8;
  const GqMix mix{(const uint4*)B.w, (const uint2*)B.zeros, (const uint4*)B.hdr, (const __half*)B.scales, A.N / 64};
  const double bytes = (double)A.algo_bytes + (double)B.algo_bytes;
#define LAUNCH_GQM(FIX, NJ) BZ_LAUNCH("gemv_q4_K+q6_K<slim>", bytes, (k_gemv_gq_slim<GQ_MIX, PRO_NORM, FIX, NJ>), dim3(nsb * ntg), dim3(512), 0, s, (const uint4*)A.w, \
    (const uint2*)nullptr, (const uint4*)A.hdr, (const __half*)nullptr, (const float*)nullptr, N, K, pro, out.acc, out.zero_buf, out.zero_n, mix)
#define LAUNCH_GQM_NJ(FIX) do { if (K == 2048) LAUNCH_GQM(FIX, 1); else if (K == 4096) LAUNCH_GQM(FIX, 2); else LAUNCH_GQM(FIX, 4); } while (0)
  if (pro.src.fix) LAUNCH_GQM_NJ(1); else LAUNCH_GQM_NJ(0);
#undef LAUNCH_GQM_NJ
#undef LAUNCH_GQM
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

static size_t gq_smem(int SBW) { size_t KR = (size_t)SBW * 256; return KR * 4 + KR * 3 + KR + 64; }

static size_t q4g_smem(int GW) {
  size_t KR = (size_t)GW * 128;
  return KR * 4 + (KR >> 5) * XQ_NP * 16 + (size_t)GW * 32 + (size_t)4 * GW * 64 * 2 + (size_t)4 * GW * 64 + 16;
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

// raw 8-weight piece of one row: one 16-byte load for f16 / bf16, two for f32
template <int WDT> struct RowPiece { uint4 a; };
template <> struct RowPiece<BZ_F32> { uint4 a, b; };
template <int WDT>
__device__ __forceinline__ RowPiece<WDT> piece_load(const void* W, size_t elem_off) {
  RowPiece<WDT> p;
  if constexpr (WDT == BZ_F32) { p.a = ldnt((const uint4*)((const float*)W + elem_off)); p.b = ldnt((const uint4*)((const float*)W + elem_off + 4)); }
  else p.a = ldnt((const uint4*)((const unsigned short*)W + elem_off));
  return p;
}
template <int WDT>
__device__ __forceinline__ float piece_dot(const RowPiece<WDT>& p, const float4& xa, const float4& xb) {
  float w[8];
  if constexpr (WDT == BZ_F32) {
    w[0] = __uint_as_float(p.a.x); w[1] = __uint_as_float(p.a.y); w[2] = __uint_as_float(p.a.z); w[3] = __uint_as_float(p.a.w);
    w[4] = __uint_as_float(p.b.x); w[5] = __uint_as_float(p.b.y); w[6] = __uint_as_float(p.b.z); w[7] = __uint_as_float(p.b.w);
  } else {
    const unsigned u[4] = {p.a.x, p.a.y, p.a.z, p.a.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (WDT == BZ_F16) {
        w[2 * i] = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu)));
        w[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16)));
      } else {
        w[2 * i] = __uint_as_float(u[i] << 16);
        w[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
      }
    }
  }
  return w[0] * xa.x + w[1] * xa.y + w[2] * xa.z + w[3] * xa.w + w[4] * xb.x + w[5] * xb.y + w[6] * xb.z + w[7] * xb.w;
}
// The same eight products added to `acc` in double.  The oracle DEFINES a linear layer's output as the exactly rounded dot product (oracle/orc_quant.c: products
// exact in double, the sum carried in double, one rounding to f32, one to the activation dtype); a 16-bit weight times a 16-bit-valued activation is exact, the sum
// over K in double is order-independent to ~1e-16, so a dense GEMV that accumulates like this produces the oracle's bits whatever its decomposition over lanes,
// waves and workgroups -- which f32 FMA chains do not (round 2: two correct bf16 pipelines sat at the bf16 noise floor from each other, 2e-2 at 16 layers).
// Cost: 8 converts + 8 double FMAs per eight weights against 8 f32 FMAs; the dense GEMVs are bound by their stream, not by this.
template <int WDT>
__device__ __forceinline__ double piece_dot_d(const RowPiece<WDT>& p, const double (&x)[8], double acc) {
  float w[8];
  if constexpr (WDT == BZ_F32) {
    w[0] = __uint_as_float(p.a.x); w[1] = __uint_as_float(p.a.y); w[2] = __uint_as_float(p.a.z); w[3] = __uint_as_float(p.a.w);
    w[4] = __uint_as_float(p.b.x); w[5] = __uint_as_float(p.b.y); w[6] = __uint_as_float(p.b.z); w[7] = __uint_as_float(p.b.w);
  } else {
    const unsigned u[4] = {p.a.x, p.a.y, p.a.z, p.a.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if constexpr (WDT == BZ_F16) {
        w[2 * i] = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu)));
        w[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16)));
      } else {
        w[2 * i] = __uint_as_float(u[i] << 16);
        w[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) acc = fma((double)w[e], x[e], acc);
  return acc;
}
// 16-bit weights times 16-bit-valued activations: the f32 product is already exact (8 + 8 or 11 + 11 significant bits), so the double only has to carry the SUM --
// one f32 multiply, one convert, one double add per element, and no double copy of x in registers (the 832-thread routing form of k_gemv_rows2 has 128 registers)
template <int WDT>
__device__ __forceinline__ double piece_dot_x(const RowPiece<WDT>& p, const float4& xa, const float4& xb, double acc) {
  static_assert(WDT == BZ_F16 || WDT == BZ_BF16, "exact f32 products need 16-bit operands");
  const unsigned u[4] = {p.a.x, p.a.y, p.a.z, p.a.w};
  const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float w0, w1;
    if constexpr (WDT == BZ_F16) { w0 = __half2float(__ushort_as_half((unsigned short)(u[i] & 0xffffu))); w1 = __half2float(__ushort_as_half((unsigned short)(u[i] >> 16))); }
    else { w0 = __uint_as_float(u[i] << 16); w1 = __uint_as_float(u[i] & 0xffff0000u); }
    acc += (double)__fmul_rn(w0, x[2 * i]);
    acc += (double)__fmul_rn(w1, x[2 * i + 1]);
  }
  return acc;
}
__device__ __forceinline__ void x8_to_d(const float4& xa, const float4& xb, double (&x)[8]) {
  x[0] = (double)xa.x; x[1] = (double)xa.y; x[2] = (double)xa.z; x[3] = (double)xa.w; x[4] = (double)xb.x; x[5] = (double)xb.y; x[6] = (double)xb.z; x[7] = (double)xb.w;
}

// A wave walks its rows four at a time, one 512-k chunk per step (row group, chunk).  Loads run two steps ahead of the
// FMAs (three stage buffers), and the first two stages are issued before the prologue so that the weight stream is
// already in flight while x is built.
// SPLIT: split-K.  blockIdx = row block * SK + ks; slice ks covers KCs chunks of 512 k; partial sums go to the 64-bit
// fixed-point accumulator `accbuf` (order-independent, hence deterministic) and are rounded by the consumer.
template <int WDT, bool SPLIT>
__device__ __forceinline__ void rows_body(const void* __restrict__ W, const float* __restrict__ bias, int N, int K, int rows_per_wg, const Pro& pro,
                                          float* out, int act, float* pval, int* pidx, long long* zero_buf, int zero_n, int SK, long long* accbuf,
                                          const int bid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KP = (K + 511) & ~511;
  float* xs = (float*)smem;     // [KP] swizzled
  float* red = xs + KP;         // [4] + argmax scratch [4] + [4]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rpw = rows_per_wg >> 2;
  const int rb = SPLIT ? bid / SK : bid, ks = SPLIT ? bid % SK : 0;
  const int rbeg = rb * rows_per_wg + wave * rpw;
  const int rend = min(rbeg + rpw, N);
  const int KCall = KP >> 9, KCs = SPLIT ? (KCall + SK - 1) / SK : KCall;
  const int kc0 = ks * KCs;
  const int KC = max(min(KCs, KCall - kc0), 1);
  const int kb = kc0 * 512;                                 // first k of this slice
  const int ngroups = rend > rbeg ? (rend - rbeg + 3) >> 2 : 0;
  const int nsteps = ngroups * KC;
  struct Stage { RowPiece<WDT> p[4]; };
  // issue cursor (clamped at the last step: a few redundant loads at the tail, never a branch around a load)
  int ir = min(rbeg, N - 1), ikc = 0, ist = 0;
  auto issue = [&](Stage& S) {
    const int k = kb + ikc * 512 + lane * 8;
    const int ko = k < K ? k : 0;                           // xs is 0 beyond K
#pragma unroll
    for (int rr = 0; rr < 4; rr++) S.p[rr] = piece_load<WDT>(W, (size_t)min(ir + rr, N - 1) * K + ko);
    if (ist + 1 < nsteps) { ist++; if (++ikc == KC) { ikc = 0; ir += 4; } }
  };
  Stage s0, s1, s2;
  issue(s0);
  issue(s1);
  __builtin_amdgcn_sched_barrier(0);
  zero_duty<256>(zero_buf, zero_n);
  const float4* xs4;
  if (!SPLIT || pro.mode == PRO_GATED) {
    // whole vector in LDS (the gated norm needs every element of its group anyway)
    for (int i = K + threadIdx.x; i < KP; i += 256) xs[xs_pos<true>(i)] = 0.f;
    build_x_simple<true>(pro, 0, K, xs, red, bid == 0);
    xs4 = (const float4*)(xs + kb);
  } else {
    const int KR = max(min(K - kb, KC * 512), 0);
    for (int i = KR + threadIdx.x; i < KC * 512; i += 256) xs[xs_pos<true>(i)] = 0.f;
    build_x_simple<true>(pro, kb, KR, xs, red, bid == 0);
    xs4 = (const float4*)xs;
  }
  float bestv = -INFINITY; int besti = 0x7fffffff;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};       // exact sums (piece_dot_d): a row's value does not depend on the lane / chunk decomposition
  int cr = rbeg, ckc = 0, cst = 0;
  auto consume = [&](const Stage& S) {
    if (cst >= nsteps) return;
    cst++;
    const float4 xa = xs4[ckc * 128 + lane], xb = xs4[ckc * 128 + 64 + lane];
    if constexpr (WDT == BZ_F32) {       // f32 weights: the products need the double too
      double xd[8];
      x8_to_d(xa, xb, xd);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) acc[rr] = piece_dot_d<WDT>(S.p[rr], xd, acc[rr]);
    } else {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) acc[rr] = piece_dot_x<WDT>(S.p[rr], xa, xb, acc[rr]);
    }
    if (++ckc == KC) {
      ckc = 0;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const double vd = wave_sum_d(acc[rr]);
        acc[rr] = 0.0;
        if (cr + rr < rend) {
          if (SPLIT) {
            if (lane == 0) atomicAdd((unsigned long long*)(accbuf + cr + rr), (unsigned long long)d2fix(vd + ((bias && ks == 0) ? (double)bias[cr + rr] : 0.0), pro.act));
          } else {
            float v = (float)vd;                 // one rounding of the exact sum, then the bias in f32 (the oracle's order)
            if (bias) v += bias[cr + rr];
            v = round_act(v, act);
            if (lane == 0) out[cr + rr] = v;
            if (v > bestv) { bestv = v; besti = cr + rr; }
          }
        }
      }
      cr += 4;
    }
  };
  for (int st = 0; st < nsteps; st += 3) {
    issue(s2); consume(s0);
    issue(s0); consume(s1);
    issue(s1); consume(s2);
  }
  if (!SPLIT && pval) {
    float* bv = red + 4; int* bi = (int*)(red + 8);
    if (lane == 0) { bv[wave] = bestv; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = bv[0]; int ix = bi[0];
      for (int w2 = 1; w2 < 4; w2++) if (bv[w2] > v) { v = bv[w2]; ix = bi[w2]; }
      pval[bid] = v; pidx[bid] = ix;
    }
  }
}

template <int WDT, bool SPLIT>
__global__ __launch_bounds__(256) void k_gemv_rows(const void* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                                   int rows_per_wg, Pro pro, float* out, int act, float* pval, int* pidx,
                                                   long long* zero_buf, int zero_n, int SK, long long* accbuf) {
  rows_body<WDT, SPLIT>(W, bias, N, K, rows_per_wg, pro, out, act, pval, pidx, zero_buf, zero_n, SK, accbuf, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// Dense row GEMV, second form (16-bit weights, fixed-point output): balanced unit ranges + wave roles.
//  * unit = (512-k chunk kc, group rg of four rows), ordered chunk-major: u = kc * NRG + rg.  Workgroup b owns units [U b / nb, U (b+1) / nb) and
//    its eight tile waves split that range evenly again, so that any [N, K] loads the 256 CUs to within one unit (16-row x full-K workgroups
//    left 35 % of the chip idle on N = 10576, and ran 1024 norm prologues against 32 MB of L2 reads on N = 16384).  A range lies in at most two
//    chunks (nb >= K / 512), so a workgroup needs 1024 elements of x, eight per lane and chunk, held in REGISTERS by the tile waves.
//  * roles (see k_mlp_q4g): waves 0-3 hold the activation row -- their loads go out first, then the rendezvous barrier, then the tile waves
//    (4-11) request up to 16 KiB each and sit in the issue queue while the norm runs.  The halves meet through two LDS counters.
//  * every unit ends in a 4-value transposed reduction (3 swaps + 4 DPP steps) and one 64-bit fixed-point atomic per row: N K / 512 atomics.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
  // permlane32_swap(a, b): a's upper half <-> b's lower half; permlane16_swap(a, b): a's odd rows <-> b's even rows.  Result: the total of
  // v0 in every lane of row 0 (lanes 0-15), v2 in row 1, v1 in row 2, v3 in row 3
  bz_u2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
  const float s01 = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
  const float s23 = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
  return grp_reduce<16, OpAdd>(__uint_as_float(r.x) + __uint_as_float(r.y));
}
// the same transposed reduction of four doubles (both 32-bit halves through the same swaps)
__device__ __forceinline__ void swap32_d(double a, double b, double& ra, double& rb) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const bz_u2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false), hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  ra = __longlong_as_double(((long long)hi.x << 32) | lo.x); rb = __longlong_as_double(((long long)hi.y << 32) | lo.y);
}
__device__ __forceinline__ void swap16_d(double a, double b, double& ra, double& rb) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const bz_u2_t lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false), hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  ra = __longlong_as_double(((long long)hi.x << 32) | lo.x); rb = __longlong_as_double(((long long)hi.y << 32) | lo.y);
}
__device__ __forceinline__ double wave_sum4_d(double v0, double v1, double v2, double v3) {
  double a, b;
  swap32_d(v0, v1, a, b); const double s01 = a + b;
  swap32_d(v2, v3, a, b); const double s23 = a + b;
  swap16_d(s01, s23, a, b);
  return grp_sum_d<16>(a + b);
}

// Mamba2: the B / C channels' conv state is shifted by the launch AFTER the SSM step (every head has read the old state by then): a side duty of
// the out_proj GEMV, like the ring's zeroing duty
__device__ __forceinline__ void conv_shift_one(const ConvShift& c, int i, int act) {
  const int ch = c.ch0 + i;
  float* cs = c.cs + (size_t)ch * (c.kc - 1);
  const float xr = vsrc_get(c.src, c.x_off + ch, act);
  for (int j = 0; j + 1 < c.kc - 1; j++) cs[j] = cs[j + 1];
  cs[c.kc - 2] = xr;
}
__global__ void k_conv_shift(ConvShift c, int act) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) conv_shift_one(c, i, act);
}

// MoE form (slots.n > 1): the same launch covers the selected experts' matrices -- slot s multiplies W[sel[s]] ([N, K] each, expert_stride elements apart) with
// ITS input row (src_stride elements apart; 0: one shared row) into ITS accumulator row.  A segment is then a (chunk, slot) pair, seg = kc n_slots + slot,
// units are ordered (segment, row group), and a workgroup's range still lies in at most two segments (nb >= n_slots K / 512).
// ROUTE (DeepSeek MoE gate / up): the router's logits arrive as the fixed-point output of the previous launch (a plain k_gemv_rows2<norm> over the
// router matrix); row wave 0 turns them into the top-k selection (moe_softmax_topk) while the tile waves wait for the expert ids, workgroup 0
// publishes selection and weights for the down / combine launches.  One 5 us GEMV + ~3 us in front of this stream instead of a 14 us router launch
// whose last workgroup ran the top-k behind a device-scope counter.
__device__ void moe_softmax_topk(const float* lg, int E, int top_k, int n_shared, float routed_scale, int norm_topk, int* sel, float* wsel, int lane);
// EX: exact sums (piece_dot_d / wave_sum4_d: Llama and Mamba2, whose every other op is exact too); DeepSeek-V2 (MLA sums, router softmax in f32) keeps the f32
// chains -- there the exact GEMVs bought no bits and cost 35 % of the step (the 832-thread ROUTE form went 22 -> 62 us under the doubles' register pressure)
template <int WDT, int MODE, bool FIX, int ROUTE = 0, bool EX = false>
__global__ __launch_bounds__(ROUTE ? 832 : 768) void k_gemv_rows2(const void* __restrict__ W, const float* __restrict__ bias, int N, int K, Pro pro,
                                                    long long* __restrict__ acc, long long* zero_buf, int zero_n, ConvShift shift, MoeSlots slots, RouteArgs route) {
  // NR unit ranges per workgroup.  ROUTE: range 0 = the shared experts' slots [top_k, top_k + n_shared) (ids known: the stream starts at entry),
  // range 1 = the routed slots [0, top_k) (ids from the top-k, a few us later); otherwise one range over all slots.
  constexpr int NR = ROUTE ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float xs[NR * 1024];   // x of the (at most) two segments of each range, 512 each
  __shared__ double dred[4];
  __shared__ float grs[4][8];
  __shared__ unsigned cnt[4];
  __shared__ float lgs[ROUTE ? 1024 : 1];
  __shared__ int ssel[ROUTE ? 128 : 1];
  __shared__ float swgt[ROUTE ? 128 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NRG = (N + 3) >> 2, KCall = (K + 511) >> 9;
  int u0[NR], u1[NR], ub[NR], kcA[NR], slA[NR], kcB[NR], slB[NR], sbase[NR];
  bool hasB[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    sbase[r] = ROUTE ? (r == 0 ? route.top_k : 0) : 0;
    const int nsl = ROUTE ? (r == 0 ? route.n_shared : route.top_k) : slots.n;
    const int nseg = KCall * nsl;
    const long long U = (long long)NRG * nseg;
    u0[r] = (int)(U * blockIdx.x / gridDim.x); u1[r] = (int)(U * (blockIdx.x + 1) / gridDim.x);
    const int segA = u0[r] / NRG, segB = min(segA + 1, max(nseg - 1, 0));
    hasB[r] = segA + 1 < nseg;
    const int nd = max(nsl, 1);
    kcA[r] = segA / nd; slA[r] = segA - kcA[r] * nd; kcB[r] = segB / nd; slB[r] = segB - kcB[r] * nd;
    ub[r] = (segA + 1) * NRG;                      // first unit of the second segment
  }
  const int act = pro.act;
  // diagnostic only (BZ_MOE_STAMPS): s_memrealtime (100 MHz) of workgroup 0 -- row wave 0: entry, rendezvous, x published, top-k done; tile wave 4: range 0 done, ids seen, end
#define RSTAMP(i) do { if (ROUTE && pro.stamps && blockIdx.x == 0 && lane == 0) pro.stamps[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; cnt[2] = 0; cnt[3] = 0; }
  if (wave == 0) RSTAMP(0);
  if (ROUTE && wave == 12) {
    // ---- the routing wave (13th): logits -> softmax + top-k -> expert ids for the tile waves' second range; nobody waits for it before that
    __builtin_amdgcn_s_setprio(3);
    for (int j = lane; j < route.E; j += 64) lgs[j] = fix2f(route.lg[j], pro.act);
    __syncthreads();                               // (the rendezvous barrier: every wave passes it once)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    moe_softmax_topk(lgs, route.E, route.top_k, route.n_shared, route.routed_scale, route.norm_topk, ssel, swgt, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) atomicAdd(&cnt[2], 1u);
    RSTAMP(3);
    if (blockIdx.x == 0)
      for (int j = lane; j < route.top_k + route.n_shared; j += 64) { route.sel_out[j] = ssel[j]; route.w_out[j] = swgt[j]; }
    return;
  }
  if (wave < 4) {
    // ---- row waves: x of segments A | B (of every range) -> LDS --------------------------------------------
    __builtin_amdgcn_s_setprio(3);                                // everybody waits for these four waves
    const int hb = tid >> 7;                                      // this thread's segment (0: A, 1: B) and its four elements there
    int sk[NR], skc[NR]; bool son[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
      sk[r] = (hb ? kcB[r] : kcA[r]) * 512 + (tid & 127) * 4;
      skc[r] = min(sk[r], K - 4);                  // clamped for addressing (K % 8 == 0)
      son[r] = sk[r] < K && (hb == 0 || hasB[r]);
    }
    if (MODE == PRO_NORM) {
      const int H = pro.H;
      const bool hasprev = pro.src.p != nullptr;
      const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
      float4 nw[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) nw[r] = *(const float4*)(pro.norm_w + skc[r]);
      double ssd = 0.0;
      bool first = true;
      for (int base = 0; base < H; base += 4096) {
        float4 ha[2], hb4[2];
        typename SrcRaw<FIX>::T pv[2][8];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int i0 = min(base + (j * 256 + tid) * 8, H - 8);
          ha[j] = *(const float4*)(pro.h_in + i0); hb4[j] = *(const float4*)(pro.h_in + i0 + 4);
#pragma unroll
          for (int e = 0; e < 8; e++) pv[j][e] = src_raw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
        }
        if (first) {
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();                         // rendezvous: these loads are ahead of the weight stream
          first = false;
          if (wave == 0) RSTAMP(1);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int i0 = base + (j * 256 + tid) * 8;
          float v[8] = {ha[j].x, ha[j].y, ha[j].z, ha[j].w, hb4[j].x, hb4[j].y, hb4[j].z, hb4[j].w};
          if (hasprev) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = round_act(v[e] + src_cvt<FIX>(pv[j][e], act), act);
          }
          if (i0 < H) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) ssd += (double)(v[e] * v[e]) + (double)(v[e + 1] * v[e + 1]);
            if (blockIdx.x == 0 && pro.h_out) { *(float4*)(pro.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
#pragma unroll
            for (int r = 0; r < NR; r++) {
              float* xr = xs + r * 1024;
              if ((i0 >> 9) == kcA[r]) { *(float4*)(xr + (i0 & 511)) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(xr + (i0 & 511) + 4) = make_float4(v[4], v[5], v[6], v[7]); }
              if ((i0 >> 9) == kcB[r]) { *(float4*)(xr + 512 + (i0 & 511)) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(xr + 512 + (i0 & 511) + 4) = make_float4(v[4], v[5], v[6], v[7]); }
            }
          }
        }
      }
      ssd = wave_sum_d(ssd);
      if (lane == 0) dred[wave] = ssd;
      __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the parked slices and the partial are in LDS
      if (lane == 0) atomicAdd(&cnt[0], 1u);
      lds_wait_count(&cnt[0], 4);
      const float ss = (float)((dred[0] + dred[1]) + (dred[2] + dred[3]));   // the rounded exact sum of squares (oracle: orc_rms_norm)
      const float rs = rms_scale(ss, (float)H, pro.eps);
#pragma unroll
      for (int r = 0; r < NR; r++) {
        float4 v = *(const float4*)(xs + r * 1024 + tid * 4);
        const float nwv[4] = {nw[r].x, nw[r].y, nw[r].z, nw[r].w};
        float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) x[e] = son[r] ? round_act(nwv[e] * round_act(x[e] * rs, act), act) : 0.f;
        *(float4*)(xs + r * 1024 + tid * 4) = make_float4(x[0], x[1], x[2], x[3]);
      }
    } else if (MODE == PRO_GATED2) {
      // gated RMSNorm with the gate and the per-head sums of squares already applied / reduced by the SSM kernel (see build_x_simple)
      const int G = pro.aux > 0 ? pro.aux : 1, gsz = pro.H / G, hpg = pro.aux2 / G;
      const float4 vv = *(const float4*)((const float*)pro.src.p + skc[0]);
      const float4 nw = *(const float4*)(pro.norm_w + skc[0]);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      for (int g = 0; g < G && g < 8; g++) {        // every row wave keeps its own copy of the (<= 8) group factors
        double t = 0.0;                                                     // per-head exact sums arrive as hi + lo floats (k_ssm_step)
        for (int h = lane; h < hpg; h += 64) t += (double)pro.h_in[2 * (g * hpg + h)] + (double)pro.h_in[2 * (g * hpg + h) + 1];
        t = wave_sum_d(t);
        if (lane == 0) grs[wave][g] = rms_scale((float)t, (float)gsz, pro.eps);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      const float rs = grs[wave][min(skc[0] / gsz, 7)];
      const float vs[4] = {vv.x, vv.y, vv.z, vv.w}, nwv[4] = {nw.x, nw.y, nw.z, nw.w};
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; e++) x[e] = son[0] ? round_act(nwv[e] * round_act(vs[e] * rs, act), act) : 0.f;
      *(float4*)(xs + tid * 4) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
      const size_t soff = (size_t)(hb ? slB[0] : slA[0]) * (size_t)slots.src_stride;   // this slot's input row
      typename SrcRaw<FIX>::T a[4], b[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        a[e] = src_raw<FIX>(pro.src.p, soff + skc[0] + e);
        if (MODE == PRO_SILU) b[e] = src_raw<FIX>(pro.src.p, soff + pro.H + skc[0] + e);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = src_cvt<FIX>(a[e], act);
        if (MODE == PRO_SILU) v = round_act(round_act(silu_f(v), act) * src_cvt<FIX>(b[e], act), act);
        x[e] = son[0] ? v : 0.f;
      }
      *(float4*)(xs + tid * 4) = make_float4(x[0], x[1], x[2], x[3]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's part of x is in LDS
    if (lane == 0) atomicAdd(&cnt[1], 1u);
    if (wave == 0) RSTAMP(2);
    if (shift.cs)                                  // the row waves are done: they take the side duty
      for (int i = blockIdx.x * 256 + tid; i < shift.n; i += gridDim.x * 256) conv_shift_one(shift, i, act);
    return;
  }
  // ---- tile waves ---------------------------------------------------------------------------------------
  const int tw = wave - 4;
  struct Stage { RowPiece<WDT> p[4]; };
  const int jrow = ((lane >> 4) & 1) * 2 + (lane >> 5);   // wave_sum4: row (of 16 lanes) -> value index
  __syncthreads();                                 // rendezvous: the row waves' loads are in the queue
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int ua = u0[r] + (int)((long long)(u1[r] - u0[r]) * tw / 8), ue = u0[r] + (int)((long long)(u1[r] - u0[r]) * (tw + 1) / 8);
    size_t eoffA = 0, eoffB = 0;
    if (ROUTE) {
      if (r == 0) { eoffA = (size_t)(route.E + slA[0]) * (size_t)slots.expert_stride; eoffB = (size_t)(route.E + slB[0]) * (size_t)slots.expert_stride; }
      else { if (tw == 0) RSTAMP(4); lds_wait_count(&cnt[2], 1); if (tw == 0) RSTAMP(5); eoffA = (size_t)ssel[slA[1]] * (size_t)slots.expert_stride; eoffB = (size_t)ssel[slB[1]] * (size_t)slots.expert_stride; }
    } else if (slots.sel) { eoffA = (size_t)slots.sel[slA[0]] * (size_t)slots.expert_stride; eoffB = (size_t)slots.sel[slB[0]] * (size_t)slots.expert_stride; }
    const int ubr = ub[r], kA = kcA[r], kB = kcB[r];
    auto issue = [&](Stage& S, int u) {
      const bool second = u >= ubr;
      const int rg = u - (second ? ubr : ubr - NRG);
      const int k = (second ? kB : kA) * 512 + lane * 8;
      const int ko = k < K ? k : 0;                 // x is 0 beyond K
      const size_t eo = second ? eoffB : eoffA;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) S.p[rr] = piece_load<WDT>(W, eo + (size_t)min(4 * rg + rr, N - 1) * K + ko);
    };
    Stage st[4];
    if (ua < ue) {
#pragma unroll
      for (int q = 0; q < 4; q++) if (ua + q < ue) issue(st[q], ua + q);
    }
    if (r == 0) {
      if (zero_buf)                                   // the ring protocol's zeroing duty, by the 512 tile threads
        for (int i = blockIdx.x * 512 + (tid - 256); i < zero_n; i += gridDim.x * 512) zero_buf[i] = 0;
      lds_wait_count(&cnt[1], 4);
    }
    const float* xr = xs + r * 1024;
    const float4 xa0 = *(const float4*)(xr + lane * 8), xa1 = *(const float4*)(xr + lane * 8 + 4);
    const float4 xb0 = *(const float4*)(xr + 512 + lane * 8), xb1 = *(const float4*)(xr + 512 + lane * 8 + 4);

    long long* apA = acc + (size_t)min(sbase[r] + slA[r], slots.acc_slots - 1) * (size_t)slots.acc_stride;
    long long* apB = acc + (size_t)min(sbase[r] + slB[r], slots.acc_slots - 1) * (size_t)slots.acc_stride;
    for (int uc = ua; uc < ue; uc += 4) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int u = uc + q;
        if (u < ue) {
          const bool second = u >= ubr;
          const int row = 4 * (u - (second ? ubr : ubr - NRG)) + jrow;
          if constexpr (EX) {
            const float4 xa = second ? xb0 : xa0, xb = second ? xb1 : xa1;
            // exact sums (piece_dot_x): the unit's partial is the same number whatever the lane / wave decomposition
            const double d0 = piece_dot_x<WDT>(st[q].p[0], xa, xb, 0.0), d1 = piece_dot_x<WDT>(st[q].p[1], xa, xb, 0.0);
            const double d2 = piece_dot_x<WDT>(st[q].p[2], xa, xb, 0.0), d3 = piece_dot_x<WDT>(st[q].p[3], xa, xb, 0.0);
            if (u + 4 < ue) issue(st[q], u + 4);
            double v = wave_sum4_d(d0, d1, d2, d3);
            if ((lane & 15) == 0 && row < N) {
              if (bias && (second ? kB : kA) == 0) v += (double)bias[row];
              atomicAdd((unsigned long long*)((second ? apB : apA) + row), (unsigned long long)d2fix(v, pro.act));
            }
          } else {
            const float4 xa = second ? xb0 : xa0, xb = second ? xb1 : xa1;
            const float d0 = piece_dot<WDT>(st[q].p[0], xa, xb), d1 = piece_dot<WDT>(st[q].p[1], xa, xb);
            const float d2 = piece_dot<WDT>(st[q].p[2], xa, xb), d3 = piece_dot<WDT>(st[q].p[3], xa, xb);
            if (u + 4 < ue) issue(st[q], u + 4);
            float v = wave_sum4(d0, d1, d2, d3);
            if ((lane & 15) == 0 && row < N) {
              if (bias && (second ? kB : kA) == 0) v += bias[row];
              atomicAdd((unsigned long long*)((second ? apB : apA) + row), (unsigned long long)f2fix(v, pro.act));
            }
          }
        }
      }
    }
  }
  if (tw == 0) RSTAMP(6);
#undef RSTAMP
}

// ---------------------------------------------------------------------------------------------------------
// Fused MLP for dense 16-bit weights (Llama-3.2-1B shape): acc_down += Wd[:, slab] . (silu(Wg[slab] x) * (Wu[slab] x)),  x = RMSNorm(h + prev).
// grid = I / 32 workgroups (256 at I = 8192) of 4 row waves + 8 tile waves.  A workgroup owns 32 columns of gate and of up over the FULL K (no split, no
// intermediate round trip): tile wave w streams gate rows 4w..4w+3 and up rows 4w..4w+3 of the slab, 512 k at a time, two chunks of loads in flight, x from LDS;
// its eight sums (wave_sum4 twice) become four activations without leaving the wave.  Then the 32-k slab of down_proj -- stored slab-major at load,
// [I / 32][H][32 k], so that a workgroup's slab is 128 KiB contiguous -- four lanes per output row (16 bytes = 8 k each), 16 rows per wave-wide load, reduced
// over the four lanes, one fixed-point atomic per output row.  One launch instead of two row GEMVs: 16.3 + 9.1 us -> see DESIGN.
// ---------------------------------------------------------------------------------------------------------
template <int WDT>
__global__ void k_repack_down_slabs(const unsigned short* __restrict__ W, int H, int I, unsigned short* __restrict__ out) {   // W [H][I] -> out [I/32][H][32]
  const size_t n = (size_t)H * I;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / I), k = (int)(i % I);
    out[((size_t)(k >> 5) * H + row) * 32 + (k & 31)] = W[i];
  }
}
int bzk_repack_down_slabs(hipStream_t s, const void* w, int H, int I, void* out) {
  hipLaunchKernelGGL(k_repack_down_slabs<BZ_F16>, dim3(2048), dim3(256), 0, s, (const unsigned short*)w, H, I, (unsigned short*)out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

template <int WDT, bool FIX>
__global__ __launch_bounds__(768) void k_mlp_dense(const void* __restrict__ Wgu, const float* __restrict__ bgu, const void* __restrict__ Wds, const float* __restrict__ bd,
                                                   int H, int I, Pro pro, long long* __restrict__ acc, long long* zero_buf, int zero_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = (float*)smem;                      // [H] normalised activations
  float* actl = xs + H;                          // [32] the slab's activations
  double* dred = (double*)(actl + 32);           // [4]
  unsigned* cnt = (unsigned*)(dred + 4);         // [0] sum of squares complete, [1] x published, [2] activations published
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x;
  const int act = pro.act;
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; cnt[2] = 0; }
  if (wave < 4) {
    // ---- row waves: x = R(w R(h' rs)), h' = R(h + prev), the whole row -> LDS (the structure of k_gemv_rows2's norm prologue) ----
    __builtin_amdgcn_s_setprio(3);
    const bool hasprev = pro.src.p != nullptr;
    const void* prevp = hasprev ? pro.src.p : (const void*)pro.h_in;
    double ssd = 0.0;
    bool first = true;
    for (int base = 0; base < H; base += 2048) {
      const int i0 = min(base + tid * 8, H - 8);
      const float4 ha = *(const float4*)(pro.h_in + i0), hb = *(const float4*)(pro.h_in + i0 + 4);
      const float4 na = *(const float4*)(pro.norm_w + i0), nb = *(const float4*)(pro.norm_w + i0 + 4);
      typename SrcRaw<FIX>::T pv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) pv[e] = src_raw<FIX>(prevp, (FIX || hasprev) ? i0 + e : 0);
      if (first) { __builtin_amdgcn_sched_barrier(0); __syncthreads(); first = false; }   // rendezvous: these loads are ahead of the weight stream
      float v[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
      if (hasprev) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = round_act(v[e] + src_cvt<FIX>(pv[e], act), act);
      }
      if (base + tid * 8 < H) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) ssd += (double)(v[e] * v[e]) + (double)(v[e + 1] * v[e + 1]);
        if (blockIdx.x == 0 && pro.h_out) { *(float4*)(pro.h_out + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(pro.h_out + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]); }
        // parked as v * w-independent halves: the weight is applied after rs is known (both factors kept: v here, w in registers of the same thread)
        *(float4*)(xs + i0) = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(xs + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (base + 2048 >= H) {                      // last (usually only) batch: finish here, this thread still holds its norm weights
        ssd = wave_sum_d(ssd);
        if (lane == 0) dred[wave] = ssd;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane == 0) atomicAdd(&cnt[0], 1u);
        lds_wait_count(&cnt[0], 4);
        const float ss = (float)((dred[0] + dred[1]) + (dred[2] + dred[3]));
        const float rs = rms_scale(ss, (float)H, pro.eps);
        for (int b2 = 0; b2 < H; b2 += 2048) {     // every batch of this thread (H <= 2048: exactly the values above)
          const int j0 = b2 + tid * 8;
          if (j0 < H) {
            float4 wa = na, wb = nb;
            if (b2 != base) { wa = *(const float4*)(pro.norm_w + j0); wb = *(const float4*)(pro.norm_w + j0 + 4); }
            const float4 va = *(const float4*)(xs + j0), vb = *(const float4*)(xs + j0 + 4);
            *(float4*)(xs + j0) = make_float4(round_act(wa.x * round_act(va.x * rs, act), act), round_act(wa.y * round_act(va.y * rs, act), act),
                                              round_act(wa.z * round_act(va.z * rs, act), act), round_act(wa.w * round_act(va.w * rs, act), act));
            *(float4*)(xs + j0 + 4) = make_float4(round_act(wb.x * round_act(vb.x * rs, act), act), round_act(wb.y * round_act(vb.y * rs, act), act),
                                                  round_act(wb.z * round_act(vb.z * rs, act), act), round_act(wb.w * round_act(vb.w * rs, act), act));
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) atomicAdd(&cnt[1], 1u);
    return;
  }
  // ---- tile waves -----------------------------------------------------------------------------------------------------
  const int tw = wave - 4;
  const unsigned short* wg = (const unsigned short*)Wgu;
  const int KC = H >> 9;                            // 512-k chunks (H % 512 == 0)
  const int c0 = slab * 32 + tw * 4;                // this wave's four columns: gate rows c0.., up rows I + c0..
  struct Stage { RowPiece<WDT> g[4], u[4]; };
  auto issue = [&](Stage& S, int kc) {
    const size_t ko = (size_t)kc * 512 + lane * 8;
#pragma unroll
    for (int r = 0; r < 4; r++) { S.g[r] = piece_load<WDT>(wg, (size_t)(c0 + r) * H + ko); S.u[r] = piece_load<WDT>(wg, (size_t)(I + c0 + r) * H + ko); }
  };
  __syncthreads();                                  // rendezvous: the row waves' loads are in the queue
  Stage st[2];
  issue(st[0], 0);
  if (KC > 1) issue(st[1], 1);
  if (zero_buf)
    for (int i = blockIdx.x * 512 + (tid - 256); i < zero_n; i += gridDim.x * 512) zero_buf[i] = 0;
  lds_wait_count(&cnt[1], 4);
  double ag[4] = {0.0, 0.0, 0.0, 0.0}, au[4] = {0.0, 0.0, 0.0, 0.0};     // exact sums (piece_dot_d)
  for (int kc = 0; kc < KC; kc += 2) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (kc + q < KC) {
        const float4 xa = *(const float4*)(xs + (kc + q) * 512 + lane * 8), xb = *(const float4*)(xs + (kc + q) * 512 + lane * 8 + 4);
#pragma unroll
        for (int r = 0; r < 4; r++) { ag[r] = piece_dot_x<WDT>(st[q].g[r], xa, xb, ag[r]); au[r] = piece_dot_x<WDT>(st[q].u[r], xa, xb, au[r]); }
        if (kc + q + 2 < KC) issue(st[q], kc + q + 2);
      }
    }
  }
  // the down slab's loads go out now: rows 256 tw .. 256 tw + 255 of the output (H = 2048: 16 wave-wide loads of 16 rows), under the reductions below.
  // (Requesting them earlier -- behind the last two gate / up chunks, with the stage registers still live -- was measured: 30.6 vs 23.8 us.)
  const unsigned short* wd = (const unsigned short*)Wds + (size_t)slab * H * 32;
  const int NLD = H / (8 * 16);                     // loads per wave: H rows / 8 waves / 16 rows per load
  const int n0 = tw * (H / 8) + (lane >> 2), ksub = (lane & 3) * 8;
  uint4 D[16];
#pragma unroll
  for (int t = 0; t < 16; t++) if (t < NLD) D[t] = ldnt((const uint4*)(wd + (size_t)(n0 + t * 16) * 32 + ksub));
  {
    double sgd = wave_sum4_d(ag[0], ag[1], ag[2], ag[3]), sud = wave_sum4_d(au[0], au[1], au[2], au[3]);   // value j in 16-lane row [0, 2, 1, 3][j]
    const int j = ((lane >> 4) & 1) * 2 + (lane >> 5);
    if (bgu) { sgd += (double)bgu[c0 + j]; sud += (double)bgu[I + c0 + j]; }
    const float g = (float)sgd, u = (float)sud;       // ONE rounding of the exact sum to f32, as the oracle
    const float a = round_act(round_act(silu_f(round_act(g, act)), act) * round_act(u, act), act);
    if ((lane & 15) == 0) actl[tw * 4 + j] = a;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) atomicAdd(&cnt[2], 1u);
  lds_wait_count(&cnt[2], 8);
  const float4 aa = *(const float4*)(actl + ksub), ab = *(const float4*)(actl + ksub + 4);
#pragma unroll
  for (int t = 0; t < 16; t++) {
    if (t < NLD) {
      RowPiece<WDT> pc; pc.a = D[t];
      double d = piece_dot_x<WDT>(pc, aa, ab, 0.0);
      d += dpp_get<DPP_XOR1>(d); d += dpp_get<DPP_XOR2>(d);      // the row's four lanes
      const int n = n0 + t * 16;
      if ((lane & 3) == 0) {
        if (bd != nullptr && slab == 0) d += (double)bd[n];
        atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(d, pro.act));
      }
    }
  }
}

bool bzk_mlp_dense_fusable(const LinearDev& gu, const LinearDev& dn, const void* down_slabs, int H, int I, int act) {
  static const bool off = getenv("BZ_NO_MLP_FUSION") != nullptr;
  return !off && down_slabs != nullptr && gu.kind == LK_ROWS && dn.kind == LK_ROWS && gu.wdt == dn.wdt && (gu.wdt == BZ_F16 || gu.wdt == BZ_BF16) && gu.N == 2 * I &&
         gu.K == H && dn.N == H && dn.K == I && H % 512 == 0 && H >= 1024 && H <= 2048 && I % 32 == 0 && I / 32 >= 128 && gu.sk > 1 && dn.sk > 1 && (act == BZ_F16 || act == BZ_BF16 || act == BZ_F32);
}
int bzk_mlp_dense(hipStream_t s, const LinearDev& gu, const LinearDev& dn, const void* down_slabs, int H, int I, const Pro& pro, long long* acc, long long* zero_buf, int zero_n) {
  if (!bzk_mlp_dense_fusable(gu, dn, down_slabs, H, I, pro.act) || pro.mode != PRO_NORM || pro.H != H) BZ_FAIL(BZ_E_INVALID, "dense fused MLP does not apply to this shape");
  const size_t smem = (size_t)H * 4 + 32 * 4 + 4 * 8 + 64;
  const double bytes = (double)gu.algo_bytes + (double)dn.algo_bytes;
#define LAUNCH_MD(DT, FX) BZ_LAUNCH("mlp_dense<norm+gate/up+silu+down>", bytes, (k_mlp_dense<DT, FX>), dim3(I / 32), dim3(768), smem, s, gu.w, gu.bias, down_slabs, dn.bias, H, I, pro, \
    acc, zero_buf, zero_n)
#define LAUNCH_MD_F(DT) do { if (pro.src.fix) LAUNCH_MD(DT, true); else LAUNCH_MD(DT, false); } while (0)
  if (gu.wdt == BZ_F16) LAUNCH_MD_F(BZ_F16); else LAUNCH_MD_F(BZ_BF16);
#undef LAUNCH_MD_F
#undef LAUNCH_MD
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// MoE grouped GEMV: blockIdx.y = expert slot.  The expert id comes from the router's device-side selection, so the whole MoE
// layer stays capturable in a hipGraph.  Stacked weights [E + n_shared][N][K]; per-slot prologue source / output offsets.
template <int WDT, bool SPLIT>
__global__ __launch_bounds__(256) void k_moe_rows(MoeGemvArgs g, Pro pro, int act) {
  const int slot = blockIdx.y;
  const int e = g.sel[slot];
  const size_t es = WDT == BZ_F32 ? 4 : 2;
  const void* W = (const char*)g.w + (size_t)e * (size_t)g.expert_stride * es;
  pro.src.p = (const float*)pro.src.p + (size_t)slot * g.src_stride;
  float* out = SPLIT ? nullptr : g.out + (size_t)slot * g.out_stride;
  long long* acc = SPLIT ? g.acc + (size_t)min(slot, g.acc_slots - 1) * g.acc_stride : nullptr;
  rows_body<WDT, SPLIT>(W, nullptr, g.N, g.K, 16, pro, out, act, nullptr, nullptr, nullptr, 0, 1, acc, blockIdx.x);
}

static int rows_per_wg_for(int N) {
  // ~1000 workgroups when N is large, at least 16 rows (4 per wave) per workgroup
  int r = 16;
  while (r < 256 && (N + r - 1) / r > 1024) r <<= 1;
  return r;
}
static bool rows2_enabled() { static const bool off = getenv("BZ_NO_ROWS2") != nullptr; return !off; }
// split count for a dense [N,K] GEMV: enough (16-row x K-slice) workgroups to fill the chip when N alone is too small
int bzk_rows_choose_sk(int N, int K) {
  const int rbs = (N + 15) / 16, KC = (K + 511) / 512;
  if (rows2_enabled() && K % 8 == 0 && K <= 131072) {   // fixed-point output whatever the shape: k_gemv_rows2 balances any [N, K] over the chip
    if (rbs >= 512 || KC < 8) return 2;                // (the value only matters to the fallback kernel)
  } else if (rbs >= 512 || KC < 8) return 1;
  int sk = std::min((512 + rbs - 1) / rbs, KC / 4);   // a slice keeps >= 4 chunks (2048 k): below that the prologue dominates
  const int kcs = (KC + sk - 1) / sk;
  sk = (KC + kcs - 1) / kcs;
  return sk;
}
// the balanced role kernel (k_gemv_rows2) takes every fixed-point-output dense GEMV with 16-bit weights
static bool bzk_rows2_ok(const LinearDev& L, const Pro& pro, const GemvOut& out) {
  if (!rows2_enabled() || L.kind != LK_ROWS || L.sk <= 1 || !out.acc || out.amax_val) return false;
  if (L.wdt != BZ_F16 && L.wdt != BZ_BF16) return false;
  if (L.K % 8 || L.K > 131072 || pro.perm) return false;
  if (pro.mode == PRO_NORM) return pro.H == L.K && pro.H % 8 == 0;
  if (pro.mode == PRO_GATED2) return !pro.src.fix && (pro.aux <= 8) && pro.H == L.K && (pro.H / (pro.aux > 0 ? pro.aux : 1)) % 4 == 0;
  return pro.mode == PRO_PLAIN || pro.mode == PRO_SILU;
}
int bzk_gemv_rows_blocks(const LinearDev& L) { int r = rows_per_wg_for(L.N); return (L.N + r - 1) / r; }

// ---- GGUF: load-time repack from raw row-major ggml blocks ([N][K/blk] blocks of 34 / 144 / 210 bytes) -----------
__device__ __forceinline__ unsigned q4k_raw_nib(const unsigned char* blk, int k) {   // weight k (0..255) of a raw block_q4_K
  const int j64 = k >> 6, l = k & 63;
  const unsigned char b = blk[16 + j64 * 32 + (l & 31)];
  return l < 32 ? (b & 15u) : (unsigned)(b >> 4);
}
__device__ __forceinline__ unsigned q6k_raw(const unsigned char* blk, int k) {       // 6-bit value (0..63) of weight k of a raw block_q6_K
  const int n128 = k >> 7, l = k & 127, quad = l >> 5, pos = l & 31;
  const unsigned char qlb = blk[n128 * 64 + (quad & 1) * 32 + pos];
  const unsigned lo = quad < 2 ? (qlb & 15u) : (unsigned)(qlb >> 4);
  const unsigned hi = (blk[128 + n128 * 32 + pos] >> (2 * quad)) & 3u;
  return lo | (hi << 4);
}

__global__ void k_repack_gq(int fmt, const unsigned char* raw, int N, int K, uint32_t* wq, uint32_t* wh, uint32_t* hd, __half* dd) {
  const size_t gsz = (size_t)gridDim.x * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fmt == GQ_Q80) {
    const size_t rb = (size_t)(K / 32) * 34;
    const size_t total = (size_t)N * (K >> 2);                 // output words (4 int8 each)
    for (size_t idx = gid; idx < total; idx += gsz) {
      const int j = idx & 3, lane = (idx >> 2) & 63;
      const size_t t = idx >> 8;
      const int kc = (int)(t % (size_t)(K >> 4)), nt = (int)(t / (size_t)(K >> 4));
      const int n = nt * 64 + lane, k = kc * 16 + j * 4;
      const unsigned char* b = raw + (size_t)n * rb + (size_t)(k >> 5) * 34 + 2 + (k & 31);
      wq[idx] = (unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)b[2] << 16) | ((unsigned)b[3] << 24);
    }
    const size_t nb = (size_t)N * (K >> 5);
    for (size_t idx = gid; idx < nb; idx += gsz) {
      const int lane = idx & 63;
      const size_t t = idx >> 6;
      const int b = (int)(t % (size_t)(K >> 5)), nt = (int)(t / (size_t)(K >> 5));
      const unsigned char* p = raw + (size_t)(nt * 64 + lane) * rb + (size_t)b * 34;
      dd[idx] = __ushort_as_half((unsigned short)(p[0] | (p[1] << 8)));
    }
    return;
  }
  const size_t bsz = fmt == GQ_Q4K ? 144 : 210;
  const size_t rb = (size_t)(K / 256) * bsz;
  const size_t total = (size_t)N * (K >> 3);                   // nibble words (8 weights each)
  for (size_t idx = gid; idx < total; idx += gsz) {
    const int j = idx & 3, lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int kc = (int)(t % (size_t)(K >> 5)), nt = (int)(t / (size_t)(K >> 5));
    const int n = nt * 64 + lane;
    const unsigned char* blk = raw + (size_t)n * rb + (size_t)(kc >> 3) * bsz;
    const int kb = (kc & 7) * 32 + j * 8;
    unsigned word = 0;
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      if (fmt == GQ_Q4K) {
        const unsigned q1 = q4k_raw_nib(blk, kb + bb), q2 = q4k_raw_nib(blk, kb + 4 + bb);
        word |= (q1 | ((q2 ^ 8u) << 4)) << (8 * bb);
      } else {
        const unsigned q1 = q6k_raw(blk, kb + bb) & 15u, q2 = q6k_raw(blk, kb + 4 + bb) & 15u;
        word |= (q1 | (q2 << 4)) << (8 * bb);
      }
    }
    wq[idx] = word;
  }
  if (fmt == GQ_Q6K) {
    const size_t th = (size_t)N * (K >> 5) * 2;               // two high-bit words per (chunk, lane)
    for (size_t idx = gid; idx < th; idx += gsz) {
      const int hw = idx & 1, lane = (idx >> 1) & 63;
      const size_t t = idx >> 7;
      const int kc = (int)(t % (size_t)(K >> 5)), nt = (int)(t / (size_t)(K >> 5));
      const unsigned char* blk = raw + (size_t)(nt * 64 + lane) * rb + (size_t)(kc >> 3) * bsz;
      unsigned word = 0;
      for (int f = 0; f < 4; f++) {            // field F = hw*4 + f  <-> k-order word F: k = 4 F + b
        const int F = hw * 4 + f;
        for (int bb = 0; bb < 4; bb++) {
          const unsigned hi = q6k_raw(blk, (kc & 7) * 32 + 4 * F + bb) >> 4;
          word |= hi << (8 * bb + 2 * f);
        }
      }
      wh[idx] = word;
    }
  }
  const size_t nsb = (size_t)N * (K >> 8);
  for (size_t idx = gid; idx < nsb * 4; idx += gsz) {          // 16-byte header per (superblock, lane)
    const int wi = idx & 3, lane = (idx >> 2) & 63;
    const size_t t = idx >> 8;
    const int sb = (int)(t % (size_t)(K >> 8)), nt = (int)(t / (size_t)(K >> 8));
    const unsigned char* blk = raw + (size_t)(nt * 64 + lane) * rb + (size_t)sb * bsz;
    const unsigned char* src = fmt == GQ_Q4K ? blk + wi * 4 : blk + 192 + wi * 4;   // Q4_K: d,dmin,scales[12] ; Q6_K: scales[16]
    hd[idx] = (unsigned)src[0] | ((unsigned)src[1] << 8) | ((unsigned)src[2] << 16) | ((unsigned)src[3] << 24);
    if (fmt == GQ_Q6K && wi == 0) dd[((size_t)nt * (K >> 8) + sb) * 64 + lane] = __ushort_as_half((unsigned short)(blk[208] | (blk[209] << 8)));
  }
}
int bzk_repack_gq(hipStream_t s, int kind, const void* raw, int N, int K, void* wq, void* wh, void* hd, void* dd) {
  const int fmt = kind == LK_Q80 ? GQ_Q80 : (kind == LK_Q4K ? GQ_Q4K : GQ_Q6K);
  hipLaunchKernelGGL(k_repack_gq, dim3(2048), dim3(256), 0, s, fmt, (const unsigned char*)raw, N, K, (uint32_t*)wq, (uint32_t*)wh, (uint32_t*)hd, (__half*)dd);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// one weight of the REPACKED layouts as the f32 value the decode kernels' arithmetic stands for (ggml's dequantisation formula per format)
__device__ __forceinline__ float gq_elem(int fmt, const uint32_t* wq, const uint32_t* wh, const uint32_t* hd, const __half* dd, int K, int n, int k) {
  const int nt = n >> 6, lane = n & 63;
  if (fmt == GQ_Q80) {
    const unsigned word = wq[(((size_t)nt * (K >> 4) + (k >> 4)) * 64 + lane) * 4 + ((k & 15) >> 2)];
    const int q = (int)(signed char)((word >> (8 * (k & 3))) & 255u);
    return __half2float(dd[((size_t)nt * (K >> 5) + (k >> 5)) * 64 + lane]) * (float)q;
  }
  const int kc = k >> 5, r = k & 31, j = r >> 3, rr = r & 7;
  const unsigned word = wq[(((size_t)nt * (K >> 5) + kc) * 64 + lane) * 4 + j];
  const unsigned byte = (word >> (8 * (rr & 3))) & 255u;
  const size_t hi = (((size_t)nt * (K >> 8) + (k >> 8)) * 64 + lane) * 4;
  const unsigned hw[4] = {hd[hi], hd[hi + 1], hd[hi + 2], hd[hi + 3]};
  if (fmt == GQ_Q4K) {
    const float q = rr < 4 ? (float)(byte & 15u) : (float)((byte >> 4) ^ 8u);
    const float d = __half2float(__ushort_as_half((unsigned short)(hw[0] & 0xffffu))), dmin = __half2float(__ushort_as_half((unsigned short)(hw[0] >> 16)));
    int sc, mn;
    q4k_scale_min(hw, (k & 255) >> 5, sc, mn);
    return (d * (float)sc) * q - (dmin * (float)mn);
  }
  const unsigned lo = rr < 4 ? (byte & 15u) : (byte >> 4);
  const int F = r >> 2;   // k-order word
  const unsigned hword = wh[(((size_t)nt * (K >> 5) + kc) * 64 + lane) * 2 + (F >> 2)];
  const unsigned hi2 = (hword >> (8 * (r & 3) + 2 * (F & 3))) & 3u;
  const int q = (int)(lo | (hi2 << 4)) - 32;
  const int si = (k & 255) >> 4;
  const int sc = (int)(signed char)((hw[si >> 2] >> (8 * (si & 3))) & 255u);
  return __half2float(dd[((size_t)nt * (K >> 8) + (k >> 8)) * 64 + lane]) * (float)sc * (float)q;
}
// dequantise the REPACKED layouts to f32 [N][K] (validates the repack against the oracle's ggml dequant)
__global__ void k_dequant_gq(int fmt, const uint32_t* wq, const uint32_t* wh, const uint32_t* hd, const __half* dd, int N, int K, float* out) {
  const size_t total = (size_t)N * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    out[idx] = gq_elem(fmt, wq, wh, hd, dd, K, (int)(idx / (size_t)K), (int)(idx % (size_t)K));
}
static int gq_fmt(const LinearDev& L) { return L.kind == LK_Q80 ? GQ_Q80 : (L.kind == LK_Q4K ? GQ_Q4K : GQ_Q6K); }
int bzk_dequant_gq(hipStream_t s, const LinearDev& L, float* out) {
  hipLaunchKernelGGL(k_dequant_gq, dim3(2048), dim3(256), 0, s, gq_fmt(L), (const uint32_t*)L.w, (const uint32_t*)L.zeros, (const uint32_t*)L.hdr,
                     (const __half*)L.scales, L.N, L.K, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---- the batched prompt path of the block formats (bz_prefill.hip "f32-activation rows on the 16-bit matrix cores") -----------------------------------
// largest |w| of a matrix (bits of a non-negative float order like unsigned integers): *amax must be zeroed first
__global__ void k_gq_absmax(int fmt, const uint32_t* wq, const uint32_t* wh, const uint32_t* hd, const __half* dd, int N, int K, unsigned* amax) {
  const size_t total = (size_t)N * K;
  float am = 0.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    am = fmaxf(am, fabsf(gq_elem(fmt, wq, wh, hd, dd, K, (int)(idx / (size_t)K), (int)(idx % (size_t)K))));
  am = wave_max(am);
  if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(am));
}
// *amax (float bits) -> the power of two that brings it into [2^13, 2^14)
__global__ void k_gq_wscale(const unsigned* amax, float* wscale) {
  const unsigned eb = (*amax >> 23) & 255u;
  *wscale = (eb == 0u || eb == 255u || eb > 240u || eb < 20u) ? 1.0f : __uint_as_float((127u + 13u + 127u - eb) << 23);
}
// W' rows [row0 + n][3 K] = [ wh | wh 2^-11 | wl 2^11 ] of w 2^e (f16 pieces; see k_pf_split3).  A thread takes 8 consecutive k of one column: three 16-byte stores.
__global__ __launch_bounds__(256) void k_gq_split3(int fmt, const uint32_t* wq, const uint32_t* wh, const uint32_t* hd, const __half* dd, int N, int K, const float* wscale,
                                                   unsigned short* out, int row0) {
  const size_t total = (size_t)N * (K >> 3);
  const float sc = *wscale;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / (size_t)(K >> 3)), k0 = (int)(idx % (size_t)(K >> 3)) * 8;
    unsigned short a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float vs = gq_elem(fmt, wq, wh, hd, dd, K, n, k0 + e) * sc;
      const __half h = f16_cvt(vs);
      const float hf = __half2float(h), l = vs - hf;
      a[e] = __half_as_ushort(h); b[e] = __half_as_ushort(f16_cvt(hf * (1.0f / 2048.0f))); c[e] = __half_as_ushort(f16_cvt(l * 2048.0f));
    }
    unsigned short* o = out + (size_t)(row0 + n) * 3 * K + k0;
    *(uint4*)o = make_uint4(a[0] | ((unsigned)a[1] << 16), a[2] | ((unsigned)a[3] << 16), a[4] | ((unsigned)a[5] << 16), a[6] | ((unsigned)a[7] << 16));
    *(uint4*)(o + K) = make_uint4(b[0] | ((unsigned)b[1] << 16), b[2] | ((unsigned)b[3] << 16), b[4] | ((unsigned)b[5] << 16), b[6] | ((unsigned)b[7] << 16));
    *(uint4*)(o + 2 * K) = make_uint4(c[0] | ((unsigned)c[1] << 16), c[2] | ((unsigned)c[3] << 16), c[4] | ((unsigned)c[5] << 16), c[6] | ((unsigned)c[7] << 16));
  }
}
bool bzk_gq_split_ok(const LinearDev& L) { return (L.kind == LK_Q80 || L.kind == LK_Q4K || L.kind == LK_Q6K) && L.K % 64 == 0 && L.N % 64 == 0; }
// amax: device word holding the running maximum of the fused linear's parts (zeroed by the caller before the first part)
int bzk_gq_absmax(hipStream_t s, const LinearDev& L, unsigned* amax) {
  hipLaunchKernelGGL(k_gq_absmax, dim3(2048), dim3(256), 0, s, gq_fmt(L), (const uint32_t*)L.w, (const uint32_t*)L.zeros, (const uint32_t*)L.hdr, (const __half*)L.scales, L.N, L.K, amax);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_gq_wscale(hipStream_t s, const unsigned* amax, float* wscale) {
  hipLaunchKernelGGL(k_gq_wscale, dim3(1), dim3(1), 0, s, amax, wscale);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_gq_split3(hipStream_t s, const LinearDev& L, const float* wscale, void* out, int row0) {
  hipLaunchKernelGGL(k_gq_split3, dim3(4096), dim3(256), 0, s, gq_fmt(L), (const uint32_t*)L.w, (const uint32_t*)L.zeros, (const uint32_t*)L.hdr, (const __half*)L.scales, L.N, L.K, wscale,
                     (unsigned short*)out, row0);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// partial argmax over a plain f32 vector (quantised lm_head path): nb blocks -> pval/pidx
int bzk_argmax_partials(hipStream_t s, const float* v, long long n, float* pval, int* pidx, int nb);

int bzk_gemv(hipStream_t s, const LinearDev& L, const Pro& pro, const GemvOut& out, int act) {
  if (L.kind == LK_Q4G && act == BZ_F16 && bzk_gemv_slim_ok(L, pro)) {
    if (!out.acc) BZ_FAIL(BZ_E_INVALID, "int4 gemv needs a fixed-point accumulator");
    const int nks = L.K / 256, ntg = (L.N / 64 + 7) / 8;
#define LAUNCH_SLIM(FIX, NJ, DG) BZ_LAUNCH("gemv_q4g<norm>", L.algo_bytes, (k_gemv_q4g_slim<FIX, NJ, BZ_F16, DG>), dim3(nks * ntg), dim3(768), 0, s, (const uint4*)L.w, \
    (const __half*)L.scales, (const unsigned char*)L.zeros, L.bias, L.N, L.K, pro, out.acc, out.zero_buf, out.zero_n)
#define LAUNCH_SLIM_NJ(FIX) do { if (L.K == 2048) LAUNCH_SLIM(FIX, 1, 0); else if (L.K == 4096) LAUNCH_SLIM(FIX, 2, 0); else LAUNCH_SLIM(FIX, 4, 0); } while (0)
    if (pro.stamps && L.K == 4096 && pro.src.fix) LAUNCH_SLIM(1, 2, 1);   // diagnostic build (bz_tune_gemv flag 16)
    else if (pro.src.fix) LAUNCH_SLIM_NJ(1); else LAUNCH_SLIM_NJ(0);
#undef LAUNCH_SLIM_NJ
#undef LAUNCH_SLIM
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
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
#define LAUNCH_Q4G_N(MODE, FIX, MJ) do { if (GW == 1) LAUNCH_Q4G(MODE, FIX, MJ, 1); else if (L.npf >= 4 && GW >= 4) LAUNCH_Q4G(MODE, FIX, MJ, 4); \
                                          else LAUNCH_Q4G(MODE, FIX, MJ, 2); } while (0)
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
  if (L.kind == LK_ROWS && bzk_rows2_ok(L, pro, out)) {
    const long long U = (long long)((L.N + 3) / 4) * ((L.K + 511) / 512);
    const int KC = (L.K + 511) / 512;
    const int nb = (int)std::max<long long>(std::min<long long>(256, U), KC);   // one workgroup per CU; a range lies in <= 2 chunks when nb >= K / 512
    const char* lbl = pro.mode == PRO_NORM ? "gemv_rows2<norm>" : pro.mode == PRO_GATED2 ? "gemv_rows2<gated>" : pro.mode == PRO_SILU ? "gemv_rows2<silu>" : "gemv_rows2";
#define LAUNCH_R2(DT, MODE, FIX) do { if (pro.f32_sums) LAUNCH_R2X(DT, MODE, FIX, false); else LAUNCH_R2X(DT, MODE, FIX, true); } while (0)
#define LAUNCH_R2X(DT, MODE, FIX, EX_) BZ_LAUNCH(lbl, L.algo_bytes, (k_gemv_rows2<DT, MODE, FIX, 0, EX_>), dim3(nb), dim3(768), 0, s, (const void*)L.w, L.bias, L.N, L.K, pro, \
    out.acc, out.zero_buf, out.zero_n, out.shift, MoeSlots{nullptr, 0, 1, 0, 0, 1}, RouteArgs{})
#define LAUNCH_R2_F(DT, MODE) do { if (pro.src.fix) LAUNCH_R2(DT, MODE, true); else LAUNCH_R2(DT, MODE, false); } while (0)
#define LAUNCH_R2_M(DT) do { if (pro.mode == PRO_NORM) LAUNCH_R2_F(DT, PRO_NORM); else if (pro.mode == PRO_SILU) LAUNCH_R2_F(DT, PRO_SILU); \
    else if (pro.mode == PRO_GATED2) LAUNCH_R2(DT, PRO_GATED2, false); else LAUNCH_R2_F(DT, PRO_PLAIN); } while (0)
    if (L.wdt == BZ_F16) LAUNCH_R2_M(BZ_F16); else LAUNCH_R2_M(BZ_BF16);
#undef LAUNCH_R2_M
#undef LAUNCH_R2_F
#undef LAUNCH_R2
#undef LAUNCH_R2X
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  if (L.kind == LK_ROWS) {
    const int SK = L.sk > 1 ? L.sk : 1;
    if (out.shift.cs) { hipLaunchKernelGGL(k_conv_shift, dim3((out.shift.n + 255) / 256), dim3(256), 0, s, out.shift, act); BZ_HIP(hipGetLastError()); }   // (k_gemv_rows2 does it in-launch)
    if (SK == 1 && !out.direct) BZ_FAIL(BZ_E_INVALID, "rows gemv needs a direct output");
    if (SK > 1 && (!out.acc || out.amax_val)) BZ_FAIL(BZ_E_INVALID, "split-K rows gemv needs a fixed-point accumulator");
    const int rpw = SK > 1 ? 16 : rows_per_wg_for(L.N);
    const int grid = ((L.N + rpw - 1) / rpw) * SK;
    const int KP = (L.K + 511) & ~511;
    const size_t smem = (size_t)KP * 4 + 64;
    if (smem > 160 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "K=%d too large for the rows GEMV", L.K);
    const char* lbl = out.amax_val ? "gemv_rows<lm_head+argmax>" : pro.mode == PRO_NORM ? "gemv_rows<norm>" : (pro.mode == PRO_GATED || pro.mode == PRO_GATED2) ? "gemv_rows<gated>"
                      : pro.mode == PRO_SILU ? "gemv_rows<silu>" : "gemv_rows";
#define LAUNCH_ROWS1(DT, SP) BZ_LAUNCH(lbl, L.algo_bytes, (k_gemv_rows<DT, SP>), dim3(grid), \
    dim3(256), smem, s, (const void*)L.w, L.bias, L.N, L.K, rpw, pro, out.direct, act, out.amax_val, out.amax_idx, out.zero_buf, out.zero_n, SK, out.acc)
#define LAUNCH_ROWS(DT) do { if (SK > 1) LAUNCH_ROWS1(DT, true); else LAUNCH_ROWS1(DT, false); } while (0)
    if (L.wdt == BZ_F16) LAUNCH_ROWS(BZ_F16); else if (L.wdt == BZ_BF16) LAUNCH_ROWS(BZ_BF16); else LAUNCH_ROWS(BZ_F32);
#undef LAUNCH_ROWS
#undef LAUNCH_ROWS1
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  if ((L.kind == LK_Q4K || L.kind == LK_Q6K) && bzk_gq_slim_ok(L, pro)) {
    if (!out.acc) BZ_FAIL(BZ_E_INVALID, "block-quant gemv needs a fixed-point accumulator");
    const int nsb = L.K / 256, ntg = (L.N / 64 + 7) / 8;
    const char* label = L.kind == LK_Q4K ? "gemv_q4_K<slim>" : "gemv_q6_K<slim>";
#define LAUNCH_GQS(FMT, MODE, FIX, NJ) BZ_LAUNCH(label, L.algo_bytes, (k_gemv_gq_slim<FMT, MODE, FIX, NJ>), dim3(nsb * ntg), dim3(512), 0, s, (const uint4*)L.w, \
    (const uint2*)L.zeros, (const uint4*)L.hdr, (const __half*)L.scales, L.bias, L.N, L.K, pro, out.acc, out.zero_buf, out.zero_n, GqMix{})
#define LAUNCH_GQS_NJ(FMT, FIX) do { if (pro.mode == PRO_SILU) LAUNCH_GQS(FMT, PRO_SILU, FIX, 1); else if (L.K == 2048) LAUNCH_GQS(FMT, PRO_NORM, FIX, 1); \
    else if (L.K == 4096) LAUNCH_GQS(FMT, PRO_NORM, FIX, 2); else LAUNCH_GQS(FMT, PRO_NORM, FIX, 4); } while (0)
#define LAUNCH_GQS_F(FMT) do { if (pro.src.fix) LAUNCH_GQS_NJ(FMT, 1); else LAUNCH_GQS_NJ(FMT, 0); } while (0)
    if (L.kind == LK_Q4K) LAUNCH_GQS_F(GQ_Q4K); else LAUNCH_GQS_F(GQ_Q6K);
#undef LAUNCH_GQS_F
#undef LAUNCH_GQS_NJ
#undef LAUNCH_GQS
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  if (L.kind == LK_Q80 || L.kind == LK_Q4K || L.kind == LK_Q6K) {
    if (!out.acc) BZ_FAIL(BZ_E_INVALID, "block-quant gemv needs a fixed-point accumulator");
    const int SB = L.K / 256, SBW = L.gw;
    if (SBW <= 0 || SB % SBW || SBW > 8) BZ_FAIL(BZ_E_INVALID, "block-quant gemv: bad k-slice %d of %d superblocks", SBW, SB);
    const int nst = (L.N + 255) / 256;
    const int grid = nst * (SB / SBW);
    const size_t smem = gq_smem(SBW);
    const int maxj = pro.mode == PRO_NORM ? (pro.H + 1023) / 1024 : 1;
    const char* label = L.kind == LK_Q80 ? "gemv_q8_0" : (L.kind == LK_Q4K ? "gemv_q4_K" : "gemv_q6_K");
#define LAUNCH_GQ(FMT, MODE, FIX, MJ) BZ_LAUNCH(label, L.algo_bytes, (k_gemv_gq<FMT, MODE, FIX, MJ>), dim3(grid), dim3(256), smem, s, (const uint4*)L.w, \
    (const uint2*)L.zeros, (const uint4*)L.hdr, (const __half*)L.scales, L.bias, L.N, L.K, SBW, nst, pro, out.acc, out.zero_buf, out.zero_n)
#define LAUNCH_GQ_F(FMT, MODE, MJ) do { if (pro.src.fix) LAUNCH_GQ(FMT, MODE, 1, MJ); else LAUNCH_GQ(FMT, MODE, 0, MJ); } while (0)
#define LAUNCH_GQ_M(FMT) do { if (pro.mode == PRO_PLAIN) LAUNCH_GQ_F(FMT, PRO_PLAIN, 1); else if (pro.mode == PRO_SILU) LAUNCH_GQ_F(FMT, PRO_SILU, 1); \
    else if (maxj <= 1) LAUNCH_GQ_F(FMT, PRO_NORM, 1); else if (maxj <= 2) LAUNCH_GQ_F(FMT, PRO_NORM, 2); else if (maxj <= 4) LAUNCH_GQ_F(FMT, PRO_NORM, 4); \
    else if (maxj <= 8) LAUNCH_GQ_F(FMT, PRO_NORM, 8); else BZ_FAIL(BZ_E_UNSUPPORTED, "hidden size %d too large for the fused norm prologue", pro.H); } while (0)
    if (L.kind == LK_Q80) LAUNCH_GQ_M(GQ_Q80); else if (L.kind == LK_Q4K) LAUNCH_GQ_M(GQ_Q4K); else LAUNCH_GQ_M(GQ_Q6K);
#undef LAUNCH_GQ_M
#undef LAUNCH_GQ_F
#undef LAUNCH_GQ
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
      word |= ((q1 ^ 8u) | ((q2 ^ 8u) << 4)) << (8 * bb);     // both nibbles as two's-complement (q - 8): V_DOT8_I32_I4 operands
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
    sout[idx] = f16_cvt(sc[(size_t)g * N + n]);
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
      word |= ((q1 ^ 8u) | ((q2 ^ 8u) << 4)) << (8 * bb);     // both nibbles as two's-complement (q - 8): V_DOT8_I32_I4 operands
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
    sout[idx] = f16_cvt(sc[(size_t)g * N + n]);
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
    const float q = (r < 4) ? (float)((byte & 15u) ^ 8u) : (float)((byte >> 4) ^ 8u);
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
// embedding row of the new token; block 0 also stages the RoPE row of the new position at a FIXED address ([cos half | sin half]), so that the
// attention kernels' cos/sin loads do not depend on the position word (one memory latency off every layer's critical path)
__global__ void k_embed(const void* table, int tdt, const long long* tok, int H, int act, float* h, const int* pos, const float* cos_t,
                        const float* sin_t, int half, float* rope_cur) {
  if (rope_cur != nullptr && blockIdx.x == 0 && threadIdx.x < 2 * half) {
    const int p = pos[0], i = threadIdx.x % half;
    rope_cur[threadIdx.x] = threadIdx.x < half ? cos_t[(size_t)p * half + i] : sin_t[(size_t)p * half + i];
  }
  if (table == nullptr) return;
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
int bzk_embed(hipStream_t s, const void* table, int tdt, const long long* tok, int H, int act, float* h, const int* pos, const float* cos_t,
              const float* sin_t, int half, float* rope_cur) {
  if (rope_cur != nullptr && (2 * half > 256 || !pos || !cos_t || !sin_t)) BZ_FAIL(BZ_E_INVALID, "embed: bad RoPE staging arguments");
  BZ_LAUNCH("embed", (double)H * (tdt == BZ_F32 ? 4 : 2), k_embed, dim3((H + 255) / 256), dim3(256), 0, s, table, tdt, tok, H, act, h, pos, cos_t, sin_t,
            half, rope_cur);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_rope_row(hipStream_t s, const int* pos, const float* cos_t, const float* sin_t, int half, float* rope_cur) {
  if (2 * half > 256 || !pos || !cos_t || !sin_t || !rope_cur) BZ_FAIL(BZ_E_INVALID, "rope_row: bad arguments");
  hipLaunchKernelGGL(k_embed, dim3(1), dim3(256), 0, s, (const void*)nullptr, 0, (const long long*)nullptr, 0, 0, (float*)nullptr, pos, cos_t, sin_t, half, rope_cur);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

__global__ void k_fix_to_f32(const long long* acc, int n, int act, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = round_act(fix2f(acc[i], act), act);
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
__device__ __forceinline__ float kv_ld(const void* base, size_t off, int dt) {
  if (dt == BZ_F16) return __half2float(((const __half*)base)[off]);
  if (dt == BZ_BF16) return __uint_as_float((unsigned)((const unsigned short*)base)[off] << 16);
  return ((const float*)base)[off];
}
__device__ __forceinline__ void kv_st(void* base, size_t off, int dt, float v) {
  if (dt == BZ_F16) ((__half*)base)[off] = f16_cvt(v);
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
          if (KVDT == BZ_F16) u[j] = (unsigned)__half_as_ushort(f16_cvt(x0)) | ((unsigned)__half_as_ushort(f16_cvt(x1)) << 16);
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
  zero_duty<256>(a.zero_buf, a.zero_n);

  if (!a.q_only) {
    const float* cr = a.cos_t + (size_t)pos * half;
    const float* sr = a.sin_t + (size_t)pos * half;
    for (int idx = tid; idx < 2 * half; idx += 256) {
      const int hh = idx / half, i = idx % half;       // hh 0: this query head, 1: the kv head's new key
      const int base = hh == 0 ? hq * HD : a.nq * HD + kvh * HD;
      const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
      const float x0 = vsrc_get(a.qkv, base + ia, a.act), x1 = vsrc_get(a.qkv, base + ib, a.act);
      const float c = cr[i], s = sr[i];
      const float y0 = round_act(rope_lo(x0, x1, c, s), a.act), y1 = round_act(rope_hi(x0, x1, c, s), a.act);
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

  const float scale = div_rn(1.0f, sqrt_rn((float)HD));
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
      const float alpha = bz_expf(m - mn), e = bz_expf(s - mn);
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
  const float w = (m == -INFINITY) ? 0.f : bz_expf(m - M);
  const float ws = wave_sum(l * w);
  if (lane == 0) wred[4 + wave] = ws;
  const int nrows = min(len, 256);
  if (tid < nrows) {
#pragma unroll
    for (int i = 0; i < HD; i += 4) *(float4*)(ored + tid * LDR + i) = make_float4(o[i] * w, o[i + 1] * w, o[i + 2] * w, o[i + 3] * w);
  }
  __syncthreads();
  const float inv = div_rn(1.0f, (wred[4] + wred[5]) + (wred[6] + wred[7]));
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


// ---------------------------------------------------------------------------------------------------------
// attention decode v2 (head_dim 128, 16-bit cache), optionally fused with o_proj.
//   scores : lane == position; K row stays packed, q is packed f16/bf16 pairs in LDS, V_DOT2_F32_{F16,BF16}: no converts
//   PV     : the packed V rows go through an LDS image [row][272 B] (conflict-free b128 writes); thread t sums
//            d-pair (t & 63) over rows = (t >> 6) mod 4 with the softmax weights broadcast from LDS
//   chunks of 256 positions are merged online (running max / sum / output in the 128 output threads)
// FUSE: block (head, column slice) multiplies the head output by its slab of Wo (o_proj = sum over heads of Wo[:, head slab] . attn_head) -- slab loads are
// issued at kernel entry and stay in flight through the whole attention.
// ---------------------------------------------------------------------------------------------------------
template <int KVDT, int FUSE, int TPW, int NW, int PAGED, int HD = 128>   // NW waves per block: each owns 256/NW positions of a chunk; HD = 128 or 64 (64: not fused)
__global__ __launch_bounds__(NW * 64) void k_attn2(AttnArgs a, const uint4* __restrict__ W, const __half* __restrict__ S,
                                               const unsigned char* __restrict__ Z, const float* __restrict__ bias, int CS, long long* acc) {
  // Mapping: a chunk is 256 positions in row groups of RPL = 512 / HD whole rows (one wave-wide 16-byte load: 1 KiB contiguous in the contiguous cache; 4 rows of
  // 128 elements, 8 rows of 64), dealt ROUND-ROBIN to the waves: load i of wave w is row group i NW + w, lane l holds piece (l % NPC) = elements 8 piece .. +7 of
  // row RPL (i NW + w) + l / NPC (NPC = HD / 8 pieces per row).  A context of c positions is ceil(c / RPL) groups spread over all NW waves -- ceil(c / (RPL NW))
  // iterations per wave, where contiguous blocks of 256 / NW positions per wave (rounds 1-3) gave the first wave all NL iterations of a short context and ran
  // two live waves per SIMD from 129 positions on (attention + o_proj 8.6 us at 16 positions, 10.4 us averaged over a 128-token generation).  Dead iterations
  // (wave-uniform) skip their loads, scores and P.V.  A row's score is a 4-DOT2 partial per lane reduced over its 16 lanes -- and lands exactly in the
  // lanes that hold that row's V pieces, so P.V accumulates in registers (8 outputs per lane) with no LDS image.
  //
  // Prologue discipline (measured with s_memrealtime stamps per wave):
  //  * every global load up to the first barrier is unconditional per lane (clamped indices, selects instead of branches) and batched -- a load
  //    under a divergent branch costs a full s_waitcnt vmcnt(0), which used to serialise five ~1.4 us memory latencies here;
  //  * all kernel arguments are fetched in one scalar-load batch; blockDim is never read (a vector load from the dispatch packet);
  //  * cos/sin come from a row staged at a fixed address by the embed kernel, so RoPE does not wait for position -> table;
  //  * the CU's vector-memory path moves 64 B/clk, so a full 256-row K/V chunk (128 KiB per block) is ~1 us of issue at ANY context length:
  //    waves whose positions lie beyond the context skip their loads and their score / PV work.
  // FUSE: 0 attention only; 1 + int4 (AWQ / GPTQ) o_proj; 2 + dense 16-bit o_proj (weights [N][K] in the cache dtype: the head's HD k of an output row are one
  // 16-byte piece per lane, RPL output rows per wave-wide load -- the K/V row mapping again)
  static_assert(HD == 128 || (HD == 64 && FUSE != 1), "head_dim 128, or 64 without the int4 o_proj");
  constexpr int half = HD / 2, NPC = HD / 8, RPL = 64 / NPC, PW = 256 / NW, NL = PW / RPL, OW = NW > 8 ? 8 : NW, NTH = NW * 64;   // OW waves carry o_proj tiles
  asm volatile("" :: "s"(a.zero_buf), "s"(a.zero_n), "s"(a.kv.k), "s"(a.kv.v), "s"(a.kv.cap), "s"(a.kv.layer_stride), "s"(a.layer), "s"(a.act),
               "s"(a.interleaved), "s"(a.rope_cur), "s"(a.qkv.p), "s"(a.qkv.fix), "s"(a.nq), "s"(a.nkv), "s"(a.q_only), "s"(a.pos), "s"(W), "s"(S),
               "s"(Z), "s"(bias), "s"(CS), "s"(acc), "s"(a.out), "s"(a.stamps), "s"(a.kv.bs), "s"(a.kv.n_kv));   // one scalar-load batch for all arguments
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* q2 = (unsigned*)smem;             // [64] packed q pairs
  unsigned* k2 = q2 + 64;                     // [64] packed new key
  unsigned* v2 = k2 + 64;                     // [64] packed new value
  double* pout = (double*)(v2 + 64);          // [NW][128] PV partials of the waves (double: see the sums below)
  double* lred = pout + NW * 128;             // [NW] the waves' sums of the softmax weights
  float* wred = (float*)(lred + NW);          // [2 NW] the waves' maxima
  float* outh = wred + 2 * NW;                // [128] head output (FUSE)
  uint4* xpl = (uint4*)(outh + 128);          // [4 chunks][6 planes]: 384 B (as the three int8 planes before)
  int4* gpar = (int4*)(xpl + 4 * XQ_NP);      // [2]
  const int rep = a.nq / a.nkv;
  const int hq = FUSE ? blockIdx.x / CS : blockIdx.x, cs = FUSE ? blockIdx.x % CS : 0, kvh = hq / rep;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int piece = lane % NPC, rsub = lane / NPC;
#define STAMP(i) do { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
  STAMP(0);
  const KvView& kv = a.kv;
  // (0) the position word: issued first, needed late
  const int pos_v = a.pos[0];
  const int slot_v = PAGED ? *(kv.slot ? kv.slot : a.pos) : 0;
  // (0a) the q/k/v finishing's values (vmcnt is in-order: what is issued first returns first)
  const int fixq = a.q_only ? 0 : a.qkv.fix;
  int i0, i1;
  {
    const int hh = tid / half, i = tid % half;
    const int base = hh == 0 ? hq * HD : a.nq * HD + kvh * HD;
    const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
    const int vb = a.nq * HD + a.nkv * HD + kvh * HD + 2 * i;
    i0 = hh < 2 ? base + ia : vb; i1 = hh < 2 ? base + ib : vb + 1;
    if (a.q_only) { i0 = hq * HD + 2 * i; i1 = i0 + 1; }
    if (hh > 2) { i0 = 0; i1 = 0; }
  }
  unsigned r0l, r0h, r1l, r1h;
  vsrc_issue(a.qkv.p, fixq, i0, r0l, r0h);
  vsrc_issue(a.qkv.p, fixq, i1, r1l, r1h);
  const float* rc = a.q_only ? (const float*)a.qkv.p : a.rope_cur;   // staged RoPE row of this position (k_embed); unused values in q_only mode
  const float pc = rc[tid % half], ps = rc[half + tid % half];
  if (a.zero_buf) {   // zero duty (compile-time strides: blockDim / gridDim reads are loads from the dispatch packet)
    const int nblk = FUSE ? a.nq * CS : a.nq;
    for (int i = blockIdx.x * NTH + tid; i < a.zero_n; i += nblk * NTH) a.zero_buf[i] = 0;
  }
  // (1) o_proj slab of this wave (FUSE): does not depend on the position; in flight through the whole attention
  uint4 Wb[FUSE == 1 ? TPW : 1][4];
  const int t0 = (cs * OW + wave % OW) * TPW;   // waves >= OW mirror a slab (L2 hits) and skip the atomics
  constexpr int DLD = FUSE == 2 ? TPW : 1;       // FUSE 2: TPW = wave-wide loads of RPL output rows each
  uint4 Wd[DLD];
  const int nd0 = (cs * NW + wave) * DLD * RPL;  // first output row of this wave
  if (FUSE == 2) {
    const int K = a.nq * HD;
    const unsigned short* wd = (const unsigned short*)W;
#pragma unroll
    for (int t = 0; t < DLD; t++) Wd[t] = ldnt((const uint4*)(wd + (size_t)(nd0 + t * RPL + rsub) * K + hq * HD + piece * 8));
  }
  if (FUSE == 1) {
    const int K = a.nq * HD;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      const uint4* wp = W + ((size_t)(t0 + t) * (K >> 5) + hq * 4) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; c++) Wb[t][c] = ldnt(wp + c * 64);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // (1a) the first chunk's K/V rows, only for waves whose positions exist.  The CU's vector-memory path moves 64 B/clk: a full 256-row chunk
  //      is 128 KiB per block, ~1 us of issue at ANY context length (measured: the second wave of each SIMD reached the first barrier 1.3 us
  //      after the first).  Waves beyond the context skip the loads here and the score / PV work below.
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);
  const int pmax = pos > 0 ? pos - 1 : 0;
  const int len = a.q_only ? pos : pos + 1;
  const unsigned short* kb0 = (const unsigned short*)kv.k + (size_t)a.layer * kv.layer_stride;
  const unsigned short* vb0 = (const unsigned short*)kv.v + (size_t)a.layer * kv.layer_stride;
  uint4 kr[NL], vr[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) { kr[i] = make_uint4(0, 0, 0, 0); vr[i] = make_uint4(0, 0, 0, 0); }
#define ATT_ROW(i) (((i) * NW + wave) * RPL + rsub)                 /* row of load i inside a chunk */
#define ATT_LIVE(c0_, i) ((c0_) + ((i) * NW + wave) * RPL < len)    /* wave-uniform: the row group has a position of the context */
  if (wave * RPL < len) {
    if (!PAGED) {
      const unsigned short* kb = kb0 + (size_t)kvh * kv.cap * HD;
      const unsigned short* vb = vb0 + (size_t)kvh * kv.cap * HD;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (ATT_LIVE(0, i)) {
          const unsigned off = (unsigned)min(ATT_ROW(i), pmax) * HD + piece * 8;
          kr[i] = *(const uint4*)(kb + off);
          vr[i] = *(const uint4*)(vb + off);
        }
      }
    } else {
      int blk[NL];
#pragma unroll
      for (int i = 0; i < NL; i++) blk[i] = ATT_LIVE(0, i) ? kv.block_table[min(ATT_ROW(i), pmax) / kv.bs] : 0;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (ATT_LIVE(0, i)) {
          const int pp = min(ATT_ROW(i), pmax);
          const size_t off = (((size_t)blk[i] * kv.n_kv + kvh) * kv.bs + (pp % kv.bs)) * HD + piece * 8;
          kr[i] = *(const uint4*)(kb0 + off);
          vr[i] = *(const uint4*)(vb0 + off);
        }
      }
    }
  }
  STAMP(1);
  // (2) q/k/v finishing: fixed point -> f32, rounding, RoPE; packed q / new key / new value to LDS; KV append
  const float px0 = vsrc_finish(fixq, r0l, r0h, a.act), px1 = vsrc_finish(fixq, r1l, r1h, a.act);
  if (!a.q_only) {
    if (tid < 2 * half) {
      const int hh = tid / half, i = tid % half;
      const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
      const float y0 = round_act(rope_lo(px0, px1, pc, ps), a.act), y1 = round_act(rope_hi(px0, px1, pc, ps), a.act);
      unsigned short* dst = (unsigned short*)(hh == 0 ? q2 : k2);
      const unsigned p0 = pack2<KVDT>(y0, y1);
      dst[ia] = (unsigned short)(p0 & 0xffffu);
      dst[ib] = (unsigned short)(p0 >> 16);
    } else if (tid < 3 * half) {
      v2[tid - 2 * half] = pack2<KVDT>(px0, px1);
    }
    __syncthreads();
    if (hq % rep == 0 && cs == 0 && tid < half) {   // KV append, once per kv head: HD / 2 threads x 4-byte pairs
      size_t woff;
      if (PAGED) woff = kv_slot_off(kv, a.layer, kvh, kv.slot ? slot_v : (kv.block_table[pos / kv.bs] * kv.bs + pos % kv.bs));
      else woff = kv_row_off_t<0>(kv, a.layer, kvh, pos);
      ((unsigned*)((unsigned short*)kv.k + woff))[tid] = k2[tid];
      ((unsigned*)((unsigned short*)kv.v + woff))[tid] = v2[tid];
    }
  } else {
    if (tid < half) q2[tid] = pack2<KVDT>(px0, px1);
    __syncthreads();
  }
  STAMP(2);

  // Exact sums, two passes (the oracle's definition, oracle/orc_ops.c: orc_attn_decode): score = f32(sum q k) * scale with the sum carried in double over exact
  // products; the maximum over ALL positions first, then p = exp(score - max) (bz_expf), L = f32(sum p), O = f32(sum p v), both sums in double -- order-independent
  // to ~1e-16, so the result does not depend on the chunk / wave / lane decomposition and equals the oracle's bits (round 2 merged 256-position chunks online
  // with f32 sums: ~1e-7 of order-dependent error, i.e. about one flipped f16 rounding per layer).  A context of up to 256 positions keeps its rows and scores in
  // registers between the passes; longer ones re-read K (L2 hits) in pass 2.
  const float scale = div_rn(1.0f, sqrt_rn((float)HD));
  const uint4 qq = ((const uint4*)q2)[piece];
  float qf[8];
  unpack2<KVDT>(qq.x, qf[0], qf[1]); unpack2<KVDT>(qq.y, qf[2], qf[3]); unpack2<KVDT>(qq.z, qf[4], qf[5]); unpack2<KVDT>(qq.w, qf[6], qf[7]);
  const bool single = len <= 256;
  float sc_[NL];
  float Mw = -INFINITY;
#define ATT_SCORES(c0_)                                                                                                    \
  _Pragma("unroll") for (int i = 0; i < NL; i++) {                                                                         \
    if (!ATT_LIVE(c0_, i)) { sc_[i] = -INFINITY; continue; }                                                               \
    const int p = (c0_) + ATT_ROW(i);                                                                                      \
    uint4 kk = kr[i];                                                                                                      \
    if (!a.q_only && p == pos) { kk = ((const uint4*)k2)[piece]; vr[i] = ((const uint4*)v2)[piece]; }                      \
    float kf[8];                                                                                                           \
    unpack2<KVDT>(kk.x, kf[0], kf[1]); unpack2<KVDT>(kk.y, kf[2], kf[3]); unpack2<KVDT>(kk.z, kf[4], kf[5]); unpack2<KVDT>(kk.w, kf[6], kf[7]); \
    double d = 0.0;                                                                                                        \
    _Pragma("unroll") for (int e = 0; e < 8; e++) d = fma((double)kf[e], (double)qf[e], d);                                \
    d = grp_sum_d<NPC>(d);                                                                                                 \
    sc_[i] = (p < len) ? (float)d * scale : -INFINITY;                                                                     \
  }
  for (int c0 = 0; c0 < len; c0 += 256) {
    const bool won = c0 + wave * RPL < len;   // wave-uniform: this wave has live positions in the chunk
    if (!won) continue;
    if (c0 > 0) {
#pragma unroll
      for (int i = 0; i < NL; i++)
        if (ATT_LIVE(c0, i)) kr[i] = *(const uint4*)(kb0 + kv_row_off_t<PAGED>(kv, 0, kvh, min(c0 + ATT_ROW(i), pmax)) + piece * 8);
    }
    ATT_SCORES(c0)
#pragma unroll
    for (int i = 0; i < NL; i++) Mw = fmaxf(Mw, sc_[i]);
  }
  Mw = wave_max(Mw);
  STAMP(3);
  if (lane == 0) wred[wave] = Mw;
  __syncthreads();
  float Mall = wred[0];
#pragma unroll
  for (int w = 1; w < NW; w++) Mall = fmaxf(Mall, wred[w]);
  double accv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double lsum = 0.0;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const bool won = c0 + wave * RPL < len;
    if (!won) continue;
    if (!single) {
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (ATT_LIVE(c0, i)) {
          const size_t off = kv_row_off_t<PAGED>(kv, 0, kvh, min(c0 + ATT_ROW(i), pmax)) + piece * 8;
          kr[i] = *(const uint4*)(kb0 + off);
          vr[i] = *(const uint4*)(vb0 + off);
        }
      }
      ATT_SCORES(c0)
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
      if (!ATT_LIVE(c0, i)) continue;
      const float e = (sc_[i] == -INFINITY) ? 0.f : bz_expf(sc_[i] - Mall);
      const double ed = (double)e;
      lsum += ed;
      float v[8];
      unpack2<KVDT>(vr[i].x, v[0], v[1]); unpack2<KVDT>(vr[i].y, v[2], v[3]);
      unpack2<KVDT>(vr[i].z, v[4], v[5]); unpack2<KVDT>(vr[i].w, v[6], v[7]);
#pragma unroll
      for (int q = 0; q < 8; q++) accv[q] = fma(ed, (double)v[q], accv[q]);
    }
  }
#undef ATT_SCORES
#undef ATT_LIVE
#undef ATT_ROW
  // rows of the wave: the RPL lane groups hold different rows -> reduce over xor (8,) 16, 32 (the lanes of a group are replicas for lsum)
  if (NPC == 8) lsum += dpp_get<DPP_ROR8>(lsum);
  lsum = xrow32_d(xrow16_d(lsum));
#pragma unroll
  for (int q = 0; q < 8; q++) {
    if (NPC == 8) accv[q] += dpp_get<DPP_ROR8>(accv[q]);
    accv[q] = xrow32_d(xrow16_d(accv[q]));
  }
  STAMP(4);
  if (lane < NPC) {
#pragma unroll
    for (int q = 0; q < 8; q++) pout[wave * HD + piece * 8 + q] = accv[q];
  }
  if (lane == 0) lred[wave] = lsum;
  __syncthreads();
  STAMP(5);
  float Orun = 0.f, Lrun = 1.f;
  if (tid < HD) {
    double oc = 0.0, Lc = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) { oc += pout[w * HD + tid]; Lc += lred[w]; }
    Orun = (float)oc; Lrun = (float)Lc;
  }
  if (!FUSE) {
    if (tid < HD) a.out[(size_t)hq * HD + tid] = round_act(div_rn(Orun, Lrun), a.act);
    return;
  }
  if (tid < HD) outh[tid] = round_act(div_rn(Orun, Lrun), a.act);
  if (FUSE == 2) {
    // dense o_proj: this lane's 8 weights of each output row against its 8 head outputs, reduced over the row's NPC lanes, one fixed-point atomic per (head, row)
    __syncthreads();
    const float4 oa = *(const float4*)(outh + piece * 8), ob = *(const float4*)(outh + piece * 8 + 4);
    const float of8[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
#pragma unroll
    for (int t = 0; t < DLD; t++) {
      float w8[8];
      unpack2<KVDT>(Wd[t].x, w8[0], w8[1]); unpack2<KVDT>(Wd[t].y, w8[2], w8[3]); unpack2<KVDT>(Wd[t].z, w8[4], w8[5]); unpack2<KVDT>(Wd[t].w, w8[6], w8[7]);
      double d = 0.0;                 // exact sums, as the other dense GEMVs (piece_dot_x: 16-bit x 16-bit products are exact in f32)
#pragma unroll
      for (int e = 0; e < 8; e++) d += (double)__fmul_rn(w8[e], of8[e]);
      d = grp_sum_d<NPC>(d);
      const int n = nd0 + t * RPL + rsub;
      if (piece == 0) {
        if (bias != nullptr && hq == 0) d += (double)bias[n];
        atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(d, a.act));
      }
    }
    return;
  }
  // slab scales / zero points (L2-resident, tiny): fetched now, used after the quantisation
  float sc[FUSE == 1 ? TPW : 1]; int zp[FUSE == 1 ? TPW : 1];
  {
    const int G = (a.nq * HD) >> 7;
#pragma unroll
    for (int t = 0; t < (FUSE == 1 ? TPW : 1); t++) {
      sc[t] = __half2float(S[((size_t)(t0 + t) * G + hq) * 64 + lane]);
      zp[t] = Z[((size_t)(t0 + t) * G + hq) * 64 + lane];
    }
  }
  __syncthreads();
  STAMP(6);
  quant_x128<NW * 64>(outh, HD, xpl, gpar);
  __syncthreads();
  STAMP(7);
#pragma unroll
  for (int t = 0; t < (FUSE == 1 ? TPW : 1); t++) {
    double y = 0.0;
    q4g_consume(Wb[t], 0, xpl, gpar, sc[t], zp[t], y);
    const int n = (t0 + t) * 64 + lane;
    if (bias != nullptr && hq == 0) y += (double)bias[n];
    if (wave < OW) atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(y, a.act));
  }
  STAMP(8);
#undef STAMP
}


// ---------------------------------------------------------------------------------------------------------
// attention decode, head_dim 128, F32 cache (GGUF models keep f32 activations and an f32 cache): k_attn2's structure and prologue
// discipline -- 8 waves, lane = (row l >> 4, 8-element piece l & 15), in-register PV, online merge of 256-position chunks, dead waves
// skipped -- with 32-byte row pieces and plain FMAs.  Replaces the one-thread-per-position kernel (14 us per layer on the Mistral shape).
// grid = n_heads, 512 threads.  FUSE: + a Q4_K o_proj, as k_attn2 does for the int4 one: grid = n_heads x CS column slices, wave w of slice cs owns output
// tile cs * 8 + w and the head's 128 k of it -- four 32-k chunks of superblock hq / 2 (a 256-k superblock spans two heads) with that superblock's
// header --, requested at entry; the head output is quantised per 32-k chunk (quant8_x32) and the tile's partial goes to the fixed-point ring.
// ---------------------------------------------------------------------------------------------------------
template <int PAGED, int FUSE = 0>
__global__ __launch_bounds__(512) void k_attn2f(AttnArgs a, const uint4* __restrict__ Wq, const uint4* __restrict__ Hd, const float* __restrict__ bias, int CS, long long* acc) {
  constexpr int HD = 128, half = 64, NW = 8, PW = 256 / NW, NL = PW / 4, NTH = NW * 64;
  __shared__ __attribute__((aligned(16))) float qf[HD], kf[HD], vf[HD];
  __shared__ float wred[2 * NW];
  __shared__ __attribute__((aligned(16))) float pout[NW * 128];
  __shared__ __attribute__((aligned(16))) float outh[FUSE ? HD : 4];
  __shared__ __attribute__((aligned(16))) unsigned xh[FUSE ? 32 : 4], xm[FUSE ? 32 : 4], xl[FUSE ? 32 : 4];
  __shared__ int4 cpar[FUSE ? 8 : 1];
  asm volatile("" :: "s"(a.zero_buf), "s"(a.zero_n), "s"(a.kv.k), "s"(a.kv.v), "s"(a.kv.cap), "s"(a.kv.layer_stride), "s"(a.layer), "s"(a.act),
               "s"(a.interleaved), "s"(a.rope_cur), "s"(a.qkv.p), "s"(a.qkv.fix), "s"(a.nq), "s"(a.nkv), "s"(a.q_only), "s"(a.pos), "s"(a.out),
               "s"(a.kv.bs), "s"(a.kv.n_kv), "s"(Wq), "s"(Hd), "s"(bias), "s"(CS), "s"(acc));
  const int rep = a.nq / a.nkv;
  const int hq = FUSE ? blockIdx.x / CS : blockIdx.x, cs = FUSE ? blockIdx.x % CS : 0, kvh = hq / rep;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int piece = lane & 15, rsub = lane >> 4;
  const KvView& kv = a.kv;
  const int pos_v = a.pos[0];
  const int slot_v = PAGED ? *(kv.slot ? kv.slot : a.pos) : 0;
  const int fixq = a.q_only ? 0 : a.qkv.fix;
  int i0, i1;
  {
    const int hh = tid / half, i = tid % half;
    const int base = hh == 0 ? hq * HD : a.nq * HD + kvh * HD;
    const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
    const int vb = a.nq * HD + a.nkv * HD + kvh * HD + 2 * i;
    i0 = hh < 2 ? base + ia : vb; i1 = hh < 2 ? base + ib : vb + 1;
    if (a.q_only) { i0 = hq * HD + 2 * i; i1 = i0 + 1; }
    if (hh > 2) { i0 = 0; i1 = 0; }
  }
  unsigned r0l, r0h, r1l, r1h;
  vsrc_issue(a.qkv.p, fixq, i0, r0l, r0h);
  vsrc_issue(a.qkv.p, fixq, i1, r1l, r1h);
  const float* rc = a.q_only ? (const float*)a.qkv.p : a.rope_cur;
  const float pc = rc[tid % half], ps = rc[half + tid % half];
  if (a.zero_buf)
    for (int i = blockIdx.x * NTH + tid; i < a.zero_n; i += (FUSE ? a.nq * CS : a.nq) * NTH) a.zero_buf[i] = 0;
  // o_proj slab of this wave (FUSE): does not depend on the position; in flight through the whole attention
  uint4 Wb[4]; uint4 hdw = make_uint4(0, 0, 0, 0);
  const int t0 = cs * NW + wave;
  if (FUSE) {
    const int K = a.nq * HD;
    const uint4* wp = Wq + ((size_t)t0 * (K >> 5) + hq * 4) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; c++) Wb[c] = ldnt(wp + c * 64);
    hdw = Hd[((size_t)t0 * (K >> 8) + (hq >> 1)) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);
  const int pmax = pos > 0 ? pos - 1 : 0;
  const int len = a.q_only ? pos : pos + 1;
  const float* kb0 = (const float*)kv.k + (size_t)a.layer * kv.layer_stride;
  const float* vb0 = (const float*)kv.v + (size_t)a.layer * kv.layer_stride;
  float4 kr[NL][2], vr[NL][2];
#pragma unroll
  for (int i = 0; i < NL; i++) { kr[i][0] = kr[i][1] = vr[i][0] = vr[i][1] = make_float4(0.f, 0.f, 0.f, 0.f); }
  // row groups of 4 positions dealt round-robin to the waves (k_attn2's mapping: a short context spreads over all 8 waves, dead iterations are skipped)
#define ATF_ROW(i) (((i) * NW + wave) * 4 + rsub)
#define ATF_LIVE(c0_, i) ((c0_) + ((i) * NW + wave) * 4 < len)
  if (wave * 4 < len) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
      if (ATF_LIVE(0, i)) {
        const size_t off = kv_row_off_t<PAGED>(kv, 0, kvh, min(ATF_ROW(i), pmax)) + piece * 8;
        kr[i][0] = *(const float4*)(kb0 + off); kr[i][1] = *(const float4*)(kb0 + off + 4);
        vr[i][0] = *(const float4*)(vb0 + off); vr[i][1] = *(const float4*)(vb0 + off + 4);
      }
    }
  }
  const float px0 = vsrc_finish(fixq, r0l, r0h, a.act), px1 = vsrc_finish(fixq, r1l, r1h, a.act);
  if (!a.q_only) {
    if (tid < 2 * half) {
      const int hh = tid / half, i = tid % half;
      const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
      float* dst = hh == 0 ? qf : kf;
      dst[ia] = round_act(rope_lo(px0, px1, pc, ps), a.act);
      dst[ib] = round_act(rope_hi(px0, px1, pc, ps), a.act);
    } else if (tid < 2 * half + 64) {
      vf[2 * (tid - 2 * half)] = px0; vf[2 * (tid - 2 * half) + 1] = px1;
    }
    __syncthreads();
    if (hq % rep == 0 && cs == 0 && tid < HD) {     // KV append, once per kv head
      size_t woff;
      if (PAGED) woff = kv_slot_off(kv, a.layer, kvh, kv.slot ? slot_v : (kv.block_table[pos / kv.bs] * kv.bs + pos % kv.bs));
      else woff = kv_row_off_t<0>(kv, a.layer, kvh, pos);
      ((float*)kv.k)[woff + tid] = kf[tid];
      ((float*)kv.v)[woff + tid] = vf[tid];
    }
  } else {
    if (tid < 64) { qf[2 * tid] = px0; qf[2 * tid + 1] = px1; }
    __syncthreads();
  }
  const float scale = div_rn(1.0f, sqrt_rn((float)HD));
  const float4 qa = *(const float4*)(qf + piece * 8), qb = *(const float4*)(qf + piece * 8 + 4);
  float Mrun = -INFINITY, Lrun = 0.f, Orun = 0.f;   // running state (threads < 128 own output d = tid)
  for (int c0 = 0; c0 < len; c0 += 256) {
    const bool won = c0 + wave * 4 < len;
    if (c0 > 0) {
      __syncthreads();
      if (won) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
          if (ATF_LIVE(c0, i)) {
            const size_t off = kv_row_off_t<PAGED>(kv, 0, kvh, min(c0 + ATF_ROW(i), pmax)) + piece * 8;
            kr[i][0] = *(const float4*)(kb0 + off); kr[i][1] = *(const float4*)(kb0 + off + 4);
            vr[i][0] = *(const float4*)(vb0 + off); vr[i][1] = *(const float4*)(vb0 + off + 4);
          }
        }
      }
    }
    float sc_[NL];
    float mloc = -INFINITY;
    if (won) {
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (!ATF_LIVE(c0, i)) { sc_[i] = -INFINITY; continue; }
        const int p = c0 + ATF_ROW(i);
        float4 k0 = kr[i][0], k1 = kr[i][1];
        if (!a.q_only && p == pos) {
          k0 = *(const float4*)(kf + piece * 8); k1 = *(const float4*)(kf + piece * 8 + 4);
          vr[i][0] = *(const float4*)(vf + piece * 8); vr[i][1] = *(const float4*)(vf + piece * 8 + 4);
        }
        float d = k0.x * qa.x + k0.y * qa.y + k0.z * qa.z + k0.w * qa.w + k1.x * qb.x + k1.y * qb.y + k1.z * qb.z + k1.w * qb.w;
        d = grp_reduce<16, OpAdd>(d);
        sc_[i] = (p < len) ? d * scale : -INFINITY;
        mloc = fmaxf(mloc, sc_[i]);
      }
      mloc = wave_max(mloc);
    }
    if (lane == 0) wred[wave] = mloc;
    __syncthreads();
    float Mc = wred[0];
#pragma unroll
    for (int w = 1; w < NW; w++) Mc = fmaxf(Mc, wred[w]);
    float accv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    if (won) {
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (!ATF_LIVE(c0, i)) continue;
        const float e = (sc_[i] == -INFINITY) ? 0.f : bz_expf(sc_[i] - Mc);
        lsum += e;
        accv[0] = fmaf(e, vr[i][0].x, accv[0]); accv[1] = fmaf(e, vr[i][0].y, accv[1]); accv[2] = fmaf(e, vr[i][0].z, accv[2]); accv[3] = fmaf(e, vr[i][0].w, accv[3]);
        accv[4] = fmaf(e, vr[i][1].x, accv[4]); accv[5] = fmaf(e, vr[i][1].y, accv[5]); accv[6] = fmaf(e, vr[i][1].z, accv[6]); accv[7] = fmaf(e, vr[i][1].w, accv[7]);
      }
      lsum = xrow32<OpAdd>(xrow16<OpAdd>(lsum));
#pragma unroll
      for (int q = 0; q < 8; q++) accv[q] = xrow32<OpAdd>(xrow16<OpAdd>(accv[q]));
    }
    if (lane < 16) {
      *(float4*)(pout + wave * 128 + piece * 8) = make_float4(accv[0], accv[1], accv[2], accv[3]);
      *(float4*)(pout + wave * 128 + piece * 8 + 4) = make_float4(accv[4], accv[5], accv[6], accv[7]);
    }
    if (lane == 0) wred[NW + wave] = lsum;
    __syncthreads();
    if (tid < 128) {
      float oc = 0.f, Lc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w += 2) { oc += pout[w * 128 + tid] + pout[(w + 1) * 128 + tid]; Lc += wred[NW + w] + wred[NW + w + 1]; }
      const float Mn = fmaxf(Mrun, Mc);
      const float fa = (Mrun == -INFINITY) ? 0.f : bz_expf(Mrun - Mn), fb = bz_expf(Mc - Mn);
      Orun = Orun * fa + oc * fb;
      Lrun = Lrun * fa + Lc * fb;
      Mrun = Mn;
    }
  }
  if (!FUSE) {
    if (tid < 128) a.out[(size_t)hq * HD + tid] = round_act(div_rn(Orun, Lrun), a.act);
    return;
  }
  if (tid < 128) outh[tid] = round_act(div_rn(Orun, Lrun), a.act);
  __syncthreads();
  quant_x32<GQ_Q4K>(outh, HD, xh, xm, xl, cpar);
  __syncthreads();
  {
    const unsigned hw[4] = {hdw.x, hdw.y, hdw.z, hdw.w};
    const float d = __half2float(__ushort_as_half((unsigned short)(hw[0] & 0xffffu)));
    const float dmin = __half2float(__ushort_as_half((unsigned short)(hw[0] >> 16)));
    float y = 0.f;
    if (hq & 1) {     // (compile-time chunk indices for the sub-scale extraction: two copies)
#pragma unroll
      for (int c = 0; c < 4; c++) y += gq_chunk_q4k(Wb[c], c, 4 + c, hw, d, dmin, (const uint4*)xh, (const uint4*)xm, (const uint4*)xl, cpar);
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) y += gq_chunk_q4k(Wb[c], c, c, hw, d, dmin, (const uint4*)xh, (const uint4*)xm, (const uint4*)xl, cpar);
    }
    const int n = t0 * 64 + lane;
    if (bias != nullptr && hq == 0) y += bias[n];
    atomicAdd((unsigned long long*)(acc + n), (unsigned long long)f2fix(y, a.act));
  }
}

static size_t attn2_smem(int nw) { return (size_t)(64 * 3 + 2 * nw + 128 + 4 * XQ_NP * 4 + 8) * 4 + (size_t)(nw * 128 + nw) * 8; }

// picks the column-slice count so that nq * CS ~ 256 workgroups, and the wave count (8 waves halve the per-wave attention chain);
// returns 0 when the fused form does not apply
static int attn_oproj_plan(const AttnArgs& a, const LinearDev& L, int& NW) {
  NW = 0;
  if (a.hd != 128 || L.kind != LK_Q4G || L.perm != nullptr || L.K != a.nq * 128 || a.q_only) return 0;
  const int NT = L.N / 64;
  for (int nw = 8; nw >= 4; nw >>= 1) {
    const int ow = nw > 8 ? 8 : nw;
    for (int cs = 8; cs >= 1; cs >>= 1)
      if (NT % (cs * ow) == 0 && NT / (cs * ow) <= 2 && a.nq * cs <= 1024) { NW = nw; return cs; }
  }
  return 0;
}
// the f32-cache form (GGUF models): Q4_K o_proj, one tile per wave, CS = N / 512 column slices; single-launch contexts only (no split-KV path for the f32 cache)
static int attn2f_oproj_slices(const AttnArgs& a, const LinearDev& L) {
  static const bool off = getenv("BZ_NO_ATTN_F32") != nullptr;
  if (off || a.hd != 128 || a.kv.dtype != BZ_F32 || a.rope_cur == nullptr || a.q_only || L.kind != LK_Q4K || L.K != a.nq * 128 || L.N % 512 || L.K % 256) return 0;
  const int cs = L.N / 512;
  return a.nq * cs <= 2048 ? cs : 0;
}
// dense 16-bit o_proj in the cache dtype (Llama-3.2-1B: head_dim 64, N 2048): 8 column slices, RPL rows per wave-wide load, 1..4 loads per wave
static int attn_dense_oproj_loads(const AttnArgs& a, const LinearDev& L) {
  static const bool off = getenv("BZ_NO_ATTN2_HD64") != nullptr;
  if (off || L.kind != LK_ROWS || L.wdt != a.kv.dtype || (a.kv.dtype != BZ_F16 && a.kv.dtype != BZ_BF16) || (a.hd != 64 && a.hd != 128) || a.q_only || a.rope_cur == nullptr) return 0;
  if (L.K != a.nq * a.hd || a.nq * 8 > 2048) return 0;
  const int rpl = 512 / a.hd, per_wave = L.N / 64;          // 8 slices x 8 waves
  if (L.N % 64 || per_wave % rpl) return 0;
  const int loads = per_wave / rpl;
  return (loads == 1 || loads == 2 || loads == 4) ? loads : 0;
}
int bzk_attn_oproj_slices(const AttnArgs& a, const LinearDev& L) {
  if (a.kv.dtype == BZ_F32) return attn2f_oproj_slices(a, L);
  if (L.kind == LK_ROWS) return attn_dense_oproj_loads(a, L) > 0 ? 8 : 0;
  int nw; return attn_oproj_plan(a, L, nw);
}

int bzk_attn_oproj(hipStream_t s, const AttnArgs& a, const LinearDev& L, long long* acc) {
  if (a.kv.dtype == BZ_F32) {
    const int CS = attn2f_oproj_slices(a, L);
    if (CS <= 0) BZ_FAIL(BZ_E_INVALID, "attn+o_proj fusion (f32 cache) does not apply to this shape");
    if (a.kv.paged) BZ_LAUNCH("attn+o_proj<q4_K>", L.algo_bytes, (k_attn2f<1, 1>), dim3(a.nq * CS), dim3(512), 0, s, a, (const uint4*)L.w, (const uint4*)L.hdr, L.bias, CS, acc);
    else BZ_LAUNCH("attn+o_proj<q4_K>", L.algo_bytes, (k_attn2f<0, 1>), dim3(a.nq * CS), dim3(512), 0, s, a, (const uint4*)L.w, (const uint4*)L.hdr, L.bias, CS, acc);
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  if (L.kind == LK_ROWS) {
    const int DL = attn_dense_oproj_loads(a, L);
    if (DL <= 0) BZ_FAIL(BZ_E_INVALID, "attn+o_proj fusion (dense) does not apply to this shape");
#define LAUNCH_AD(DT, T, PG, HDV) BZ_LAUNCH("attn+o_proj<dense>", L.algo_bytes, (k_attn2<DT, 2, T, 8, PG, HDV>), dim3(a.nq * 8), dim3(512), attn2_smem(8), s, a, \
    (const uint4*)L.w, (const __half*)nullptr, (const unsigned char*)nullptr, L.bias, 8, acc)
#define LAUNCH_AD_P(DT, T, HDV) do { if (a.kv.paged) LAUNCH_AD(DT, T, 1, HDV); else LAUNCH_AD(DT, T, 0, HDV); } while (0)
#define LAUNCH_AD_T(DT, HDV) do { if (DL == 1) LAUNCH_AD_P(DT, 1, HDV); else if (DL == 2) LAUNCH_AD_P(DT, 2, HDV); else LAUNCH_AD_P(DT, 4, HDV); } while (0)
#define LAUNCH_AD_H(DT) do { if (a.hd == 64) LAUNCH_AD_T(DT, 64); else LAUNCH_AD_T(DT, 128); } while (0)
    if (a.kv.dtype == BZ_F16) LAUNCH_AD_H(BZ_F16); else LAUNCH_AD_H(BZ_BF16);
#undef LAUNCH_AD_H
#undef LAUNCH_AD_T
#undef LAUNCH_AD_P
#undef LAUNCH_AD
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  int NW;
  const int CS = attn_oproj_plan(a, L, NW);
  if (CS <= 0) BZ_FAIL(BZ_E_INVALID, "attn+o_proj fusion does not apply to this shape");
  const int TPW = (L.N / 64) / (CS * (NW > 8 ? 8 : NW));
#define LAUNCH_AO(DT, T, W_, PG) BZ_LAUNCH("attn+o_proj", L.algo_bytes, (k_attn2<DT, 1, T, W_, PG>), dim3(a.nq * CS), dim3(W_ * 64), attn2_smem(W_), s, a, \
    (const uint4*)L.w, (const __half*)L.scales, (const unsigned char*)L.zeros, L.bias, CS, acc)
#define LAUNCH_AO_W(DT, T, PG) do { if (NW == 8) LAUNCH_AO(DT, T, 8, PG); else LAUNCH_AO(DT, T, 4, PG); } while (0)
#define LAUNCH_AO_P(DT, T) do { if (a.kv.paged) LAUNCH_AO_W(DT, T, 1); else LAUNCH_AO_W(DT, T, 0); } while (0)
#define LAUNCH_AO_T(DT) do { if (TPW == 1) LAUNCH_AO_P(DT, 1); else LAUNCH_AO_P(DT, 2); } while (0)
  if (a.kv.dtype == BZ_F16) LAUNCH_AO_T(BZ_F16);
  else if (a.kv.dtype == BZ_BF16) LAUNCH_AO_T(BZ_BF16);
  else BZ_FAIL(BZ_E_UNSUPPORTED, "attn+o_proj fusion: f32 KV cache not built");
#undef LAUNCH_AO_P
#undef LAUNCH_AO_W
#undef LAUNCH_AO_T
#undef LAUNCH_AO
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Long-context decode: split-KV attention (flash-decoding) + merge.
// The fused kernel above gives every (head, column-slice) block the WHOLE context (8x redundant K/V reads, one 256-row chunk at a time):
// fine up to a few hundred positions, 69 us per layer at 4096.  Beyond BZ_SPLIT_MIN positions the step runs two launches instead:
//   k_attn_split       : block (kv head, split s) owns positions [s*SPL, (s+1)*SPL) for all REP query heads of the group (K/V rows are
//                        loaded once per group); it also finishes q (RoPE), and the block that owns the new position appends K/V.
//                        Writes one partial per (head, split): m, l, o[128] (unnormalised, relative to m).
//   k_attn_merge[_oproj]: merges the live partials of a head (weights exp(m_s - M)), then -- fused form -- multiplies by the o_proj slab exactly
//                        like k_attn2's tail.
// The grid of k_attn_split is sized for the cache CAPACITY so that a captured graph is valid at every position; blocks beyond the
// context return at once.
// ---------------------------------------------------------------------------------------------------------
#define ATT_PSTRIDE 132   // floats per partial: [0] m, [1] l, [4..131] o

template <int KVDT, int PAGED, int REP>
__global__ __launch_bounds__(REP == 8 ? 512 : 256) void k_attn_split(AttnArgs a, int SPL, int nsplit, float* __restrict__ ws) {
  constexpr int HD = 128, half = 64, NW = REP == 8 ? 8 : 4, NTH = NW * 64, PW = 128 / NW, NL = PW / 4;
  __shared__ __attribute__((aligned(16))) unsigned q2[REP][64];
  __shared__ __attribute__((aligned(16))) unsigned k2[64], v2[64];
  __shared__ float wred[REP][NW], lred[REP][NW];
  __shared__ __attribute__((aligned(16))) float pout[NW][REP][128];
  asm volatile("" :: "s"(a.kv.k), "s"(a.kv.v), "s"(a.kv.cap), "s"(a.kv.layer_stride), "s"(a.layer), "s"(a.act), "s"(a.interleaved), "s"(a.rope_cur),
               "s"(a.qkv.p), "s"(a.qkv.fix), "s"(a.nq), "s"(a.nkv), "s"(a.pos), "s"(SPL), "s"(nsplit), "s"(ws), "s"(a.kv.bs), "s"(a.kv.n_kv),
               "s"(a.zero_buf), "s"(a.zero_n));
  const int kvh = blockIdx.x / nsplit, s = blockIdx.x % nsplit;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int piece = lane & 15, rsub = lane >> 4;
  const KvView& kv = a.kv;
  const int pos_v = a.pos[0];
  const int slot_v = PAGED ? *(kv.slot ? kv.slot : a.pos) : 0;
  // q pairs: thread t -> head kvh*REP + t/64, pair t%64 (threads beyond REP*64 repeat the last head); new k / v pairs: threads 0..63 / 64..127
  const int hl = min(tid / half, REP - 1), i = tid % half;
  const int ia = a.interleaved ? 2 * i : i, ib = a.interleaved ? 2 * i + 1 : i + half;
  const int qb = (kvh * REP + hl) * HD;
  unsigned q0l, q0h, q1l, q1h, n0l, n0h, n1l, n1h;
  vsrc_issue(a.qkv.p, a.qkv.fix, qb + ia, q0l, q0h);
  vsrc_issue(a.qkv.p, a.qkv.fix, qb + ib, q1l, q1h);
  {
    const int kb_ = a.nq * HD + kvh * HD, vb_ = a.nq * HD + a.nkv * HD + kvh * HD;
    const bool isk = tid < half;
    vsrc_issue(a.qkv.p, a.qkv.fix, isk ? kb_ + ia : vb_ + 2 * i, n0l, n0h);
    vsrc_issue(a.qkv.p, a.qkv.fix, isk ? kb_ + ib : vb_ + 2 * i + 1, n1l, n1h);
  }
  const float pc = a.rope_cur[i], ps = a.rope_cur[half + i];
  if (a.zero_buf)
    for (int z = blockIdx.x * NTH + tid; z < a.zero_n; z += gridDim.x * NTH) a.zero_buf[z] = 0;
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);
  const int len = pos + 1, pmax = pos > 0 ? pos - 1 : 0;
  const int p0 = s * SPL;
  if (p0 >= len) return;                    // split beyond the context (the grid covers the capacity)
  const int p1 = min(p0 + SPL, len);
  const bool owner = p1 == len;             // this block's range holds the new position
  {
    const float x0 = vsrc_finish(a.qkv.fix, q0l, q0h, a.act), x1 = vsrc_finish(a.qkv.fix, q1l, q1h, a.act);
    const unsigned pk = pack2<KVDT>(round_act(rope_lo(x0, x1, pc, ps), a.act), round_act(rope_hi(x0, x1, pc, ps), a.act));
    ((unsigned short*)q2[hl])[ia] = (unsigned short)(pk & 0xffffu);
    ((unsigned short*)q2[hl])[ib] = (unsigned short)(pk >> 16);
  }
  if (owner && tid < 2 * half) {
    const float x0 = vsrc_finish(a.qkv.fix, n0l, n0h, a.act), x1 = vsrc_finish(a.qkv.fix, n1l, n1h, a.act);
    if (tid < half) {
      const unsigned pk = pack2<KVDT>(round_act(rope_lo(x0, x1, pc, ps), a.act), round_act(rope_hi(x0, x1, pc, ps), a.act));
      ((unsigned short*)k2)[ia] = (unsigned short)(pk & 0xffffu);
      ((unsigned short*)k2)[ib] = (unsigned short)(pk >> 16);
    } else {
      v2[i] = pack2<KVDT>(x0, x1);
    }
  }
  __syncthreads();
  if (owner && tid < 64) {                  // KV append
    size_t woff;
    if (PAGED) woff = kv_slot_off(kv, a.layer, kvh, kv.slot ? slot_v : (kv.block_table[pos / kv.bs] * kv.bs + pos % kv.bs));
    else woff = kv_row_off_t<0>(kv, a.layer, kvh, pos);
    ((unsigned*)((unsigned short*)kv.k + woff))[tid] = k2[tid];
    ((unsigned*)((unsigned short*)kv.v + woff))[tid] = v2[tid];
  }
  const unsigned short* kb0 = (const unsigned short*)kv.k + (size_t)a.layer * kv.layer_stride;
  const unsigned short* vb0 = (const unsigned short*)kv.v + (size_t)a.layer * kv.layer_stride;
  const float scale = div_rn(1.0f, sqrt_rn((float)HD));
  uint4 qq[REP];
#pragma unroll
  for (int h = 0; h < REP; h++) qq[h] = ((const uint4*)q2[h])[piece];
  float Mrun[REP], Lrun[REP], Orun[REP];    // threads < 128 own output dim tid of every head
#pragma unroll
  for (int h = 0; h < REP; h++) { Mrun[h] = -INFINITY; Lrun[h] = 0.f; Orun[h] = 0.f; }
  for (int c0 = p0; c0 < p1; c0 += 128) {
    if (c0 > p0) __syncthreads();           // the previous chunk's pout / wred / lred reads are done
    const bool won = c0 + wave * PW < p1;   // wave-uniform
    uint4 vr[NL];
    float sc_[REP][NL], mloc[REP];
#pragma unroll
    for (int h = 0; h < REP; h++) mloc[h] = -INFINITY;
    if (won) {
      uint4 kr[NL];
#pragma unroll
      for (int r = 0; r < NL; r++) {
        const size_t off = kv_row_off_t<PAGED>(kv, 0, kvh, min(c0 + wave * PW + 4 * r + rsub, pmax)) + piece * 8;
        kr[r] = *(const uint4*)(kb0 + off);
        vr[r] = *(const uint4*)(vb0 + off);
      }
#pragma unroll
      for (int r = 0; r < NL; r++) {
        const int p = c0 + wave * PW + 4 * r + rsub;
        uint4 kk = kr[r];
        if (p == pos) { kk = ((const uint4*)k2)[piece]; vr[r] = ((const uint4*)v2)[piece]; }
#pragma unroll
        for (int h = 0; h < REP; h++) {
          float d = dot2acc<KVDT>(kk.x, qq[h].x, 0.f);
          d = dot2acc<KVDT>(kk.y, qq[h].y, d); d = dot2acc<KVDT>(kk.z, qq[h].z, d); d = dot2acc<KVDT>(kk.w, qq[h].w, d);
          d = grp_reduce<16, OpAdd>(d);
          sc_[h][r] = (p < p1) ? d * scale : -INFINITY;
          mloc[h] = fmaxf(mloc[h], sc_[h][r]);
        }
      }
#pragma unroll
      for (int h = 0; h < REP; h++) mloc[h] = wave_max(mloc[h]);
    }
    if (lane == 0) {
#pragma unroll
      for (int h = 0; h < REP; h++) wred[h][wave] = mloc[h];
    }
    __syncthreads();
    float Mc[REP];
#pragma unroll
    for (int h = 0; h < REP; h++) {
      Mc[h] = wred[h][0];
#pragma unroll
      for (int w = 1; w < NW; w++) Mc[h] = fmaxf(Mc[h], wred[h][w]);
    }
    if (won) {
      float accv[REP][8], lsum[REP];
#pragma unroll
      for (int h = 0; h < REP; h++) {
        lsum[h] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) accv[h][q] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < NL; r++) {
        float v[8];
        unpack2<KVDT>(vr[r].x, v[0], v[1]); unpack2<KVDT>(vr[r].y, v[2], v[3]);
        unpack2<KVDT>(vr[r].z, v[4], v[5]); unpack2<KVDT>(vr[r].w, v[6], v[7]);
#pragma unroll
        for (int h = 0; h < REP; h++) {
          const float e = (sc_[h][r] == -INFINITY) ? 0.f : bz_expf(sc_[h][r] - Mc[h]);
          lsum[h] += e;
#pragma unroll
          for (int q = 0; q < 8; q++) accv[h][q] = fmaf(e, v[q], accv[h][q]);
        }
      }
#pragma unroll
      for (int h = 0; h < REP; h++) {
        lsum[h] = xrow32<OpAdd>(xrow16<OpAdd>(lsum[h]));
#pragma unroll
        for (int q = 0; q < 8; q++) accv[h][q] = xrow32<OpAdd>(xrow16<OpAdd>(accv[h][q]));
        if (lane < 16) {
          *(float4*)(&pout[wave][h][piece * 8]) = make_float4(accv[h][0], accv[h][1], accv[h][2], accv[h][3]);
          *(float4*)(&pout[wave][h][piece * 8 + 4]) = make_float4(accv[h][4], accv[h][5], accv[h][6], accv[h][7]);
        }
        if (lane == 0) lred[h][wave] = lsum[h];
      }
    } else {
      if (lane < 16) {
#pragma unroll
        for (int h = 0; h < REP; h++) {
          *(float4*)(&pout[wave][h][piece * 8]) = make_float4(0.f, 0.f, 0.f, 0.f);
          *(float4*)(&pout[wave][h][piece * 8 + 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int h = 0; h < REP; h++) lred[h][wave] = 0.f;
      }
    }
    __syncthreads();
    if (tid < 128) {
#pragma unroll
      for (int h = 0; h < REP; h++) {
        float oc = 0.f, Lc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) { oc += pout[w][h][tid]; Lc += lred[h][w]; }
        const float Mn = fmaxf(Mrun[h], Mc[h]);
        const float fa = (Mrun[h] == -INFINITY) ? 0.f : bz_expf(Mrun[h] - Mn), fb = bz_expf(Mc[h] - Mn);
        Orun[h] = Orun[h] * fa + oc * fb;
        Lrun[h] = Lrun[h] * fa + Lc * fb;
        Mrun[h] = Mn;
      }
    }
  }
  if (tid < 128) {
#pragma unroll
    for (int h = 0; h < REP; h++) {
      float* dst = ws + ((size_t)(kvh * REP + h) * nsplit + s) * ATT_PSTRIDE;
      dst[4 + tid] = Orun[h];
      if (tid == 0) { dst[0] = Mrun[h]; dst[1] = Lrun[h]; }
    }
  }
}

// merge of the live partials of head hq into outh[128] (LDS, rounded to the activation dtype); 512 threads; ns <= 128
__device__ __forceinline__ void att_merge_512(const float* __restrict__ ws, int hq, int nsplit, int ns, int act, float* wS, float* red, float (*osum)[128],
                                              float* outh) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* base = ws + (size_t)hq * nsplit * ATT_PSTRIDE;
  const int sc = min(tid, ns - 1);
  float m = base[(size_t)sc * ATT_PSTRIDE], l = base[(size_t)sc * ATT_PSTRIDE + 1];
  if (tid >= ns) { m = -INFINITY; l = 0.f; }
  const float wm = wave_max(m);
  if (lane == 0) red[wave] = wm;
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int w = 1; w < 8; w++) M = fmaxf(M, red[w]);
  const float wgt = (m == -INFINITY) ? 0.f : bz_expf(m - M);
  if (tid < 128) wS[tid] = wgt;
  const float wl = wave_sum(wgt * l);
  if (lane == 0) red[8 + wave] = wl;
  __syncthreads();
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < 8; w++) L += red[8 + w];
  const int d = tid & 127, ph = tid >> 7;
  float o = 0.f;
#pragma unroll 8
  for (int s = ph; s < ns; s += 4) o += wS[s] * base[(size_t)s * ATT_PSTRIDE + 4 + d];
  osum[ph][d] = o;
  __syncthreads();
  if (tid < 128) outh[tid] = round_act(((osum[0][tid] + osum[1][tid]) + (osum[2][tid] + osum[3][tid])) / L, act);
}

__global__ __launch_bounds__(512) void k_attn_merge(AttnArgs a, const float* __restrict__ ws, int SPL, int nsplit) {
  __shared__ float wS[128], red[16], osum[4][128], outh[128];
  const int hq = blockIdx.x;
  const int pos = a.pos[0];
  const int ns = (pos + 1 + SPL - 1) / SPL;
  att_merge_512(ws, hq, nsplit, ns, a.act, wS, red, osum, outh);
  __syncthreads();
  if (threadIdx.x < 128) a.out[(size_t)hq * 128 + threadIdx.x] = outh[threadIdx.x];
}

template <int TPW>
__global__ __launch_bounds__(512) void k_attn_merge_oproj(AttnArgs a, const float* __restrict__ ws, int SPL, int nsplit, const uint4* __restrict__ W,
                                                          const __half* __restrict__ S, const unsigned char* __restrict__ Z,
                                                          const float* __restrict__ bias, int CS, long long* acc) {
  constexpr int HD = 128;
  __shared__ float wS[128], red[16], osum[4][128];
  __shared__ __attribute__((aligned(16))) float outh[128];
  __shared__ uint4 xpl[4 * XQ_NP];
  __shared__ int4 gpar[2];
  asm volatile("" :: "s"(a.pos), "s"(a.act), "s"(a.nq), "s"(ws), "s"(SPL), "s"(nsplit), "s"(W), "s"(S), "s"(Z), "s"(bias), "s"(CS), "s"(acc),
               "s"(a.zero_buf), "s"(a.zero_n));
  const int hq = blockIdx.x / CS, cs = blockIdx.x % CS;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int pos_v = a.pos[0];
  uint4 Wb[TPW][4];
  const int t0 = (cs * 8 + wave) * TPW;
  const int K = a.nq * HD;
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const uint4* wp = W + ((size_t)(t0 + t) * (K >> 5) + hq * 4) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; c++) Wb[t][c] = ldnt(wp + c * 64);
  }
  float sc[TPW]; int zp[TPW];
  {
    const int G = K >> 7;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      sc[t] = __half2float(S[((size_t)(t0 + t) * G + hq) * 64 + lane]);
      zp[t] = Z[((size_t)(t0 + t) * G + hq) * 64 + lane];
    }
  }
  if (a.zero_buf)
    for (int z = blockIdx.x * 512 + tid; z < a.zero_n; z += gridDim.x * 512) a.zero_buf[z] = 0;
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);
  const int ns = (pos + 1 + SPL - 1) / SPL;
  att_merge_512(ws, hq, nsplit, ns, a.act, wS, red, osum, outh);
  __syncthreads();
  quant_x128<512>(outh, HD, xpl, gpar);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    double y = 0.0;
    q4g_consume(Wb[t], 0, xpl, gpar, sc[t], zp[t], y);
    const int n = (t0 + t) * 64 + lane;
    if (bias != nullptr && hq == 0) y += (double)bias[n];
    atomicAdd((unsigned long long*)(acc + n), (unsigned long long)d2fix(y, a.act));
  }
}

// split plan from the positions the launch must cover: SPL positions per block (multiple of 128), at most 128 splits
void bzk_attn_split_plan(int positions, int* SPL, int* nsplit) {
  int spl = 128;
  while ((positions + spl - 1) / spl > 128) spl *= 2;
  *SPL = spl; *nsplit = (positions + spl - 1) / spl;
  if (*nsplit < 1) *nsplit = 1;
}
int bzk_attn_split_ok(const AttnArgs& a) {
  const int rep = a.nkv > 0 ? a.nq / a.nkv : 0;
  return a.hd == 128 && (a.kv.dtype == BZ_F16 || a.kv.dtype == BZ_BF16) && !a.q_only && a.nq % a.nkv == 0 && (rep == 1 || rep == 2 || rep == 4 || rep == 8) &&
         a.rope_cur != nullptr;
}
size_t bzk_attn_split_ws_bytes(int nq) { return (size_t)nq * 128 * ATT_PSTRIDE * 4; }

int bzk_attn_split(hipStream_t s, const AttnArgs& a, int SPL, int nsplit, float* ws) {
  if (!bzk_attn_split_ok(a) || nsplit < 1 || nsplit > 128 || SPL % 128) BZ_FAIL(BZ_E_INVALID, "split attention does not apply to this shape");
  const int rep = a.nq / a.nkv;
#define LAUNCH_SP(DT, PG, R) BZ_LAUNCH("attn_split", 0.0, (k_attn_split<DT, PG, R>), dim3(a.nkv * nsplit), dim3(R == 8 ? 512 : 256), 0, s, a, SPL, nsplit, ws)
#define LAUNCH_SP_R(DT, PG) do { if (rep == 1) LAUNCH_SP(DT, PG, 1); else if (rep == 2) LAUNCH_SP(DT, PG, 2); else if (rep == 4) LAUNCH_SP(DT, PG, 4); \
                                 else LAUNCH_SP(DT, PG, 8); } while (0)
#define LAUNCH_SP_P(DT) do { if (a.kv.paged) LAUNCH_SP_R(DT, 1); else LAUNCH_SP_R(DT, 0); } while (0)
  if (a.kv.dtype == BZ_F16) LAUNCH_SP_P(BZ_F16); else LAUNCH_SP_P(BZ_BF16);
#undef LAUNCH_SP_P
#undef LAUNCH_SP_R
#undef LAUNCH_SP
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_attn_merge(hipStream_t s, const AttnArgs& a, const float* ws, int SPL, int nsplit) {
  BZ_LAUNCH("attn_merge", 0.0, k_attn_merge, dim3(a.nq), dim3(512), 0, s, a, ws, SPL, nsplit);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_attn_merge_oproj_ok(const AttnArgs& a, const LinearDev& L) { int nw; return attn_oproj_plan(a, L, nw) > 0 && nw == 8; }
int bzk_attn_merge_oproj(hipStream_t s, const AttnArgs& a, const float* ws, int SPL, int nsplit, const LinearDev& L, long long* acc) {
  int NW;
  const int CS = attn_oproj_plan(a, L, NW);
  if (CS <= 0 || NW != 8) BZ_FAIL(BZ_E_INVALID, "attn merge + o_proj fusion does not apply to this shape");
  const int TPW = (L.N / 64) / (CS * 8);
#define LAUNCH_MO(T) BZ_LAUNCH("attn_merge+o_proj", L.algo_bytes, (k_attn_merge_oproj<T>), dim3(a.nq * CS), dim3(512), 0, s, a, ws, SPL, nsplit, \
    (const uint4*)L.w, (const __half*)L.scales, (const unsigned char*)L.zeros, L.bias, CS, acc)
  if (TPW == 1) LAUNCH_MO(1); else LAUNCH_MO(2);
#undef LAUNCH_MO
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

int bzk_attn_decode(hipStream_t s, const AttnArgs& a) {
  const size_t smem = (size_t)(3 * a.hd + 8 + 256 * (a.hd + 4)) * 4;
#define LAUNCH_ATT(HD, DT) BZ_LAUNCH("attn_decode", 0.0, (k_attn_decode<HD, DT>), dim3(a.nq), dim3(256), smem, s, a)
#define LAUNCH_ATT_DT(HD) do { if (a.kv.dtype == BZ_F16) LAUNCH_ATT(HD, BZ_F16); else if (a.kv.dtype == BZ_BF16) LAUNCH_ATT(HD, BZ_BF16); \
                               else LAUNCH_ATT(HD, BZ_F32); } while (0)
  static const bool no_a64 = getenv("BZ_NO_ATTN2_HD64") != nullptr;
  if ((a.hd == 128 || (a.hd == 64 && a.rope_cur != nullptr && !no_a64)) && a.kv.dtype != BZ_F32) {
    // the 8-wave lane = (row, 8-element piece) kernel: head_dim 128, and 64 (Llama-3.2-1B) with 8 rows per wave-wide load
#define LAUNCH_A2(DT, PG, HDV) BZ_LAUNCH("attn_decode", 0.0, (k_attn2<DT, 0, 1, 8, PG, HDV>), dim3(a.nq), dim3(512), attn2_smem(8), s, a, (const uint4*)nullptr, \
    (const __half*)nullptr, (const unsigned char*)nullptr, (const float*)nullptr, 1, (long long*)nullptr)
#define LAUNCH_A2_P(DT, HDV) do { if (a.kv.paged) LAUNCH_A2(DT, 1, HDV); else LAUNCH_A2(DT, 0, HDV); } while (0)
#define LAUNCH_A2_H(DT) do { if (a.hd == 128) LAUNCH_A2_P(DT, 128); else LAUNCH_A2_P(DT, 64); } while (0)
    if (a.kv.dtype == BZ_F16) LAUNCH_A2_H(BZ_F16); else LAUNCH_A2_H(BZ_BF16);
#undef LAUNCH_A2_H
#undef LAUNCH_A2_P
#undef LAUNCH_A2
  }
  else if (a.hd == 128 && a.kv.dtype == BZ_F32 && a.rope_cur != nullptr && getenv("BZ_NO_ATTN_F32") == nullptr) {
    if (a.kv.paged) BZ_LAUNCH("attn_decode", 0.0, (k_attn2f<1, 0>), dim3(a.nq), dim3(512), 0, s, a, (const uint4*)nullptr, (const uint4*)nullptr, (const float*)nullptr, 1, (long long*)nullptr);
    else BZ_LAUNCH("attn_decode", 0.0, (k_attn2f<0, 0>), dim3(a.nq), dim3(512), 0, s, a, (const uint4*)nullptr, (const uint4*)nullptr, (const float*)nullptr, 1, (long long*)nullptr);
  }
  else if (a.hd == 64) LAUNCH_ATT_DT(64);
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
// ---- batched decode graph (cuda_graphs_batched.rs:43-257): the step's bookkeeping on the device ----------------------------------------------------
// before the forward: feed back last step's argmax, advance every sequence's position, derive its slot from its block-table row
__global__ void k_batch_advance(long long* tok, const long long* next, int* pos, int* slot, const int* table, int stride, int bs, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  tok[i] = next[i];
  const int p = pos[i] + 1;
  pos[i] = p;
  slot[i] = table[(size_t)i * stride + p / bs] * bs + p % bs;
}
// after the forward: row-wise argmax (larger value, then smaller index -- batch_argmax_to_buf) -> next[row] and the pinned token log
__global__ __launch_bounds__(256) void k_batch_argmax(const float* __restrict__ logits, int V, long long* next, long long* log, int* step, int logcap, int N) {
  __shared__ float sv[4]; __shared__ int si[4];
  const int row = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* lg = logits + (size_t)row * V;
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = tid; i < V; i += 256) { const float v = lg[i]; if (v > bv) { bv = v; bi = i; } }   // ascending i per thread: first maximum wins
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(bv, off); const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; w++) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    next[row] = bi;
    const int st = *step;
    log[(size_t)(st % logcap) * N + row] = bi;
  }
}
__global__ void k_batch_step_inc(int* step) { *step = *step + 1; }
int bzk_batch_advance(hipStream_t s, long long* tok, const long long* next, int* pos, int* slot, const int* table, int stride, int bs, int N) {
  hipLaunchKernelGGL(k_batch_advance, dim3((N + 63) / 64), dim3(64), 0, s, tok, next, pos, slot, table, stride, bs, N);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_batch_argmax(hipStream_t s, const float* logits, int V, long long* next, long long* log, int* step, int logcap, int N) {
  hipLaunchKernelGGL(k_batch_argmax, dim3(N), dim3(256), 0, s, logits, V, next, log, step, logcap, N);
  hipLaunchKernelGGL(k_batch_step_inc, dim3(1), dim3(1), 0, s, step);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
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

int bzk_argmax_partials(hipStream_t s, const float* v, long long n, float* pval, int* pidx, int nb) {
  hipLaunchKernelGGL(k_penalised_argmax, dim3(nb), dim3(256), 0, s, v, n, (const long long*)nullptr, (const int*)nullptr, 0, 1.0f, 0.f, 0.f, pval, pidx);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
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
  const float rs = rms_scale(ss, (float)n, eps);
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
    v[ia] = round_act(rope_lo(x0, x1, c, sn), act);
    v[ib] = round_act(rope_hi(x0, x1, c, sn), act);
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

// ---------------------------------------------------------------------------------------------------------
// Mamba2 single-token kernels: causal conv1d step, SSM state update + readout
// ---------------------------------------------------------------------------------------------------------
// log1p in double, one rounding to f32 (the oracle's softplus: glibc's and ocml's f32 log1pf differ in the last bit often enough to matter; both are faithful in double)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : (float)log1p((double)bz_expf(x)); }
// Mamba2 arithmetic (round 3, the oracle's definitions -- orc_mamba2.c): the conv window's dot product and the readout C . h are exactly rounded sums (double over
// exact products), the state update h dA + (dt x) B is two exact products summed in double and rounded once to f32, and y = f32(C . h) + D x with the D x product
// rounded on its own (no fused multiply-add: __fmul_rn / __fadd_rn where a contraction would change the bits)
__device__ __forceinline__ float ssm_h_update(float h, float dA, float dtx, float B) { return (float)fma((double)h, (double)dA, (double)dtx * (double)B); }

// grid = n_heads; 256 threads = 64 rows (p) x 4 state quarters.  h = R(h * dA + (dt x) B); y = R(sum_n h C + D x)
// The causal conv1d step (+ SiLU) runs in front, inside the same launch: a head's workgroup convolves its own head_dim x channels (and shifts
// their conv state, which nobody else touches) and -- redundantly, 2 d_state channels -- its group's B and C from the OLD state.  The B / C
// state is shifted by the NEXT launch (the out_proj GEMV's ConvShift duty), when every head has read it.  All global loads of the step (state
// piece, conv taps, dt) are issued before the first use: one L2 round trip instead of four.
template <int SDT, int PARTS>   // PARTS threads share a state row: 4 (256-thread blocks) or 16 (1024 threads: 4 waves per SIMD, one 16-byte piece per thread)
__global__ __launch_bounds__(64 * PARTS) void k_ssm_step(SsmArgs a) {
  constexpr int NTH = 64 * PARTS;
  __shared__ float sB[256], sC[256], sX[256];
  __shared__ double sredd[PARTS];
  const int hd = blockIdx.x, tid = threadIdx.x;
  const int NS = a.d_state, HD = a.head_dim, KC = a.conv_kernel;
  const int hpg = a.n_heads / a.n_groups;
  const int g = hd / hpg;
  const int q = tid % PARTS, nq = NS / PARTS;      // this thread's part of the state row
  const bool vec = (SDT != BZ_F32) && (nq & 7) == 0;
  // (0) the first state piece of this thread, and dt
  uint4 pre = make_uint4(0, 0, 0, 0);
  if (SDT != BZ_F32) { if (vec && tid / PARTS < HD) pre = *(const uint4*)((const unsigned short*)a.state + ((size_t)hd * HD + tid / PARTS) * NS + q * nq); }
  const float dtraw = vsrc_get(a.zx, a.dt_off + hd, a.act);
  const float dtb = a.dt_bias[hd], alog = a.A_log[hd], Dh = a.D[hd];
  // (1) conv1d step + SiLU: B, C of the group (old state, read only) and this head's x channels (state shifted here)
  for (int t = tid; t < 2 * NS + HD; t += NTH) {
    const int ch = t < NS ? a.d_inner + g * NS + t : (t < 2 * NS ? a.d_inner + a.n_groups * NS + g * NS + (t - NS) : hd * HD + (t - 2 * NS));
    float* cs = a.conv_state + (size_t)ch * (KC - 1);
    const float xr = vsrc_get(a.zx, a.x_off + ch, a.act);
    double cd = 0.0;
    if (KC == 4) {
      const float c0 = cs[0], c1 = cs[1], c2 = cs[2];
      const float4 w = *(const float4*)(a.conv_w + (size_t)ch * 4);
      cd = fma((double)c0, (double)w.x, cd); cd = fma((double)c1, (double)w.y, cd); cd = fma((double)c2, (double)w.z, cd); cd = fma((double)xr, (double)w.w, cd);
      if (t >= 2 * NS) { cs[0] = c1; cs[1] = c2; cs[2] = xr; }
    } else {
      for (int j = 0; j < KC - 1; j++) cd = fma((double)cs[j], (double)a.conv_w[(size_t)ch * KC + j], cd);
      cd = fma((double)xr, (double)a.conv_w[(size_t)ch * KC + KC - 1], cd);
      if (t >= 2 * NS) { for (int j = 0; j + 1 < KC - 1; j++) cs[j] = cs[j + 1]; cs[KC - 2] = xr; }
    }
    float c = round_act(__fadd_rn((float)cd, a.conv_b[ch]), a.act);
    c = round_act(silu_f(c), a.act);
    if (t < NS) sB[t] = c; else if (t < 2 * NS) sC[t - NS] = c; else sX[t - 2 * NS] = c;
  }
  const float dt = round_act(softplus_f(round_act(dtraw + dtb, a.act)), a.act);
  const float dA = bz_expf(__fmul_rn(dt, -bz_expf(alog)));
  __syncthreads();
  if (a.conv_out) {      // op-level entry points: the conv output as the forward path computed it (B / C by the first head of their group)
    if (hd % hpg == 0)
      for (int t = tid; t < NS; t += NTH) { a.conv_out[a.d_inner + g * NS + t] = sB[t]; a.conv_out[a.d_inner + a.n_groups * NS + g * NS + t] = sC[t]; }
    for (int t = tid; t < HD; t += NTH) a.conv_out[hd * HD + t] = sX[t];
    if (a.conv_only) return;
  }
  double vsq = 0.0;
  for (int p0 = 0; p0 < HD; p0 += 64) {
    const int p = p0 + tid / PARTS;
    double acc = 0.0; float xv = 0.f;
    if (p < HD) {
      xv = sX[p];
      const float dtx = __fmul_rn(dt, xv);
      const size_t off = ((size_t)hd * HD + p) * NS + q * nq;
      if (vec) {
        // 16-bit state: 8 elements per 16-byte load / store
        if constexpr (SDT != BZ_F32) for (int n = 0; n < nq; n += 8) {
          uint4* sp = (uint4*)((unsigned short*)a.state + off + n);
          const uint4 raw = (p0 == 0 && n == 0) ? pre : *sp;
          unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float h0, h1;
            unpack2<SDT>(u[j], h0, h1);
            h0 = round_act(ssm_h_update(h0, dA, dtx, sB[q * nq + n + 2 * j]), a.act);
            h1 = round_act(ssm_h_update(h1, dA, dtx, sB[q * nq + n + 2 * j + 1]), a.act);
            acc = fma((double)h0, (double)sC[q * nq + n + 2 * j], acc);
            acc = fma((double)h1, (double)sC[q * nq + n + 2 * j + 1], acc);
            u[j] = pack2<SDT>(h0, h1);
          }
          *sp = make_uint4(u[0], u[1], u[2], u[3]);
        }
      } else {
        for (int n = 0; n < nq; n++) {
          float hcur;
          if (SDT == BZ_F32) hcur = ((float*)a.state)[off + n];
          else if (SDT == BZ_F16) hcur = __half2float(((__half*)a.state)[off + n]);
          else hcur = __uint_as_float((unsigned)((unsigned short*)a.state)[off + n] << 16);
          const float hn = round_act(ssm_h_update(hcur, dA, dtx, sB[q * nq + n]), a.act);
          acc = fma((double)hn, (double)sC[q * nq + n], acc);
          if (SDT == BZ_F32) ((float*)a.state)[off + n] = hn;
          else if (SDT == BZ_F16) ((__half*)a.state)[off + n] = f16_cvt(hn);
          else ((unsigned short*)a.state)[off + n] = (unsigned short)(__float_as_uint(bf16_round(hn)) >> 16);
        }
      }
    }
    acc += dpp_get<DPP_XOR1>(acc); acc += dpp_get<DPP_XOR2>(acc);                       // the PARTS lanes of a state row (4 or 16, aligned)
    if (PARTS == 16) { acc += dpp_get<DPP_HMIRROR>(acc); acc += dpp_get<DPP_MIRROR>(acc); }
    if (p < HD && q == 0) {
      float yv = round_act(__fadd_rn((float)acc, __fmul_rn(Dh, xv)), a.act);
      if (a.gate) { yv = round_act(__fmul_rn(yv, round_act(silu_f(vsrc_get(a.zx, a.z_off + hd * HD + p, a.act)), a.act)), a.act); vsq += (double)__fmul_rn(yv, yv); }
      a.y[hd * HD + p] = yv;
    }
  }
  if (a.gate) {
    // the head's sum of squares, exact (f32 squares summed in double); handed on as hi + lo floats: vss[2 head], vss[2 head + 1] (the group sum of the consumer
    // is then the oracle's double sum of the group's squares, rounded once)
    vsq = wave_sum_d(vsq);
    if ((tid & 63) == 0) sredd[tid >> 6] = vsq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < PARTS; w++) t += sredd[w];
      const float hi = (float)t;
      a.vss[2 * hd] = hi; a.vss[2 * hd + 1] = (float)(t - (double)hi);
    }
  }
}
int bzk_conv_shift(hipStream_t s, const ConvShift& c, int act) {
  if (c.n > 0) hipLaunchKernelGGL(k_conv_shift, dim3((c.n + 255) / 256), dim3(256), 0, s, c, act);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_ssm_step(hipStream_t s, const SsmArgs& a) {
  if (a.head_dim > 256 || a.conv_kernel < 2) BZ_FAIL(BZ_E_UNSUPPORTED, "ssm_step: head_dim %d / conv kernel %d unsupported", a.head_dim, a.conv_kernel);
  if (a.d_state > 256 || (a.d_state & 3) || a.n_heads % a.n_groups) BZ_FAIL(BZ_E_UNSUPPORTED, "ssm_step: d_state %d / groups %d unsupported", a.d_state, a.n_groups);
  const double bytes = 2.0 * a.n_heads * a.head_dim * a.d_state * (a.sdt == BZ_F32 ? 4 : 2);
  // 16 threads per state row when a part is then a whole number of 16-byte pieces (d_state 128 with a 16-bit state) or the state is f32
  const bool wide = a.d_state % 16 == 0 && (a.sdt == BZ_F32 || (a.d_state / 16) % 8 == 0);
#define LAUNCH_SSM(SDT) do { if (wide) BZ_LAUNCH("mamba2_ssm_step", bytes, (k_ssm_step<SDT, 16>), dim3(a.n_heads), dim3(1024), 0, s, a); \
                             else BZ_LAUNCH("mamba2_ssm_step", bytes, (k_ssm_step<SDT, 4>), dim3(a.n_heads), dim3(256), 0, s, a); } while (0)
  if (a.sdt == BZ_F32) LAUNCH_SSM(BZ_F32); else if (a.sdt == BZ_F16) LAUNCH_SSM(BZ_F16); else LAUNCH_SSM(BZ_BF16);
#undef LAUNCH_SSM
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DeepSeek-V2: MLA decode attention over the compressed-latent cache (weight-absorbed form) + MoE router / combine
// ---------------------------------------------------------------------------------------------------------
// One workgroup per head.  Cache row = [latent c (rank) | roped k_pe (rope)] in the cache dtype; n_kv = 1.
//   c = rmsnorm(kva[:rank]) ; kpe = R(rope(kva[rank:]))                        (block 0 appends the row at `pos`)
//   qabs = R(Wuk_h^T q_nope) ; s_t = (qabs . c_t + qpe . kpe_t) * scale ; p = softmax(s)
//   olat = R(sum_t p_t c_t) ; out_h = R(Wuv_h olat)
// LDS: ccur[rank] kcur[rope] qn[nope] qp[rope] qabs[rank] part[4][rank] red[8] sc[len]
// compile-time dtype loaders: a run-time dtype switch puts every load behind a branch, and hipcc then drains vmcnt after each one
template <int DT>
__device__ __forceinline__ void ld8t(const void* base, size_t off, float (&o)[8]) { load8<DT>(base, off, true, o); }
template <int DT>
__device__ __forceinline__ float ld1t(const void* base, size_t off) {
  if constexpr (DT == BZ_F16) return __half2float(((const __half*)base)[off]);
  else if constexpr (DT == BZ_BF16) return __uint_as_float((unsigned)((const unsigned short*)base)[off] << 16);
  else return ((const float*)base)[off];
}

// NW waves per head: 16 (1024 threads) when the per-wave row ranges divide evenly -- with one workgroup per head (16 of them for V2-Lite) the
// kernel is starved of parallelism, and 4 waves per SIMD let one wave's dependent chain hide under the others'
template <int NW> __device__ __forceinline__ float block_sum_nw(float v, float* red) {   // deterministic; red: LDS float[NW]
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w += 2) t += red[w] + red[w + 1];
  __syncthreads();
  return t;
}
// SPLIT (decode): grid.y = context slice.  Sixteen one-per-head workgroups each walked the whole cache twice (40 us per layer at 650 positions, 16 CUs
// busy): slice s of head h now takes positions [s ts, (s+1) ts) (the last slice also the current token), runs the same q absorption, scores and
// softmax over ITS positions, and leaves (unnormalised latent sum, local max, local sum) in the workspace; k_mla_merge rescales and sums the
// slices in slice order, normalises once, rounds, and applies Wuv.
template <int NCH, int DT, int NW, int BATCH, int SPLIT = 0>   // NCH: 512-column chunks of the latent, 1 (rank <= 512) or 2 (rank <= 1024); DT: dtype of kv_b and of the cache; BATCH: prompt rows (grid.y = token)
__global__ __launch_bounds__(NW * 64) void k_mla_attn(MlaArgs a) {
  constexpr int NTH = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = a.rank, DN = a.nope, DR = a.rope, DV = a.vdim;
  float* ccur = lds; float* kcur = ccur + R; float* qn = kcur + DR; float* qp = qn + DN; float* qabs = qp + DR;
  float* part = qabs + R; float* red = part + NW * R; float* sc = red + 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int tok = BATCH ? blockIdx.y : 0;
  const int pos = BATCH ? a.pos0 + tok : a.pos[0], len = pos + 1;
  const int nc = BATCH ? len : pos;                 // positions read from the cache (the batched form finds the token's own row there too)
  const int ts = SPLIT ? (nc + a.nsplit - 1) / a.nsplit : nc;                      // cached positions per slice
  const int tlo = SPLIT ? min((int)blockIdx.y * ts, nc) : 0, thi = SPLIT ? min(tlo + ts, nc) : nc;
  const bool own_cur = !BATCH && (!SPLIT || (int)blockIdx.y == a.nsplit - 1);      // the slice that holds the current token
  const int nloc = thi - tlo + (own_cur ? 1 : 0);   // scores of this workgroup: sc[t - tlo], the current token last
  const int QH = DN + DR, qoff = hd * QH, coff = a.n_heads * QH;
  const float* qrow = BATCH ? (const float*)a.qkv.p + (size_t)tok * a.q_stride : nullptr;
  auto qsrc = [&](int i) -> float { return BATCH ? qrow[i] : vsrc_get(a.qkv, i, a.act); };
  auto csrc = [&](int i) -> float { return a.kva ? a.kva[i] : vsrc_get(a.qkv, coff + i, a.act); };   // latent | k_pe of the current token (decode)
  const size_t rowbase = (size_t)a.layer * a.kv.layer_stride;
  const int Wd = R + DR;
  auto rowoff = [&](int p) -> size_t {
    if (a.kv.paged) return rowbase + ((size_t)a.kv.block_table[p / a.kv.bs] * a.kv.bs + (p % a.kv.bs)) * Wd;
    return rowbase + (size_t)p * Wd;
  };
  const size_t wrow0 = (size_t)hd * (DN + DV);
  constexpr int RIF = (NW == 4 ? 16 : 8) / NCH, TIF = (NW == 4 ? 8 : 4) / NCH;   // weight rows / cache rows a wave keeps in flight
  bool con[NCH]; int colc[NCH];                      // this lane's 8 columns per chunk (clamped when beyond the rank)
#pragma unroll
  for (int c = 0; c < NCH; c++) { const int col = c * 512 + lane * 8; con[c] = col < R; colc[c] = con[c] ? col : 0; }

  // ---- qabs partials first (weights only depend on the head): wave w takes nope rows [w DN/4, (w+1) DN/4), 8 rows in flight ----
  const int d0 = wave * (DN / NW), d1 = d0 + DN / NW;
  float w0[RIF][NCH][8];
#pragma unroll
  for (int u = 0; u < RIF; u++)
#pragma unroll
    for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d0 + u, d1 - 1)) * R + colc[c], w0[u][c]);
  __builtin_amdgcn_sched_barrier(0);

  // ---- current token: latent norm, k_pe / q_pe rope, q_nope ----
  const float* cr = a.cos_t + (size_t)pos * (DR / 2); const float* sr = a.sin_t + (size_t)pos * (DR / 2);
  if (!BATCH) {
    float ss = 0.f;
    for (int r = tid; r < R; r += NTH) { const float v = csrc(r); ccur[r] = v; ss += v * v; }
    ss = block_sum_nw<NW>(ss, red);
    const float rs = rms_scale(ss, (float)R, a.eps);
    for (int r = tid; r < R; r += NTH) ccur[r] = round_act(a.kv_norm[r] * round_act(ccur[r] * rs, a.act), a.act);
  }
  for (int j = tid; j < DR / 2; j += NTH) {
    const float c = cr[j], s = sr[j];
    if (!BATCH) {
      const float k0 = csrc(R + 2 * j), k1 = csrc(R + 2 * j + 1);
      kcur[2 * j] = round_act(rope_lo(k0, k1, c, s), a.act); kcur[2 * j + 1] = round_act(rope_hi(k0, k1, c, s), a.act);
    }
    const float x0 = qsrc(qoff + DN + 2 * j), x1 = qsrc(qoff + DN + 2 * j + 1);
    qp[2 * j] = round_act(rope_lo(x0, x1, c, s), a.act); qp[2 * j + 1] = round_act(rope_hi(x0, x1, c, s), a.act);
  }
  for (int d = tid; d < DN; d += NTH) qn[d] = qsrc(qoff + d);
  __syncthreads();
  if (!BATCH && hd == 0 && (!SPLIT || blockIdx.y == 0)) {
    size_t wo;
    if (a.kv.paged) { const int slot = a.kv.slot ? a.kv.slot[0] : (a.kv.block_table[pos / a.kv.bs] * a.kv.bs + pos % a.kv.bs); wo = rowbase + (size_t)slot * Wd; }
    else wo = rowbase + (size_t)pos * Wd;
    for (int i = tid; i < Wd; i += NTH) kv_st(a.kv.k, wo + i, a.kv.dtype, i < R ? ccur[i] : kcur[i - R]);
  }
  {
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 8; e++) acc[c][e] = 0.f;
    for (int d = d0; d < d1; d += RIF) {
      if (d > d0) {
#pragma unroll
        for (int u = 0; u < RIF; u++)
#pragma unroll
          for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d + u, d1 - 1)) * R + colc[c], w0[u][c]);
      }
#pragma unroll
      for (int u = 0; u < RIF; u++) {
        const float qd = (d + u < d1) ? qn[d + u] : 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int e = 0; e < 8; e++) acc[c][e] += qd * w0[u][c][e];
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (con[c])
#pragma unroll
        for (int e = 0; e < 8; e++) part[wave * R + colc[c] + e] = acc[c][e];
  }
  __syncthreads();
  for (int r = tid; r < R; r += NTH) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) t += part[w * R + r] + part[(w + 1) * R + r];
    qabs[r] = round_act(t, a.act);
  }
  __syncthreads();
  // ---- scores: wave w takes cached tokens w, w+4, ... four at a time; the current token comes from LDS ----
  float qa[NCH][8];
  const float qpl = (lane < DR) ? qp[lane] : 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int e = 0; e < 8; e++) qa[c][e] = con[c] ? qabs[colc[c] + e] : 0.f;
  float cv[TIF][NCH][8];                              // (kept for the latent sum when the slice is a single batch of NW * TIF positions)
  const bool single = SPLIT && thi - tlo <= NW * TIF;
  for (int t0 = tlo + wave; t0 < thi; t0 += NW * TIF) {
    float kp[TIF];
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      const size_t ro = rowoff(min(t0 + NW * u, thi - 1));
#pragma unroll
      for (int c = 0; c < NCH; c++) ld8t<DT>(a.kv.k, ro + colc[c], cv[u][c]);
      kp[u] = ld1t<DT>(a.kv.k, ro + R + min(lane, DR - 1));
    }
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      float dsum = (lane < DR) ? qpl * kp[u] : 0.f;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) dsum += qa[c][e] * cv[u][c][e];
      dsum = wave_sum(dsum);
      if (lane == 0 && t0 + NW * u < thi) sc[t0 + NW * u - tlo] = dsum * a.scale;
    }
  }
  if (own_cur && wave == 0) {
    float dsum = (lane < DR) ? qpl * kcur[lane] : 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (con[c])
#pragma unroll
        for (int e = 0; e < 8; e++) dsum += qa[c][e] * ccur[colc[c] + e];
    dsum = wave_sum(dsum);
    if (lane == 0) sc[thi - tlo] = dsum * a.scale;
  }
  // first rows of Wuv for the output phase: in flight during the softmax and the latent sum
  const int v0 = wave * (DV / NW), v1 = v0 + DV / NW;
  if (!SPLIT) {
#pragma unroll
    for (int u = 0; u < RIF; u++)
#pragma unroll
      for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(v0 + u, v1 - 1)) * R + colc[c], w0[u][c]);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < nloc; t += NTH) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NW; w++) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float psum = 0.f;
  for (int t = tid; t < nloc; t += NTH) { const float p = bz_expf(sc[t] - mx); sc[t] = p; psum += p; }
  psum = block_sum_nw<NW>(psum, red);
  const float inv = div_rn(1.0f, psum);
  // ---- olat = R(sum_t p_t c_t * inv) ----
  {
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 8; e++) acc[c][e] = 0.f;
    for (int t0 = tlo + wave; t0 < thi; t0 += NW * TIF) {
      if (!single) {
#pragma unroll
        for (int u = 0; u < TIF; u++) {
          const size_t ro = rowoff(min(t0 + NW * u, thi - 1));
#pragma unroll
          for (int c = 0; c < NCH; c++) ld8t<DT>(a.kv.k, ro + colc[c], cv[u][c]);
        }
      }
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const float p = (t0 + NW * u < thi) ? sc[t0 + NW * u - tlo] : 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int e = 0; e < 8; e++) acc[c][e] += p * cv[u][c][e];
      }
    }
    if (own_cur && wave == ((thi - tlo) % NW)) {       // the current token, in the wave that would own it in token order
      const float p = sc[thi - tlo];
#pragma unroll
      for (int c = 0; c < NCH; c++)
        if (con[c])
#pragma unroll
          for (int e = 0; e < 8; e++) acc[c][e] += p * ccur[colc[c] + e];
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (con[c])
#pragma unroll
        for (int e = 0; e < 8; e++) part[wave * R + colc[c] + e] = acc[c][e];
  }
  __syncthreads();
  if (SPLIT) {
    // this slice's share: sum_t exp(s_t - mx) c_t (unrounded), mx, sum_t exp(s_t - mx); an empty slice leaves (0, -inf, 0)
    float* wsp = a.ws + ((size_t)hd * a.nsplit + blockIdx.y) * (R + 2);
    for (int r = tid; r < R; r += NTH) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w += 2) t += part[w * R + r] + part[(w + 1) * R + r];
      wsp[r] = nloc > 0 ? t : 0.f;
    }
    if (tid == 0) { wsp[R] = nloc > 0 ? mx : -INFINITY; wsp[R + 1] = nloc > 0 ? psum : 0.f; }
    return;
  }
  for (int r = tid; r < R; r += NTH) {   // qabs now holds olat
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) t += part[w * R + r] + part[(w + 1) * R + r];
    qabs[r] = round_act(t * inv, a.act);
  }
  __syncthreads();
  // ---- out_h = R(Wuv olat): wave w takes v rows [w DV/4, (w+1) DV/4), 8 rows in flight ----
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int e = 0; e < 8; e++) qa[c][e] = con[c] ? qabs[colc[c] + e] : 0.f;
  for (int d = v0; d < v1; d += RIF) {
    if (d > v0) {
#pragma unroll
      for (int u = 0; u < RIF; u++)
#pragma unroll
        for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(d + u, v1 - 1)) * R + colc[c], w0[u][c]);
    }
#pragma unroll
    for (int u = 0; u < RIF; u++) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) s += w0[u][c][e] * qa[c][e];
      s = wave_sum(s);
      if (lane == 0 && d + u < v1) a.out[(BATCH ? (size_t)tok * a.out_stride : 0) + hd * DV + d + u] = round_act(s, a.act);
    }
  }
}

// Merge of the context slices + Wuv.  grid = (n_heads, 4): workgroup (h, q) rebuilds olat_h = R(sum_s e^{m_s - M} part_s / sum_s e^{m_s - M} l_s) (slice
// order) and computes rows [q DV/4, (q+1) DV/4) of out_h = R(Wuv_h olat_h), 8 weight rows per wave in flight.
template <int NCH, int DT>
__global__ __launch_bounds__(256) void k_mla_merge(MlaArgs a) {
  __shared__ float olat[1024];
  __shared__ float wgt[64];
  const int R = a.rank, DN = a.nope, DV = a.vdim, NSP = a.nsplit;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int rows = DV / 4, v0 = blockIdx.y * rows + wave * (rows / 4), v1 = v0 + rows / 4;
  const size_t wrow0 = (size_t)hd * (DN + DV);
  bool con[NCH]; int colc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) { const int col = c * 512 + lane * 8; con[c] = col < R; colc[c] = con[c] ? col : 0; }
  constexpr int RIF = 8 / NCH;
  float w0[RIF][NCH][8];
#pragma unroll
  for (int u = 0; u < RIF; u++)
#pragma unroll
    for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(v0 + u, v1 - 1)) * R + colc[c], w0[u][c]);
  __builtin_amdgcn_sched_barrier(0);
  const float* wsp = a.ws + (size_t)hd * NSP * (R + 2);
  {
    // slice weights e^{m_s - M} and 1 / L: lane s holds slice s (NSP <= 62), every wave computes the same values (fixed reduction trees)
    const int sl = min(lane, NSP - 1);
    const float ms = wsp[(size_t)sl * (R + 2) + R], ls = wsp[(size_t)sl * (R + 2) + R + 1];
    const bool on = lane < NSP && ls > 0.f;
    const float M = wave_max(on ? ms : -INFINITY);
    const float wv = on ? bz_expf(ms - M) : 0.f;
    const float L = wave_sum(wv * ls);
    if (wave == 0) { wgt[lane] = wv; if (lane == 63) wgt[63] = div_rn(1.0f, L); }
  }
  __syncthreads();
  const float inv = wgt[63];
  for (int r = tid; r < R; r += 256) {
    float t = 0.f;
    for (int s0 = 0; s0 < NSP; s0 += 8) {        // eight slices' loads in flight (clamped index, zero weight beyond the last slice)
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) pv[j] = wsp[(size_t)min(s0 + j, NSP - 1) * (R + 2) + r];
#pragma unroll
      for (int j = 0; j < 8; j++) t += (s0 + j < NSP ? wgt[s0 + j] : 0.f) * pv[j];
    }
    olat[r] = round_act(t * inv, a.act);
  }
  __syncthreads();
  float qa[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int e = 0; e < 8; e++) qa[c][e] = con[c] ? olat[colc[c] + e] : 0.f;
  for (int d = v0; d < v1; d += RIF) {
    if (d > v0) {
#pragma unroll
      for (int u = 0; u < RIF; u++)
#pragma unroll
        for (int c = 0; c < NCH; c++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(d + u, v1 - 1)) * R + colc[c], w0[u][c]);
    }
#pragma unroll
    for (int u = 0; u < RIF; u++) {
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) sacc += w0[u][c][e] * qa[c][e];
      sacc = wave_sum(sacc);
      if (lane == 0 && d + u < v1) a.out[hd * DV + d + u] = round_act(sacc, a.act);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// EXACT decode MLA (round 3): k_mla_attn<SPLIT> / k_mla_merge with every sum as the oracle defines it (orc_dsv2.c: exactly rounded -- double over exact products, one
// rounding to f32) and ONE maximum over the whole context: the context slices of a head exchange their local maxima through a device word and wait for each other
// (all n_heads x nsplit workgroups are resident: one per CU), so that p_t = exp(s_t - M) is the oracle's weight and the slices' partial sums simply add up in double
// -- the f32 form rescales slice partials by exp(m_slice - M), which is not the same number.  rank <= 512, 16 waves, 16-bit cache / kv_b.
//   ws (double): [n_heads][nsplit][rank + 1] = partial latent sums | partial weight sum        sync (unsigned): [n_heads][2] = ordered-int maximum | arrivals (k_mla_merge_x resets both)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }     // order-preserving map
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
template <int NW> __device__ __forceinline__ double block_sum_nw_d(double v, double* redd) {   // deterministic; redd: LDS double[NW]
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NW; w++) t += redd[w];
  __syncthreads();
  return t;
}
template <int DT>
__global__ __launch_bounds__(1024) void k_mla_attn_x(MlaArgs a, double* __restrict__ wsd, unsigned* __restrict__ sync, unsigned* __restrict__ err) {
  constexpr int NW = 16, NTH = 1024, RIF = 8, TIF = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = a.rank, DN = a.nope, DR = a.rope;
  float* ccur = lds; float* kcur = ccur + R; float* qn = kcur + DR; float* qp = qn + DN; float* qabs = qp + DR;
  double* partd = (double*)(qabs + R + ((2 * R + 2 * DR + DN) & 1));      // [NW][R], 8-byte aligned
  double* redd = partd + NW * R;                                            // [NW]
  float* red = (float*)(redd + NW); float* sc = red + 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int pos = a.pos[0], nc = pos;               // cached positions 0 .. pos - 1; the current token comes from LDS
  const int ts = (nc + a.nsplit - 1) / a.nsplit;
  const int tlo = min((int)blockIdx.y * ts, nc), thi = min(tlo + ts, nc);
  const bool own_cur = (int)blockIdx.y == a.nsplit - 1;
  const int nloc = thi - tlo + (own_cur ? 1 : 0);
  const int QH = DN + DR, qoff = hd * QH, coff = a.n_heads * QH;
  auto qsrc = [&](int i) -> float { return vsrc_get(a.qkv, i, a.act); };
  auto csrc = [&](int i) -> float { return a.kva ? a.kva[i] : vsrc_get(a.qkv, coff + i, a.act); };
  const size_t rowbase = (size_t)a.layer * a.kv.layer_stride;
  const int Wd = R + DR;
  auto rowoff = [&](int p) -> size_t {
    if (a.kv.paged) return rowbase + ((size_t)a.kv.block_table[p / a.kv.bs] * a.kv.bs + (p % a.kv.bs)) * Wd;
    return rowbase + (size_t)p * Wd;
  };
  const size_t wrow0 = (size_t)hd * (DN + a.vdim);
  const int col = lane * 8;
  const bool con = col < R;
  const int colc = con ? col : 0;
  // ---- qabs partials first (weights only depend on the head): wave w takes nope rows [w DN/16, (w+1) DN/16) ----
  const int d0 = wave * (DN / NW), d1 = d0 + DN / NW;
  float w0[RIF][8];
#pragma unroll
  for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d0 + u, d1 - 1)) * R + colc, w0[u]);
  __builtin_amdgcn_sched_barrier(0);
  // ---- current token: latent norm (exact sum of squares), k_pe / q_pe rope, q_nope ----
  const float* cr = a.cos_t + (size_t)pos * (DR / 2); const float* sr = a.sin_t + (size_t)pos * (DR / 2);
  {
    double ssd = 0.0;
    for (int r = tid; r < R; r += NTH) { const float v = csrc(r); ccur[r] = v; ssd += (double)__fmul_rn(v, v); }
    ssd = block_sum_nw_d<NW>(ssd, redd);
    const float rs = rms_scale((float)ssd, (float)R, a.eps);
    for (int r = tid; r < R; r += NTH) ccur[r] = round_act(__fmul_rn(a.kv_norm[r], round_act(__fmul_rn(ccur[r], rs), a.act)), a.act);
  }
  for (int j = tid; j < DR / 2; j += NTH) {
    const float c = cr[j], s = sr[j];
    const float k0 = csrc(R + 2 * j), k1 = csrc(R + 2 * j + 1);
    kcur[2 * j] = round_act(rope_lo(k0, k1, c, s), a.act); kcur[2 * j + 1] = round_act(rope_hi(k0, k1, c, s), a.act);
    const float x0 = qsrc(qoff + DN + 2 * j), x1 = qsrc(qoff + DN + 2 * j + 1);
    qp[2 * j] = round_act(rope_lo(x0, x1, c, s), a.act); qp[2 * j + 1] = round_act(rope_hi(x0, x1, c, s), a.act);
  }
  for (int d = tid; d < DN; d += NTH) qn[d] = qsrc(qoff + d);
  __syncthreads();
  if (hd == 0 && blockIdx.y == 0) {
    size_t wo;
    if (a.kv.paged) { const int slot = a.kv.slot ? a.kv.slot[0] : (a.kv.block_table[pos / a.kv.bs] * a.kv.bs + pos % a.kv.bs); wo = rowbase + (size_t)slot * Wd; }
    else wo = rowbase + (size_t)pos * Wd;
    for (int i = tid; i < Wd; i += NTH) kv_st(a.kv.k, wo + i, a.kv.dtype, i < R ? ccur[i] : kcur[i - R]);
  }
  {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // 16-bit q x 16-bit weight: the f32 product is exact, the double carries the sum
    for (int d = d0; d < d1; d += RIF) {
      if (d > d0) {
#pragma unroll
        for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d + u, d1 - 1)) * R + colc, w0[u]);
      }
#pragma unroll
      for (int u = 0; u < RIF; u++) {
        const float qd = (d + u < d1) ? qn[d + u] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += (double)__fmul_rn(qd, w0[u][e]);
      }
    }
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) partd[wave * R + colc + e] = acc[e];
  }
  __syncthreads();
  for (int r = tid; r < R; r += NTH) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) t += partd[w * R + r];
    qabs[r] = round_act((float)t, a.act);
  }
  __syncthreads();
  // ---- scores: wave w takes cached tokens tlo + w, + 16, ...; exact sums (f32 products of 16-bit values are exact) ----
  float qa[8];
  const float qpl = (lane < DR) ? qp[lane] : 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) qa[e] = con ? qabs[colc + e] : 0.f;
  for (int t0 = tlo + wave; t0 < thi; t0 += NW * TIF) {
    float cv[TIF][8], kp[TIF];
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      const size_t ro = rowoff(min(t0 + NW * u, thi - 1));
      ld8t<DT>(a.kv.k, ro + colc, cv[u]);
      kp[u] = ld1t<DT>(a.kv.k, ro + R + min(lane, DR - 1));
    }
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      double dsum = (lane < DR) ? (double)__fmul_rn(qpl, kp[u]) : 0.0;
#pragma unroll
      for (int e = 0; e < 8; e++) dsum += (double)__fmul_rn(qa[e], cv[u][e]);
      dsum = wave_sum_d(dsum);
      if (lane == 0 && t0 + NW * u < thi) sc[t0 + NW * u - tlo] = __fmul_rn((float)dsum, a.scale);
    }
  }
  if (own_cur && wave == 0) {
    double dsum = (lane < DR) ? (double)__fmul_rn(qpl, kcur[lane]) : 0.0;
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) dsum += (double)__fmul_rn(qa[e], ccur[colc + e]);
    dsum = wave_sum_d(dsum);
    if (lane == 0) sc[thi - tlo] = __fmul_rn((float)dsum, a.scale);
  }
  __syncthreads();
  // ---- the maximum over the WHOLE context: local maximum -> device word of the head -> wait for the other slices ----
  float mx = -INFINITY;
  for (int t = tid; t < nloc; t += NTH) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float m2 = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) m2 = fmaxf(m2, red[w]);
    unsigned* sw = sync + 2 * hd;
    if (nloc > 0) __hip_atomic_fetch_max(sw, f2ord(m2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sw + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // release: the maximum above is visible to whoever sees this arrival
    const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();
    bool ok = true;
    while (__hip_atomic_load(sw + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.nsplit) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)__builtin_amdgcn_s_memrealtime() - t_start > 2000000) { ok = false; break; }      // 20 ms at 100 MHz: a slice is missing -> flag, do not hang
    }
    if (!ok) atomicExch(err, 1u);
    red[0] = ord2f(__hip_atomic_load(sw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
  }
  __syncthreads();
  const float M = red[0];
  __syncthreads();
  double psum = 0.0;
  for (int t = tid; t < nloc; t += NTH) { const float pe = bz_expf(sc[t] - M); sc[t] = pe; psum += (double)pe; }
  psum = block_sum_nw_d<NW>(psum, redd);
  // ---- partial latent sum: sum_t p_t c_t in double (an f32 weight times a 16-bit value is exact in double) ----
  {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int t0 = tlo + wave; t0 < thi; t0 += NW * TIF) {
      float cv[TIF][8];
#pragma unroll
      for (int u = 0; u < TIF; u++) ld8t<DT>(a.kv.k, rowoff(min(t0 + NW * u, thi - 1)) + colc, cv[u]);
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const double pw = (t0 + NW * u < thi) ? (double)sc[t0 + NW * u - tlo] : 0.0;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = fma(pw, (double)cv[u][e], acc[e]);
      }
    }
    if (own_cur && wave == 0) {
      const double pw = (double)sc[thi - tlo];
      if (con)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = fma(pw, (double)ccur[colc + e], acc[e]);
    }
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) partd[wave * R + colc + e] = acc[e];
  }
  __syncthreads();
  double* wsp = wsd + ((size_t)hd * a.nsplit + blockIdx.y) * (R + 1);
  for (int r = tid; r < R; r += NTH) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) t += partd[w * R + r];
    wsp[r] = t;
  }
  if (tid == 0) wsp[R] = psum;
}
template <int DT>
__global__ __launch_bounds__(1024) void k_mla_scores_x(MlaArgs a, float* __restrict__ scw, float* __restrict__ mxw, int SCS) {
  constexpr int NW = 16, NTH = 1024, RIF = 8, TIF = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = a.rank, DN = a.nope, DR = a.rope;
  float* ccur = lds; float* kcur = ccur + R; float* qn = kcur + DR; float* qp = qn + DN; float* qabs = qp + DR;
  double* partd = (double*)(qabs + R + ((2 * R + 2 * DR + DN) & 1));      // [NW][R], 8-byte aligned
  double* redd = partd + NW * R;                                            // [NW]
  float* red = (float*)(redd + NW); float* sc = red + 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int pos = a.pos[0], nc = pos;               // cached positions 0 .. pos - 1; the current token comes from LDS
  const int ts = (nc + a.nsplit - 1) / a.nsplit;
  const int tlo = min((int)blockIdx.y * ts, nc), thi = min(tlo + ts, nc);
  const bool own_cur = (int)blockIdx.y == a.nsplit - 1;
  const int nloc = thi - tlo + (own_cur ? 1 : 0);
  const int QH = DN + DR, qoff = hd * QH, coff = a.n_heads * QH;
  auto qsrc = [&](int i) -> float { return vsrc_get(a.qkv, i, a.act); };
  auto csrc = [&](int i) -> float { return a.kva ? a.kva[i] : vsrc_get(a.qkv, coff + i, a.act); };
  const size_t rowbase = (size_t)a.layer * a.kv.layer_stride;
  const int Wd = R + DR;
  auto rowoff = [&](int p) -> size_t {
    if (a.kv.paged) return rowbase + ((size_t)a.kv.block_table[p / a.kv.bs] * a.kv.bs + (p % a.kv.bs)) * Wd;
    return rowbase + (size_t)p * Wd;
  };
  const size_t wrow0 = (size_t)hd * (DN + a.vdim);
  const int col = lane * 8;
  const bool con = col < R;
  const int colc = con ? col : 0;
  // ---- qabs partials first (weights only depend on the head): wave w takes nope rows [w DN/16, (w+1) DN/16) ----
  const int d0 = wave * (DN / NW), d1 = d0 + DN / NW;
  float w0[RIF][8];
#pragma unroll
  for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d0 + u, d1 - 1)) * R + colc, w0[u]);
  __builtin_amdgcn_sched_barrier(0);
  // ---- current token: latent norm (exact sum of squares), k_pe / q_pe rope, q_nope ----
  const float* cr = a.cos_t + (size_t)pos * (DR / 2); const float* sr = a.sin_t + (size_t)pos * (DR / 2);
  {
    double ssd = 0.0;
    for (int r = tid; r < R; r += NTH) { const float v = csrc(r); ccur[r] = v; ssd += (double)__fmul_rn(v, v); }
    ssd = block_sum_nw_d<NW>(ssd, redd);
    const float rs = rms_scale((float)ssd, (float)R, a.eps);
    for (int r = tid; r < R; r += NTH) ccur[r] = round_act(__fmul_rn(a.kv_norm[r], round_act(__fmul_rn(ccur[r], rs), a.act)), a.act);
  }
  for (int j = tid; j < DR / 2; j += NTH) {
    const float c = cr[j], s = sr[j];
    const float k0 = csrc(R + 2 * j), k1 = csrc(R + 2 * j + 1);
    kcur[2 * j] = round_act(rope_lo(k0, k1, c, s), a.act); kcur[2 * j + 1] = round_act(rope_hi(k0, k1, c, s), a.act);
    const float x0 = qsrc(qoff + DN + 2 * j), x1 = qsrc(qoff + DN + 2 * j + 1);
    qp[2 * j] = round_act(rope_lo(x0, x1, c, s), a.act); qp[2 * j + 1] = round_act(rope_hi(x0, x1, c, s), a.act);
  }
  for (int d = tid; d < DN; d += NTH) qn[d] = qsrc(qoff + d);
  __syncthreads();
  if (hd == 0 && blockIdx.y == 0) {
    size_t wo;
    if (a.kv.paged) { const int slot = a.kv.slot ? a.kv.slot[0] : (a.kv.block_table[pos / a.kv.bs] * a.kv.bs + pos % a.kv.bs); wo = rowbase + (size_t)slot * Wd; }
    else wo = rowbase + (size_t)pos * Wd;
    for (int i = tid; i < Wd; i += NTH) kv_st(a.kv.k, wo + i, a.kv.dtype, i < R ? ccur[i] : kcur[i - R]);
  }
  {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // 16-bit q x 16-bit weight: the f32 product is exact, the double carries the sum
    for (int d = d0; d < d1; d += RIF) {
      if (d > d0) {
#pragma unroll
        for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)min(d + u, d1 - 1)) * R + colc, w0[u]);
      }
#pragma unroll
      for (int u = 0; u < RIF; u++) {
        const float qd = (d + u < d1) ? qn[d + u] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += (double)__fmul_rn(qd, w0[u][e]);
      }
    }
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) partd[wave * R + colc + e] = acc[e];
  }
  __syncthreads();
  for (int r = tid; r < R; r += NTH) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) t += partd[w * R + r];
    qabs[r] = round_act((float)t, a.act);
  }
  __syncthreads();
  // ---- scores: wave w takes cached tokens tlo + w, + 16, ...; exact sums (f32 products of 16-bit values are exact) ----
  float qa[8];
  const float qpl = (lane < DR) ? qp[lane] : 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) qa[e] = con ? qabs[colc + e] : 0.f;
  for (int t0 = tlo + wave; t0 < thi; t0 += NW * TIF) {
    float cv[TIF][8], kp[TIF];
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      const size_t ro = rowoff(min(t0 + NW * u, thi - 1));
      ld8t<DT>(a.kv.k, ro + colc, cv[u]);
      kp[u] = ld1t<DT>(a.kv.k, ro + R + min(lane, DR - 1));
    }
#pragma unroll
    for (int u = 0; u < TIF; u++) {
      double dsum = (lane < DR) ? (double)__fmul_rn(qpl, kp[u]) : 0.0;
#pragma unroll
      for (int e = 0; e < 8; e++) dsum += (double)__fmul_rn(qa[e], cv[u][e]);
      dsum = wave_sum_d(dsum);
      if (lane == 0 && t0 + NW * u < thi) sc[t0 + NW * u - tlo] = __fmul_rn((float)dsum, a.scale);
    }
  }
  if (own_cur && wave == 0) {
    double dsum = (lane < DR) ? (double)__fmul_rn(qpl, kcur[lane]) : 0.0;
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) dsum += (double)__fmul_rn(qa[e], ccur[colc + e]);
    dsum = wave_sum_d(dsum);
    if (lane == 0) sc[thi - tlo] = __fmul_rn((float)dsum, a.scale);
  }
  __syncthreads();
  // ---- this slice's scores and their maximum go to the workspace: the weights need the maximum over ALL slices (k_mla_weights_x, after the kernel boundary) ----
  float mx = -INFINITY;
  for (int t = tid; t < nloc; t += NTH) { const float v = sc[t]; mx = fmaxf(mx, v); scw[((size_t)hd * a.nsplit + blockIdx.y) * SCS + t] = v; }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float m2 = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) m2 = fmaxf(m2, red[w]);
    mxw[hd * a.nsplit + blockIdx.y] = m2;
  }
}
// second launch of the three-launch exact form: p_t = exp(s_t - M) with M over all slices, partial weight sum and partial latent sum (double) of this slice
template <int DT>
__global__ __launch_bounds__(1024) void k_mla_weights_x(MlaArgs a, const float* __restrict__ scw, const float* __restrict__ mxw, int SCS, double* __restrict__ wsd) {
  constexpr int NW = 16, NTH = 1024, TIF = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = a.rank, DR = a.rope;
  double* partd = (double*)lds;                 // [NW][R]
  double* redd = partd + NW * R;                // [NW]
  float* sc = (float*)(redd + NW);              // [nloc]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int pos = a.pos[0], nc = pos;
  const int ts = (nc + a.nsplit - 1) / a.nsplit;
  const int tlo = min((int)blockIdx.y * ts, nc), thi = min(tlo + ts, nc);
  const bool own_cur = (int)blockIdx.y == a.nsplit - 1;
  const int nloc = thi - tlo + (own_cur ? 1 : 0);
  const size_t rowbase = (size_t)a.layer * a.kv.layer_stride;
  const int Wd = R + DR;
  auto rowoff = [&](int p) -> size_t {
    if (a.kv.paged) return rowbase + ((size_t)a.kv.block_table[p / a.kv.bs] * a.kv.bs + (p % a.kv.bs)) * Wd;
    return rowbase + (size_t)p * Wd;
  };
  const int col = lane * 8;
  const bool con = col < R;
  const int colc = con ? col : 0;
  float M = -INFINITY;
  for (int s0 = 0; s0 < a.nsplit; s0++) M = fmaxf(M, mxw[hd * a.nsplit + s0]);      // (uniform loads: the same nsplit words for every lane)
  double psum = 0.0;
  for (int t = tid; t < nloc; t += NTH) { const float pe = bz_expf(scw[((size_t)hd * a.nsplit + blockIdx.y) * SCS + t] - M); sc[t] = pe; psum += (double)pe; }
  psum = block_sum_nw_d<NW>(psum, redd);         // (its barriers also publish sc)
  {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int tend = thi + (own_cur ? 1 : 0);     // the current token's row is in the cache (written by the scores launch)
    for (int t0 = tlo + wave; t0 < tend; t0 += NW * TIF) {
      float cv[TIF][8];
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const int pp = min(t0 + NW * u, tend - 1);
        size_t ro;
        if (pp == pos && a.kv.paged) { const int slot = a.kv.slot ? a.kv.slot[0] : (a.kv.block_table[pos / a.kv.bs] * a.kv.bs + pos % a.kv.bs); ro = rowbase + (size_t)slot * Wd; }
        else ro = rowoff(pp);
        ld8t<DT>(a.kv.k, ro + colc, cv[u]);
      }
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const double pw = (t0 + NW * u < tend) ? (double)sc[t0 + NW * u - tlo] : 0.0;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = fma(pw, (double)cv[u][e], acc[e]);
      }
    }
    if (con)
#pragma unroll
      for (int e = 0; e < 8; e++) partd[wave * R + colc + e] = acc[e];
  }
  __syncthreads();
  double* wsp = wsd + ((size_t)hd * a.nsplit + blockIdx.y) * (R + 1);
  for (int r = tid; r < R; r += NTH) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) t += partd[w * R + r];
    wsp[r] = t;
  }
  if (tid == 0) wsp[R] = psum;
}
// merge of the exact partials + Wuv: grid = (n_heads, 4); olat_h = R(f32(sum_s part_s) / f32(sum_s l_s)), out_h = R(f32(Wuv_h . olat_h)); workgroup (h, 0) resets the head's sync words
template <int DT>
__global__ __launch_bounds__(256) void k_mla_merge_x(MlaArgs a, const double* __restrict__ wsd, unsigned* __restrict__ sync) {
  __shared__ float olat[512];
  __shared__ double lsh;
  const int R = a.rank, DN = a.nope, DV = a.vdim, NSP = a.nsplit;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int rows = DV / 4, v0 = blockIdx.y * rows + wave * (rows / 4), v1 = v0 + rows / 4;
  const size_t wrow0 = (size_t)hd * (DN + DV);
  const int col = lane * 8;
  const bool con = col < R;
  const int colc = con ? col : 0;
  constexpr int RIF = 8;
  float w0[RIF][8];
#pragma unroll
  for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(v0 + u, v1 - 1)) * R + colc, w0[u]);
  __builtin_amdgcn_sched_barrier(0);
  const double* wsp = wsd + (size_t)hd * NSP * (R + 1);
  if (wave == 0) {                                          // lane s holds slice s (NSP <= 62)
    const double l = wave_sum_d(lane < NSP ? wsp[(size_t)min(lane, NSP - 1) * (R + 1) + R] : 0.0);
    if (lane == 0) lsh = l;
  }
  __syncthreads();
  const float inv = div_rn(1.0f, (float)lsh);              // the oracle: inv = 1 / f32(sum p), olat = R(f32(sum p c) * inv)
  for (int r = tid; r < R; r += 256) {
    double t = 0.0;
    for (int s0 = 0; s0 < NSP; s0 += 8) {                   // eight slices' loads in flight (clamped index, nothing added beyond the last slice)
      double pv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) pv[j] = wsp[(size_t)min(s0 + j, NSP - 1) * (R + 1) + r];
#pragma unroll
      for (int j = 0; j < 8; j++) t += (s0 + j < NSP) ? pv[j] : 0.0;
    }
    olat[r] = round_act(__fmul_rn((float)t, inv), a.act);
  }
  if (blockIdx.y == 0 && tid == 0) { sync[2 * hd] = 0u; sync[2 * hd + 1] = 0u; }     // (0 = below every ordered float: the next step's atomicMax starts from it)
  __syncthreads();
  float qa[8];
#pragma unroll
  for (int e = 0; e < 8; e++) qa[e] = con ? olat[colc + e] : 0.f;
  for (int d = v0; d < v1; d += RIF) {
    if (d > v0) {
#pragma unroll
      for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)min(d + u, v1 - 1)) * R + colc, w0[u]);
    }
#pragma unroll
    for (int u = 0; u < RIF; u++) {
      double sacc = 0.0;
#pragma unroll
      for (int e = 0; e < 8; e++) sacc += (double)__fmul_rn(w0[u][e], qa[e]);
      sacc = wave_sum_d(sacc);
      if (lane == 0 && d + u < v1) a.out[hd * DV + d + u] = round_act((float)sacc, a.act);
    }
  }
}
size_t bzk_mla_x_smem(const MlaArgs& a, int max_len) {
  const int nsc = (max_len + a.nsplit - 1) / a.nsplit + 1;
  return (size_t)(2 * a.rank + 2 * a.rope + a.nope + 2 + 16 + nsc) * 4 + (size_t)(16 * a.rank + 16) * 8 + 64;
}
bool bzk_mla_x_ok(const MlaArgs& a, int max_len) {
  return a.batch == 0 && a.rank <= 512 && a.rank % 8 == 0 && a.nsplit > 1 && a.nsplit <= 62 && a.nope % 16 == 0 && a.vdim % 16 == 0 &&
         (a.wdt == BZ_F16 || a.wdt == BZ_BF16) && a.wdt == a.kv.dtype && bzk_mla_x_smem(a, max_len) <= 160 * 1024;
}
// exact decode MLA: wsd = n_heads * nsplit * (rank + 1) doubles, sync = 2 * n_heads zeroed words (+ err word)
int bzk_mla_attn_x(hipStream_t s, const MlaArgs& a, int max_len, double* wsd, unsigned* sync, unsigned* err, float* scw, float* mxw) {
  if (!bzk_mla_x_ok(a, max_len)) BZ_FAIL(BZ_E_UNSUPPORTED, "mla_attn_x: shape not supported by the exact decode kernel");
  const size_t smem = bzk_mla_x_smem(a, max_len);
  const int SCS = (max_len + a.nsplit - 1) / a.nsplit + 1;             // scores per (head, slice) in the workspace
  const size_t smem_w = (size_t)(16 * a.rank + 16) * 8 + (size_t)SCS * 4 + 64;
  const double bytes = (double)a.n_heads * (a.nope + a.vdim) * a.rank * bz_dtype_size(a.wdt);
  // one launch with the slices waiting for each other's maxima (BZ_MLA_X_WAIT=1), or three launches (scores + slice maxima | weights + partial sums | merge): the
  // wait is a device word polled from a CU with loads in flight (13 us on V2-Lite), the extra kernel boundary costs less
  static const bool wait_form = getenv("BZ_MLA_X_WAIT") != nullptr;
#define LAUNCH_MX(DT) do { \
    static bool attr_done = false; \
    if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_mla_attn_x<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                      BZ_HIP(hipFuncSetAttribute((const void*)k_mla_scores_x<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                      BZ_HIP(hipFuncSetAttribute((const void*)k_mla_weights_x<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; } \
    if (wait_form || !scw) BZ_LAUNCH("mla_attn<split,exact>", bytes, (k_mla_attn_x<DT>), dim3(a.n_heads, a.nsplit), dim3(1024), smem, s, a, wsd, sync, err); \
    else { BZ_LAUNCH("mla_scores<exact>", bytes * a.nope / (a.nope + a.vdim), (k_mla_scores_x<DT>), dim3(a.n_heads, a.nsplit), dim3(1024), smem, s, a, scw, mxw, SCS); \
           BZ_LAUNCH("mla_weights<exact>", 0.0, (k_mla_weights_x<DT>), dim3(a.n_heads, a.nsplit), dim3(1024), smem_w, s, a, (const float*)scw, (const float*)mxw, SCS, wsd); } \
    BZ_LAUNCH("mla_merge<exact>", bytes * a.vdim / (a.nope + a.vdim), (k_mla_merge_x<DT>), dim3(a.n_heads, 4), dim3(256), 0, s, a, (const double*)wsd, sync); } while (0)
  if (a.wdt == BZ_F16) LAUNCH_MX(BZ_F16); else LAUNCH_MX(BZ_BF16);
#undef LAUNCH_MX
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// Prompt rows, token-tiled (round 3).  The (head, token) workgroups of k_mla_attn<BATCH> each re-read the head's Wuk / Wuv (128 KB each at V2-Lite widths) and every
// latent row of their context: 4.4 GB of L2 traffic per layer at 512 tokens -- 563 us, L2-bandwidth-bound.  Here a workgroup takes TT consecutive tokens of one head:
// a weight row / a latent row is loaded ONCE and used for all TT tokens (TT x fewer bytes), and every token's sums run in exactly the order k_mla_attn<BATCH> uses
// (wave w: nope rows / positions / v rows w-strided, the same FMA chains, the same wave_sum trees, the same pairing of the four waves' partials), so the output is
// the same bits (tests: BZ_NO_MLA_TILE=1 against the default).    rank <= 512, nope % 16 == 0, v % 16 == 0;  grid = (n_heads, ceil(batch / TT)), 256 threads
// LDS: qn[TT][DN] qp[TT][DR] qabs[TT][R] part[4][TT][R] red[2][TT][4] sc[TT][LS]   (LS = positions the tile's last token sees, rounded up to 4)
template <int DT, int TT>
__global__ __launch_bounds__(256) void k_mla_attn_tile(MlaArgs a, int LS) {
  constexpr int NW = 4, NTH = 256, RIF = 4, TIF = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = a.rank, DN = a.nope, DR = a.rope, DV = a.vdim;
  float* qn = lds; float* qp = qn + TT * DN; float* qabs = qp + TT * DR; float* part = qabs + TT * R; float* red = part + NW * TT * R; float* sc = red + 2 * TT * NW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hd = blockIdx.x;
  const int tok0 = blockIdx.y * TT, ntok = min(TT, a.batch - tok0);
  const int lmax = a.pos0 + tok0 + ntok;                 // the tile's last token attends over positions 0 .. lmax - 1 (its own row is in the cache)
  const int QH = DN + DR, qoff = hd * QH;
  const size_t rowbase = (size_t)a.layer * a.kv.layer_stride;
  const int Wd = R + DR;
  auto rowoff = [&](int p) -> size_t {
    if (a.kv.paged) return rowbase + ((size_t)a.kv.block_table[p / a.kv.bs] * a.kv.bs + (p % a.kv.bs)) * Wd;
    return rowbase + (size_t)p * Wd;
  };
  const size_t wrow0 = (size_t)hd * (DN + DV);
  const int col = lane * 8;
  const bool con = col < R;
  const int colc = con ? col : 0;
  // ---- q of the tile's tokens: q_nope as it is, q_pe roped and rounded ----
  for (int i = tid; i < TT * DN; i += NTH) { const int t = i / DN, d = i % DN; qn[i] = t < ntok ? ((const float*)a.qkv.p)[(size_t)(tok0 + t) * a.q_stride + qoff + d] : 0.f; }
  for (int i = tid; i < TT * (DR / 2); i += NTH) {
    const int t = i / (DR / 2), j = i % (DR / 2);
    if (t < ntok) {
      const int pos = a.pos0 + tok0 + t;
      const float c = a.cos_t[(size_t)pos * (DR / 2) + j], sn = a.sin_t[(size_t)pos * (DR / 2) + j];
      const float* qrow = (const float*)a.qkv.p + (size_t)(tok0 + t) * a.q_stride;
      const float x0 = qrow[qoff + DN + 2 * j], x1 = qrow[qoff + DN + 2 * j + 1];
      qp[t * DR + 2 * j] = round_act(rope_lo(x0, x1, c, sn), a.act); qp[t * DR + 2 * j + 1] = round_act(rope_hi(x0, x1, c, sn), a.act);
    } else { qp[t * DR + 2 * j] = 0.f; qp[t * DR + 2 * j + 1] = 0.f; }
  }
  __syncthreads();
  // ---- qabs[t] = R(Wuk_h^T q_nope[t]): wave w takes nope rows [w DN/4, (w+1) DN/4) ----
  {
    const int d0 = wave * (DN / NW), d1 = d0 + DN / NW;
    float acc[TT][8];
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
      for (int e = 0; e < 8; e++) acc[t][e] = 0.f;
    float w0[RIF][8];
#pragma unroll
    for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)(d0 + u)) * R + colc, w0[u]);
    for (int d = d0; d < d1; d += RIF) {
      float wc[RIF][8];
#pragma unroll
      for (int u = 0; u < RIF; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) wc[u][e] = w0[u][e];
      if (d + RIF < d1) {
#pragma unroll
        for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + (size_t)(d + RIF + u)) * R + colc, w0[u]);
      }
#pragma unroll
      for (int u = 0; u < RIF; u++)
#pragma unroll
        for (int t = 0; t < TT; t++) {
          const float qd = qn[t * DN + d + u];
#pragma unroll
          for (int e = 0; e < 8; e++) acc[t][e] += qd * wc[u][e];
        }
    }
    if (con)
#pragma unroll
      for (int t = 0; t < TT; t++)
#pragma unroll
        for (int e = 0; e < 8; e++) part[(wave * TT + t) * R + colc + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = tid; i < TT * R; i += NTH) {
    const int t = i / R, r = i % R;
    float s2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) s2 += part[(w * TT + t) * R + r] + part[((w + 1) * TT + t) * R + r];
    qabs[i] = round_act(s2, a.act);
  }
  __syncthreads();
  // ---- scores: wave w takes positions w, w + 4, ...; a latent row is loaded once for the TT tokens ----
  {
    float qa[TT][8], qpl[TT];
#pragma unroll
    for (int t = 0; t < TT; t++) {
      qpl[t] = (lane < DR) ? qp[t * DR + lane] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) qa[t][e] = con ? qabs[t * R + colc + e] : 0.f;
    }
    for (int t0 = wave; t0 < lmax; t0 += NW * TIF) {
      float cv[TIF][8], kp[TIF];
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const size_t ro = rowoff(min(t0 + NW * u, lmax - 1));
        ld8t<DT>(a.kv.k, ro + colc, cv[u]);
        kp[u] = ld1t<DT>(a.kv.k, ro + R + min(lane, DR - 1));
      }
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const int pos = t0 + NW * u;
        if (pos < lmax) {
#pragma unroll
          for (int t = 0; t < TT; t++) {
            if (pos <= a.pos0 + tok0 + t && t < ntok) {                  // wave-uniform
              float dsum = (lane < DR) ? qpl[t] * kp[u] : 0.f;
#pragma unroll
              for (int e = 0; e < 8; e++) dsum += qa[t][e] * cv[u][e];
              dsum = wave_sum(dsum);
              if (lane == 0) sc[t * LS + pos] = dsum * a.scale;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- softmax per token (thread-strided maxima / sums, then the four waves' values in block_sum_nw's order) ----
  float mxv[TT], inv[TT];
#pragma unroll
  for (int t = 0; t < TT; t++) {
    const int nloc = t < ntok ? a.pos0 + tok0 + t + 1 : 0;
    float mx = -INFINITY;
    for (int p = tid; p < nloc; p += NTH) mx = fmaxf(mx, sc[t * LS + p]);
    mx = wave_max(mx);
    if (lane == 0) red[t * NW + wave] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TT; t++) {
    float mx = red[t * NW];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = fmaxf(mx, red[t * NW + w]);
    mxv[t] = mx;
  }
#pragma unroll
  for (int t = 0; t < TT; t++) {
    const int nloc = t < ntok ? a.pos0 + tok0 + t + 1 : 0;
    float psum = 0.f;
    for (int p = tid; p < nloc; p += NTH) { const float pe = bz_expf(sc[t * LS + p] - mxv[t]); sc[t * LS + p] = pe; psum += pe; }
    psum = wave_sum(psum);
    if (lane == 0) red[(TT + t) * NW + wave] = psum;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TT; t++) {
    float ps = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) ps += red[(TT + t) * NW + w] + red[(TT + t) * NW + w + 1];
    inv[t] = t < ntok ? div_rn(1.0f, ps) : 0.f;
  }
  // ---- olat[t] = R(sum_p p_t c_p * inv_t) ----
  {
    float acc[TT][8];
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
      for (int e = 0; e < 8; e++) acc[t][e] = 0.f;
    for (int t0 = wave; t0 < lmax; t0 += NW * TIF) {
      float cv[TIF][8];
#pragma unroll
      for (int u = 0; u < TIF; u++) ld8t<DT>(a.kv.k, rowoff(min(t0 + NW * u, lmax - 1)) + colc, cv[u]);
#pragma unroll
      for (int u = 0; u < TIF; u++) {
        const int pos = t0 + NW * u;
#pragma unroll
        for (int t = 0; t < TT; t++) {
          const float pw = (pos < lmax && t < ntok && pos <= a.pos0 + tok0 + t) ? sc[t * LS + pos] : 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) acc[t][e] += pw * cv[u][e];
        }
      }
    }
    if (con)
#pragma unroll
      for (int t = 0; t < TT; t++)
#pragma unroll
        for (int e = 0; e < 8; e++) part[(wave * TT + t) * R + colc + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = tid; i < TT * R; i += NTH) {
    const int t = i / R, r = i % R;
    float s2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) s2 += part[(w * TT + t) * R + r] + part[((w + 1) * TT + t) * R + r];
    float iv = 0.f;
#pragma unroll
    for (int q = 0; q < TT; q++) iv = q == t ? inv[q] : iv;
    qabs[i] = round_act(s2 * iv, a.act);
  }
  __syncthreads();
  // ---- out[t] = R(Wuv_h olat[t]): wave w takes v rows [w DV/4, (w+1) DV/4) ----
  {
    float qa[TT][8];
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
      for (int e = 0; e < 8; e++) qa[t][e] = con ? qabs[t * R + colc + e] : 0.f;
    const int v0 = wave * (DV / NW), v1 = v0 + DV / NW;
    for (int d = v0; d < v1; d += RIF) {
      float wc[RIF][8];
#pragma unroll
      for (int u = 0; u < RIF; u++) ld8t<DT>(a.wkvb, (wrow0 + DN + (size_t)(d + u)) * R + colc, wc[u]);
#pragma unroll
      for (int u = 0; u < RIF; u++)
#pragma unroll
        for (int t = 0; t < TT; t++) {
          float s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) s2 += wc[u][e] * qa[t][e];
          s2 = wave_sum(s2);
          if (lane == 0 && t < ntok) a.out[(size_t)(tok0 + t) * a.out_stride + hd * DV + d + u] = round_act(s2, a.act);
        }
    }
  }
}
static size_t mla_tile_smem(const MlaArgs& a, int TT, int LS) { return (size_t)(TT * (a.nope + a.rope + a.rank) + 4 * TT * a.rank + 2 * TT * 4 + TT * LS) * 4 + 64; }
static int mla_tile_tt(const MlaArgs& a, int max_len) {     // tokens per workgroup: 8, or 4 when the scores of 8 tokens do not fit; 0 = the (head, token) kernel
  static const bool off = getenv("BZ_NO_MLA_TILE") != nullptr;
  static const int env_tt = getenv("BZ_MLA_TILE") ? atoi(getenv("BZ_MLA_TILE")) : 0;
  if (off || a.batch < 2 || a.rank > 512 || a.rank % 8 || a.nope % 16 || a.vdim % 16 || a.qkv.fix) return 0;
  const int LS = (max_len + 3) & ~3;
  if (env_tt == 4 || env_tt == 8) return mla_tile_smem(a, env_tt, LS) <= 160 * 1024 ? env_tt : 0;
  if (mla_tile_smem(a, 4, LS) <= 64 * 1024) return 4;      // 4 tokens per workgroup and two or more workgroups per CU measured faster than 8 and one
  if (mla_tile_smem(a, 8, LS) <= 160 * 1024) return 8;
  if (mla_tile_smem(a, 4, LS) <= 160 * 1024) return 4;
  return 0;
}

static bool mla_split_on(const MlaArgs& a) {
  static const bool off = getenv("BZ_NO_MLA_SPLIT") != nullptr;
  return !off && a.batch == 0 && a.ws != nullptr && a.nsplit > 1 && a.nsplit <= 62 && a.vdim % 16 == 0 && a.nope % 16 == 0;
}
int bzk_mla_nsplit(int n_heads) {
  static const int env = getenv("BZ_MLA_NSPLIT") ? atoi(getenv("BZ_MLA_NSPLIT")) : 0;     // tuning override (1..62)
  if (env > 0) return std::min(env, 62);
  // 8 context slices per head: with the exact decode kernel (k_mla_attn_x: the slices wait for each other's maxima) 8 measured 377 tok/s against 365 with 16 on V2-Lite at
  // context ~580; the f32 kernels, which do not wait, preferred 16 (450 vs 439)
  return std::max(1, std::min(8, 256 / std::max(n_heads, 1)));
}

static int mla_waves(const MlaArgs& a) {
  // decode: one workgroup per head, 16 waves hide each other's dependent chains.  Prompt rows: thousands of (head, token) workgroups -- four waves
  // each, so that a CU runs several of these latency chains at once (16-wave workgroups ran one per CU: 621 us per layer at 512 tokens)
  return (a.batch == 0 && a.nope % 16 == 0 && a.vdim % 16 == 0) ? 16 : 4;
}
size_t bzk_mla_smem(const MlaArgs& a, int max_len) {
  const int nsc = mla_split_on(a) ? (max_len + a.nsplit - 1) / a.nsplit + 1 : max_len;   // scores held by one workgroup
  return (size_t)(a.rank * (2 + mla_waves(a)) + a.rope * 2 + a.nope + 16 + nsc) * 4 + 64;
}

int bzk_mla_attn(hipStream_t s, const MlaArgs& a, int max_len) {
  if (a.rank % 8 || a.rank > 1024 || a.rope > 64 || (a.rope & 1) || a.nope % 4 || a.vdim % 4 || a.kv.n_kv != 1 || a.kv.hd != a.rank + a.rope)
    BZ_FAIL(BZ_E_UNSUPPORTED, "mla_attn: rank %d / rope %d / nope %d / v %d unsupported", a.rank, a.rope, a.nope, a.vdim);
  if (a.wdt != a.kv.dtype) BZ_FAIL(BZ_E_UNSUPPORTED, "mla_attn: kv_b_proj dtype %d must equal the cache dtype %d", a.wdt, a.kv.dtype);
  const double bytes = (double)a.n_heads * (a.nope + a.vdim) * a.rank * bz_dtype_size(a.wdt);
  if (const int TT = a.batch > 0 ? mla_tile_tt(a, max_len) : 0) {     // prompt rows: token tiles
    const int LS = (max_len + 3) & ~3;
    const size_t sm = mla_tile_smem(a, TT, LS);
    const dim3 grid(a.n_heads, (a.batch + TT - 1) / TT);
#define LAUNCH_MT(DT, T_) do { \
      static bool attr_done = false; \
      if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_mla_attn_tile<DT, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; } \
      BZ_LAUNCH("mla_attn<rows,tiled>", bytes, (k_mla_attn_tile<DT, T_>), grid, dim3(256), sm, s, a, LS); } while (0)
#define LAUNCH_MT_T(DT) do { if (TT == 8) LAUNCH_MT(DT, 8); else LAUNCH_MT(DT, 4); } while (0)
    if (a.wdt == BZ_F16) LAUNCH_MT_T(BZ_F16); else if (a.wdt == BZ_BF16) LAUNCH_MT_T(BZ_BF16); else LAUNCH_MT_T(BZ_F32);
#undef LAUNCH_MT_T
#undef LAUNCH_MT
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  const size_t smem = bzk_mla_smem(a, max_len);
  if (smem > 160 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "mla_attn: context %d too long for the single-pass kernel", max_len);
  const int NWV = mla_waves(a);
#define LAUNCH_MLA_WB(NCH, DT, W_, B_) do { \
    static bool attr_done = false; \
    if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_mla_attn<NCH, DT, W_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; } \
    BZ_LAUNCH(B_ ? "mla_attn<rows>" : "mla_attn", bytes, (k_mla_attn<NCH, DT, W_, B_>), dim3(a.n_heads, B_ ? a.batch : 1), dim3(W_ * 64), smem, s, a); } while (0)
#define LAUNCH_MLA_SP(NCH, DT) do { \
    static bool attr_done = false; \
    if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_mla_attn<NCH, DT, 16, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; } \
    BZ_LAUNCH("mla_attn<split>", bytes, (k_mla_attn<NCH, DT, 16, 0, 1>), dim3(a.n_heads, a.nsplit), dim3(1024), smem, s, a); \
    BZ_LAUNCH("mla_merge", bytes * a.vdim / (a.nope + a.vdim), (k_mla_merge<NCH, DT>), dim3(a.n_heads, 4), dim3(256), 0, s, a); } while (0)
#define LAUNCH_MLA_W(NCH, DT, W_) do { if (a.batch > 0) LAUNCH_MLA_WB(NCH, DT, W_, 1); else if (W_ == 16 && mla_split_on(a)) LAUNCH_MLA_SP(NCH, DT); else LAUNCH_MLA_WB(NCH, DT, W_, 0); } while (0)
#define LAUNCH_MLA(NCH, DT) do { if (NWV == 16) LAUNCH_MLA_W(NCH, DT, 16); else LAUNCH_MLA_W(NCH, DT, 4); } while (0)
#define LAUNCH_MLA_DT(DT) do { if (a.rank <= 512) LAUNCH_MLA(1, DT); else LAUNCH_MLA(2, DT); } while (0)
  if (a.wdt == BZ_F16) LAUNCH_MLA_DT(BZ_F16); else if (a.wdt == BZ_BF16) LAUNCH_MLA_DT(BZ_BF16); else LAUNCH_MLA_DT(BZ_F32);
#undef LAUNCH_MLA_DT
#undef LAUNCH_MLA
#undef LAUNCH_MLA_W
#undef LAUNCH_MLA_SP
#undef LAUNCH_MLA_WB
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// one wave: f32 softmax over E logits (the oracle's sequential sum) + greedy top-k (ties -> lowest index); lane l owns experts l, l + 64, ... (E <= 1024).
// Shared by the decode router and the prompt-row router, so a token is routed identically on both paths.
// NJM = slots per lane the code is unrolled for (1: E <= 64, 4: E <= 256, 16: E <= 1024) -- the routing runs ONCE per workgroup, from a cold instruction
// cache: the 16-slot form is ~5000 instructions and took 6-8 us for DeepSeek-V2-Lite's 64 experts, the 1-slot form a few hundred
template <int NJM>
__device__ __forceinline__ void moe_softmax_topk_n(const float* lg, int E, int top_k, int n_shared, float routed_scale, int norm_topk, int* sel, float* wsel, int lane) {
    // wave-parallel softmax + greedy top-k; lane l owns experts l, l + 64, ... (E <= 1024): NJ = ceil(E / 64) live slots per lane
    const int NJ = (E + 63) >> 6;
    float v[NJM];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJM; j++) { v[j] = -INFINITY; if (j < NJ) { const int ee = lane + 64 * j; v[j] = ee < E ? lg[ee] : -INFINITY; m = fmaxf(m, v[j]); } }
    m = wave_max(m);
#pragma unroll
    for (int j = 0; j < NJM; j++) if (j < NJ) { const int ee = lane + 64 * j; v[j] = ee < E ? bz_expf(v[j] - m) : 0.f; }
    float sum = 0.f;                                   // the oracle's sequential order e = 0 .. E-1 (lanes beyond E hold 0: adding them changes nothing)
#pragma unroll
    for (int j = 0; j < NJM; j++)
      if (j < NJ) {
#pragma unroll
        for (int l = 0; l < 64; l++) sum += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v[j]), l));
      }
#pragma unroll
    for (int j = 0; j < NJM; j++) { const int ee = lane + 64 * j; v[j] = (j < NJ && ee < E) ? div_rn(v[j], sum) : -1.f; }
    float tsum = 0.f, myw = 0.f; int mysel = 0;
    for (int k = 0; k < top_k; k++) {
      float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < NJM; j++) if (j < NJ) { const int ee = lane + 64 * j; if (v[j] > bv) { bv = v[j]; bi = ee; } }   // ascending e within a lane: first max wins
      // wave argmax (larger value, then smaller index) without the LDS permute network: every step is symmetric, so all lanes agree
#define ROUTER_STEP(OV, OI) do { const float ov_ = (OV); const int oi_ = (OI); if (ov_ > bv || (ov_ == bv && oi_ < bi)) { bv = ov_; bi = oi_; } } while (0)
      ROUTER_STEP(dpp_get<DPP_XOR1>(bv), dpp_get<DPP_XOR1>(bi));
      ROUTER_STEP(dpp_get<DPP_XOR2>(bv), dpp_get<DPP_XOR2>(bi));
      ROUTER_STEP(dpp_get<DPP_HMIRROR>(bv), dpp_get<DPP_HMIRROR>(bi));
      ROUTER_STEP(dpp_get<DPP_MIRROR>(bv), dpp_get<DPP_MIRROR>(bi));
      {
        const bz_u2_t rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        const bz_u2_t ri = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
        const float v0 = __uint_as_float(rv.x), v1 = __uint_as_float(rv.y); const int i0 = (int)ri.x, i1 = (int)ri.y;
        const bool first = v0 > v1 || (v0 == v1 && i0 < i1);
        bv = first ? v0 : v1; bi = first ? i0 : i1;
      }
      {
        const bz_u2_t rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        const bz_u2_t ri = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
        const float v0 = __uint_as_float(rv.x), v1 = __uint_as_float(rv.y); const int i0 = (int)ri.x, i1 = (int)ri.y;
        const bool first = v0 > v1 || (v0 == v1 && i0 < i1);
        bv = first ? v0 : v1; bi = first ? i0 : i1;
      }
#undef ROUTER_STEP
      if (lane == k) { mysel = bi; myw = bv; }
      tsum += bv;
#pragma unroll
      for (int j = 0; j < NJM; j++) if (j < NJ && lane + 64 * j == bi) v[j] = -1.f;
    }
    if (lane < top_k) { sel[lane] = mysel; wsel[lane] = norm_topk ? div_rn(myw, tsum + 1e-20f) * routed_scale : myw * routed_scale; }
    if (lane < n_shared) { sel[top_k + lane] = E + lane; wsel[top_k + lane] = 1.0f; }
}
__device__ void moe_softmax_topk(const float* lg, int E, int top_k, int n_shared, float routed_scale, int norm_topk, int* sel, float* wsel, int lane) {
  if (E <= 64) moe_softmax_topk_n<1>(lg, E, top_k, n_shared, routed_scale, norm_topk, sel, wsel, lane);
  else if (E <= 256) moe_softmax_topk_n<4>(lg, E, top_k, n_shared, routed_scale, norm_topk, sel, wsel, lane);
  else moe_softmax_topk_n<16>(lg, E, top_k, n_shared, routed_scale, norm_topk, sel, wsel, lane);
}

// Router: one workgroup.  Residual add + RMSNorm (writes h' and the normalised x for the expert GEMVs), f32 logits over E
// experts, f32 softmax, greedy top-k (ties -> lowest index).  Slots [top_k, top_k + n_shared) are the shared-expert halves.
template <int WDT>
__global__ __launch_bounds__(256) void k_moe_router(Pro pro, const void* wr, int E, int top_k, int n_shared, float routed_scale, int norm_topk,
                                                    float* xn_out, int* sel, float* wsel, float* lg_glob, unsigned* counter) {
  // grid = ceil(E / 4): workgroup b computes the logits of experts 4b .. 4b+3 (one per wave, every chunk load in flight at once);
  // the last workgroup to finish (device-scope counter) runs the softmax / top-k.  Block 0 also writes h' and the normalised x.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = pro.H;
  float* xs = lds; float* red = xs + H; float* lg = red + 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int e = min((int)blockIdx.x * 4 + wave, E - 1);
  float w[8][8];                                 // hidden <= 4096 in one trip; longer rows loop
#pragma unroll
  for (int j = 0; j < 8; j++) { const int k = j * 512 + lane * 8; load8<WDT>(wr, (size_t)e * H + (k < H ? k : 0), true, w[j]); }
  __builtin_amdgcn_sched_barrier(0);
  build_x_simple<false>(pro, 0, H, xs, red, blockIdx.x == 0);
  if (blockIdx.x == 0) for (int i = tid; i < H; i += 256) xn_out[i] = xs[i];
  float acc = 0.f;
  for (int k0 = 0; k0 < H; k0 += 4096) {
    if (k0 > 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) { const int k = k0 + j * 512 + lane * 8; load8<WDT>(wr, (size_t)e * H + (k < H ? k : 0), true, w[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = k0 + j * 512 + lane * 8;
      if (k < H) {
        const float4 xa = *(const float4*)(xs + k), xb = *(const float4*)(xs + k + 4);
        acc += w[j][0] * xa.x + w[j][1] * xa.y + w[j][2] * xa.z + w[j][3] * xa.w + w[j][4] * xb.x + w[j][5] * xb.y + w[j][6] * xb.z + w[j][7] * xb.w;
      }
    }
  }
  acc = wave_sum(acc);
  // hand-off to the last workgroup without __threadfence() (an L2 write-back per block): the logits are published with device-scope atomic
  // stores, vmcnt(0) says they have been performed, then the counter; the reader uses device-scope loads (scripts/overlap_probe.hip checks
  // exactly this pattern)
  if (lane == 0 && (int)blockIdx.x * 4 + wave < E)
    __hip_atomic_store(lg_glob + blockIdx.x * 4 + wave, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ unsigned s_last;
  if (tid == 0) s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < E; i += 256) lg[i] = __hip_atomic_load(lg_glob + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) *counter = 0;                    // ready for the next launch (stream order)
  __syncthreads();
  if (wave == 0) moe_softmax_topk(lg, E, top_k, n_shared, routed_scale, norm_topk, sel, wsel, lane);
}

int bzk_moe_router(hipStream_t s, const Pro& pro, const void* wr, int wdt, int E, int top_k, int n_shared, float routed_scale, int norm_topk,
                   float* xn_out, int* sel, float* wsel, float* lg_glob, unsigned* counter) {
  if (E > 1024 || top_k > E || top_k > 64 || n_shared > 64 || pro.H % 8) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_router: E %d / top_k %d unsupported", E, top_k);
  const size_t smem = (size_t)(pro.H + 16 + E) * 4 + 64;
  if (smem > 64 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_router: hidden %d too large", pro.H);
#define LAUNCH_ROUTER(DT) BZ_LAUNCH("moe_router", (double)E * pro.H * bz_dtype_size(wdt), k_moe_router<DT>, dim3((E + 3) / 4), dim3(256), smem, s, pro, wr, E, top_k, \
                                   n_shared, routed_scale, norm_topk, xn_out, sel, wsel, lg_glob, counter)
  if (wdt == BZ_F16) LAUNCH_ROUTER(BZ_F16); else if (wdt == BZ_BF16) LAUNCH_ROUTER(BZ_BF16); else LAUNCH_ROUTER(BZ_F32);
#undef LAUNCH_ROUTER
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

bool bzk_moe_rows2_ok(int wdt, int K) { return rows2_enabled() && (wdt == BZ_F16 || wdt == BZ_BF16) && K % 8 == 0 && K <= 131072; }

int bzk_moe_gemv(hipStream_t s, const MoeGemvArgs& g, int wdt, int n_slots, const Pro& pro, int act, bool split, double bytes) {
  if (split && g.acc && bzk_moe_rows2_ok(wdt, g.K) && (pro.mode == PRO_PLAIN || pro.mode == PRO_SILU || (pro.mode == PRO_NORM && g.route.lg))) {
    // the balanced role kernel over all slots at once (fixed-point accumulators: slot s -> acc[min(s, acc_slots - 1)])
    const int KC = (g.K + 511) / 512;
    const long long U = (long long)((g.N + 3) / 4) * KC * n_slots;
    const int nb = (int)std::max<long long>(std::min<long long>(256, U), (long long)KC * n_slots);
    const MoeSlots ms{g.sel, g.expert_stride, n_slots, g.src_stride, g.acc_stride, g.acc_slots};
    if (pro.mode == PRO_NORM && (g.route.E > 1024 || g.route.top_k + g.route.n_shared > 128 || pro.H != g.K || g.src_stride != 0)) BZ_FAIL(BZ_E_UNSUPPORTED, "moe gate/up with in-launch routing: E %d unsupported", g.route.E);
    const char* lbl = pro.mode == PRO_SILU ? "moe_rows2<down>" : (pro.mode == PRO_NORM ? "moe_rows2<route+gate_up>" : "moe_rows2<gate_up>");
#define LAUNCH_MR2(DT, MODE, FIX, RT) BZ_LAUNCH(lbl, bytes, (k_gemv_rows2<DT, MODE, FIX, RT, true>), dim3(nb), dim3(RT ? 832 : 768), 0, s, g.w, (const float*)nullptr, g.N, g.K, pro, g.acc, \
    (long long*)nullptr, 0, ConvShift{}, ms, g.route)
#define LAUNCH_MR2_F(DT, MODE, RT) do { if (pro.src.fix) LAUNCH_MR2(DT, MODE, true, RT); else LAUNCH_MR2(DT, MODE, false, RT); } while (0)
#define LAUNCH_MR2_M(DT) do { if (pro.mode == PRO_SILU) LAUNCH_MR2_F(DT, PRO_SILU, 0); else if (pro.mode == PRO_NORM) LAUNCH_MR2_F(DT, PRO_NORM, 1); else LAUNCH_MR2_F(DT, PRO_PLAIN, 0); } while (0)
    if (wdt == BZ_F16) LAUNCH_MR2_M(BZ_F16); else LAUNCH_MR2_M(BZ_BF16);
#undef LAUNCH_MR2_M
#undef LAUNCH_MR2_F
#undef LAUNCH_MR2
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  const int KP = (g.K + 511) & ~511;
  const size_t smem = (size_t)KP * 4 + 64;
  if (g.K % 8 || smem > 160 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_gemv: K=%d unsupported", g.K);
  const dim3 grid((g.N + 15) / 16, n_slots);
  const char* lbl = split ? "moe_gemv<down>" : "moe_gemv<gate_up>";
#define LAUNCH_MOE(DT) do { if (split) BZ_LAUNCH(lbl, bytes, (k_moe_rows<DT, true>), grid, dim3(256), smem, s, g, pro, act); \
                            else BZ_LAUNCH(lbl, bytes, (k_moe_rows<DT, false>), grid, dim3(256), smem, s, g, pro, act); } while (0)
  if (wdt == BZ_F16) LAUNCH_MOE(BZ_F16); else if (wdt == BZ_BF16) LAUNCH_MOE(BZ_BF16); else LAUNCH_MOE(BZ_F32);
#undef LAUNCH_MOE
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// routed = R(sum_k w_k * R(y_k)) (selection order, f32) ; out = R(routed + R(y_shared)) ; the accumulators are zeroed for the next layer
__global__ void k_moe_combine(long long* acc, const float* wsel, int top_k, int has_shared, int H, int act, float* out, long long* zero_buf, int zero_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (zero_buf) for (int j = i; j < zero_n; j += gridDim.x * blockDim.x) zero_buf[j] = 0;   // the gate / up accumulators of this layer (read by the down launch)
  if (i >= H) return;
  float r = 0.f;
  for (int k = 0; k < top_k; k++) {
    r = __fadd_rn(r, __fmul_rn(wsel[k], round_act(fix2f(acc[(size_t)k * H + i], act), act)));     // the oracle's f32 chain: product rounded, then added (no contraction)
    acc[(size_t)k * H + i] = 0;
  }
  r = round_act(r, act);
  if (has_shared) {
    r = round_act(r + round_act(fix2f(acc[(size_t)top_k * H + i], act), act), act);
    acc[(size_t)top_k * H + i] = 0;
  }
  out[i] = r;
}
int bzk_moe_combine(hipStream_t s, long long* acc, const float* wsel, int top_k, int has_shared, int H, int act, float* out, long long* zero_buf, int zero_n) {
  BZ_LAUNCH("moe_combine", 0.0, k_moe_combine, dim3((H + 255) / 256), dim3(256), 0, s, acc, wsel, top_k, has_shared, H, act, out, zero_buf, zero_n);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DeepSeek-V2 prompt rows (batched prefill): latent append, routing, per-expert row lists, gather, combine
// ---------------------------------------------------------------------------------------------------------
// row s: c = R(w R(kva[:rank] rs)), k_pe = R(rope(kva[rank:])) -> cache row pos0 + s (the decode kernel's arithmetic, row-wise).  grid = S
__global__ __launch_bounds__(256) void k_mla_append_rows(const float* __restrict__ kva, long long stride, const float* __restrict__ kv_norm, float eps, int R, int DR,
                                                         const float* __restrict__ cos_t, const float* __restrict__ sin_t, int pos0, int act, KvView kv, int layer) {
  __shared__ float red[4];
  const int s = blockIdx.x, tid = threadIdx.x, pos = pos0 + s;
  const float* row = kva + (size_t)s * stride;
  float ss = 0.f;
  for (int r = tid; r < R; r += 256) { const float v = row[r]; ss += v * v; }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float rs = rms_scale(ss, (float)R, eps);
  const int Wd = R + DR;
  size_t wo = (size_t)layer * kv.layer_stride;
  if (kv.paged) wo += ((size_t)kv.block_table[pos / kv.bs] * kv.bs + (pos % kv.bs)) * Wd; else wo += (size_t)pos * Wd;
  for (int r = tid; r < R; r += 256) kv_st(kv.k, wo + r, kv.dtype, round_act(kv_norm[r] * round_act(row[r] * rs, act), act));
  const float* cr = cos_t + (size_t)pos * (DR / 2); const float* sr = sin_t + (size_t)pos * (DR / 2);
  for (int j = tid; j < DR / 2; j += 256) {
    const float c = cr[j], sn = sr[j], x0 = row[R + 2 * j], x1 = row[R + 2 * j + 1];
    kv_st(kv.k, wo + R + 2 * j, kv.dtype, round_act(rope_lo(x0, x1, c, sn), act));
    kv_st(kv.k, wo + R + 2 * j + 1, kv.dtype, round_act(rope_hi(x0, x1, c, sn), act));
  }
}
int bzk_mla_append_rows(hipStream_t s, const float* kva, long long stride, int S, const float* kv_norm, float eps, int rank, int rope, const float* cos_t, const float* sin_t,
                        int pos0, int act, const KvView& kv, int layer) {
  hipLaunchKernelGGL(k_mla_append_rows, dim3(S), dim3(256), 0, s, kva, stride, kv_norm, eps, rank, rope, cos_t, sin_t, pos0, act, kv, layer);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// routing of prompt rows: workgroup = one token; the logits are summed exactly as the decode path sums them (k_gemv_rows2 over the router matrix:
// lane = 8 columns of a 512-column chunk, the 32 / 16 / DPP tree, fixed-point sum over the chunks), so a token gets the same experts and weights
// on both paths.  grid = S, 256 threads.
template <int DT, int WDT>
__global__ __launch_bounds__(256) void k_moe_route_rows(const unsigned short* __restrict__ x16, int H, const void* __restrict__ wr, int E, int top_k, float routed_scale,
                                                        int norm_topk, int* __restrict__ sel, float* __restrict__ wsel) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds; float* lg = xs + H;
  const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < H; i += 256) {
    const unsigned short u = x16[(size_t)t * H + i];
    xs[i] = DT == BZ_F16 ? __half2float(__ushort_as_half(u)) : __uint_as_float((unsigned)u << 16);
  }
  __syncthreads();
  for (int e = wave; e < E; e += 4) {
    long long accf = 0;                     // per 512-k chunk: the exact sum (double), rounded to the fixed-point grid as k_gemv_rows2 rounds its unit partials
    for (int k0 = 0; k0 < H; k0 += 4096) {
      RowPiece<WDT> w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { const int k = k0 + j * 512 + lane * 8; w[j] = piece_load<WDT>(wr, (size_t)e * H + (k < H ? k : 0)); }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = k0 + j * 512 + lane * 8;
        if (k0 + j * 512 < H) {
          const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 xa = k < H ? *(const float4*)(xs + k) : z4, xb = k < H ? *(const float4*)(xs + k + 4) : z4;
          double dj;
          if constexpr (WDT == BZ_F32) { double xd[8]; x8_to_d(xa, xb, xd); dj = piece_dot_d<WDT>(w[j], xd, 0.0); }
          else dj = piece_dot_x<WDT>(w[j], xa, xb, 0.0);
          accf += d2fix(wave_sum_d(dj), DT);      // the chunk's exact sum on the fixed-point grid: k_gemv_rows2's value
        }
      }
    }
    if (lane == 0) lg[e] = fix2f(accf, DT);
  }
  __syncthreads();
  if (wave == 0) moe_softmax_topk(lg, E, top_k, 0, routed_scale, norm_topk, sel + (size_t)t * top_k, wsel + (size_t)t * top_k, lane);
}
int bzk_moe_route_rows(hipStream_t s, int dt, const void* x16, int S, int H, const void* wr, int wdt, int E, int top_k, float routed_scale, int norm_topk, int* sel, float* wsel) {
  if (E > 1024 || top_k > E || top_k > 64 || H % 8 || (dt != BZ_F16 && dt != BZ_BF16)) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_route_rows: E %d / top_k %d unsupported", E, top_k);
  const size_t smem = (size_t)(H + E) * 4 + 64;
#define LAUNCH_RR(DT, WDT) hipLaunchKernelGGL((k_moe_route_rows<DT, WDT>), dim3(S), dim3(256), smem, s, (const unsigned short*)x16, H, wr, E, top_k, routed_scale, norm_topk, sel, wsel)
#define LAUNCH_RR_W(DT) do { if (wdt == BZ_F16) LAUNCH_RR(DT, BZ_F16); else if (wdt == BZ_BF16) LAUNCH_RR(DT, BZ_BF16); else LAUNCH_RR(DT, BZ_F32); } while (0)
  if (dt == BZ_F16) LAUNCH_RR_W(BZ_F16); else LAUNCH_RR_W(BZ_BF16);
#undef LAUNCH_RR_W
#undef LAUNCH_RR
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// per-expert row lists of S x top_k (token, slot) pairs: counts, exclusive offsets, row_of[(t, k)] and tok_of[row].  One workgroup; the order of rows
// inside an expert's list depends on the atomics, the results do not (rows are independent and scattered back by (t, k)).
__global__ __launch_bounds__(1024) void k_moe_plan_rows(const int* __restrict__ sel, int n, int E, int* __restrict__ counts, int* __restrict__ offsets, int* __restrict__ row_of,
                                                        int* __restrict__ tok_of, int top_k) {
  __shared__ int cnt[1024], off[1024], fill[1024];
  const int tid = threadIdx.x;
  if (tid < E) { cnt[tid] = 0; fill[tid] = 0; }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&cnt[sel[i]], 1);
  __syncthreads();
  if (tid == 0) { int a = 0; for (int e = 0; e < E; e++) { off[e] = a; a += cnt[e]; } }
  __syncthreads();
  if (tid < E) { counts[tid] = cnt[tid]; offsets[tid] = off[tid]; }
  for (int i = tid; i < n; i += 1024) {
    const int e = sel[i];
    const int r = off[e] + atomicAdd(&fill[e], 1);
    row_of[i] = r; tok_of[r] = i / top_k;
  }
}
int bzk_moe_plan_rows(hipStream_t s, const int* sel, int S, int top_k, int E, int* counts, int* offsets, int* row_of, int* tok_of) {
  if (E > 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_plan_rows: %d experts", E);
  hipLaunchKernelGGL(k_moe_plan_rows, dim3(1), dim3(1024), 0, s, sel, S * top_k, E, counts, offsets, row_of, tok_of, top_k);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
__global__ void k_moe_gather_rows(const uint4* __restrict__ x16, const int* __restrict__ tok_of, int H8, uint4* __restrict__ xg16) {
  const int r = blockIdx.x;
  const uint4* src = x16 + (size_t)tok_of[r] * H8;
  for (int i = threadIdx.x; i < H8; i += blockDim.x) xg16[(size_t)r * H8 + i] = src[i];
}
int bzk_moe_gather_rows(hipStream_t s, const void* x16, const int* tok_of, int rows, int H, void* xg16) {
  if (H % 8) BZ_FAIL(BZ_E_UNSUPPORTED, "moe_gather_rows: hidden %d", H);
  hipLaunchKernelGGL(k_moe_gather_rows, dim3(rows), dim3(256), 0, s, (const uint4*)x16, tok_of, H / 8, (uint4*)xg16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
// out[t][i] = R(R(sum_k w[t][k] * ye[row_of[t][k]][i]) + R(ysh[t][i]))   -- k_moe_combine's expression, row-wise; ysh: unrounded sum of the shared slots (nullptr: none)
__global__ void k_moe_combine_rows(const float* __restrict__ ye, const int* __restrict__ row_of, const float* __restrict__ wsel, const float* __restrict__ ysh, int top_k, int H,
                                   int act, float* __restrict__ out) {
  const int t = blockIdx.x;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float r = 0.f;
    for (int k = 0; k < top_k; k++) r = __fadd_rn(r, __fmul_rn(wsel[(size_t)t * top_k + k], ye[(size_t)row_of[(size_t)t * top_k + k] * H + i]));
    r = round_act(r, act);
    if (ysh) r = round_act(r + round_act(ysh[(size_t)t * H + i], act), act);
    out[(size_t)t * H + i] = r;
  }
}
int bzk_moe_combine_rows(hipStream_t s, const float* ye, const int* row_of, const float* wsel, const float* ysh, int S, int top_k, int H, int act, float* out) {
  hipLaunchKernelGGL(k_moe_combine_rows, dim3(S), dim3(256), 0, s, ye, row_of, wsel, ysh, top_k, H, act, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
