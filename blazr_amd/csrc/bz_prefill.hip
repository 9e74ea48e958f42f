// bz_prefill.hip -- batched prefill for dense f16 / bf16 Llama-family models ("MFMA for prefill only").
//
//   Y[S,N] = R(X[S,K] . W[N,K]^T)  on the matrix cores: v_mfma_f32_32x32x16_{bf16,f16}, f32 accumulate.
//
// Both operands are K-contiguous ("NT"), which is exactly the MFMA fragment order: lane (r = l & 31, h = l >> 5) of a 32x32x16 step owns
// 8 consecutive k of row r of A and of column r of B.  K is a reduction index, so any permutation of k applied to both operands is allowed:
// within a 64-k tile, step s of lane half h takes k = 32 h + 8 s + j.  A lane then reads 64 contiguous bytes of its row per tile (four 16-byte
// loads), lanes (r,0) and (r,1) together one 128-byte line -- fragments come straight from global memory, no LDS and no transposes.
// A wave owns 32 output columns and up to 128 rows (MT = 1..4 accumulator tiles); weights are streamed once per 128-row chunk of the prompt.
// Register double buffering keeps one 64-k tile of loads in flight under the MFMAs of the previous one.
//
// Around the GEMM: row-wise residual + RMSNorm (16-bit output = GEMM input), RoPE + KV append, causal attention over the cache, SiLU*up.
// Every tensor is rounded to the activation dtype at the same op boundaries as the decode path and the oracle (oracle/orc_llama.c).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "bz_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pf_round(float x, int act) {
  if (act == BZ_F16) return __half2float(__float2half_rn(x));
  if (act == BZ_BF16) { unsigned u = __float_as_uint(x); if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u); return __uint_as_float(u & 0xffff0000u); }
  return x;
}
template <int DT> __device__ __forceinline__ unsigned short to16(float x) {
  if (DT == BZ_F16) return __half_as_ushort(__float2half_rn(x));
  unsigned u = __float_as_uint(x); if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16);
}
template <int DT> __device__ __forceinline__ float from16(unsigned short b) {
  if (DT == BZ_F16) return __half2float(__ushort_as_half(b));
  return __uint_as_float((unsigned)b << 16);
}

template <int DT>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, f32x16 c) {
  if constexpr (DT == BZ_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// grid = (ceil(N / 128), ceil(S / (32 MT))); 256 threads = 4 waves, wave w -> columns n0 + 32 w .. +31
template <int DT, int MT>
__global__ __launch_bounds__(256) void k_gemm_nt(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, const float* __restrict__ bias,
                                                 int S, int N, int K, int act, float* __restrict__ Y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 128 + wave * 32, m0 = blockIdx.y * 32 * MT;
  if (n0 >= N) return;
  const unsigned short* wrow = W + (size_t)min(n0 + r, N - 1) * K + 32 * h;
  const unsigned short* xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; t++) xrow[t] = X + (size_t)min(m0 + 32 * t + r, S - 1) * K + 32 * h;
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
  uint4 b0[4], a0[MT][4], b1[4], a1[MT][4];
  auto load = [&](int k0, uint4 (&b)[4], uint4 (&a)[MT][4]) {
#pragma unroll
    for (int s = 0; s < 4; s++) b[s] = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const u32x4*)(wrow + k0) + s));
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int s = 0; s < 4; s++) a[t][s] = *((const uint4*)(xrow[t] + k0) + s);
  };
  auto fma = [&](const uint4 (&b)[4], const uint4 (&a)[MT][4]) {
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int t = 0; t < MT; t++) acc[t] = mfma16<DT>(a[t][s], b[s], acc[t]);
  };
  load(0, b0, a0);
  for (int k0 = 0; k0 < K; k0 += 128) {
    const int k1 = min(k0 + 64, K - 64), k2 = min(k0 + 128, K - 64);   // clamped: redundant reloads at the tail, never a branch around a load
    load(k1, b1, a1);
    fma(b0, a0);
    load(k2, b0, a0);
    if (k0 + 64 < K) fma(b1, a1);
  }
  // C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int n = n0 + r;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int m = m0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < S) Y[(size_t)m * N + n] = pf_round(acc[t][i] + bv, act);
      }
  }
}

// row s: h <- R(h + prev) (prev optional) ; x16 <- to16(R(w * R(h * rs)))        grid = S
template <int DT>
__global__ __launch_bounds__(256) void k_pf_norm(float* hbuf, const float* prev, const float* w, int H, float eps, int act, unsigned short* x16) {
  __shared__ float red[4];
  float* hr = hbuf + (size_t)blockIdx.x * H;
  const float* pr = prev ? prev + (size_t)blockIdx.x * H : nullptr;
  float ss = 0.f;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = hr[i];
    if (pr) { v = pf_round(v + pr[i], act); hr[i] = v; }
    ss += v * v;
  }
  for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float rs = 1.0f / sqrtf(ss / (float)H + eps);
  for (int i = threadIdx.x; i < H; i += 256) x16[(size_t)blockIdx.x * H + i] = to16<DT>(pf_round(w[i] * pf_round(hr[i] * rs, act), act));
}

// row s, head j of [q heads | k heads | v heads]: RoPE on q and k (rounded), k / v appended to the cache at position pos0 + s.   grid = (S, nq + 2 nkv)
__global__ __launch_bounds__(64) void k_pf_rope_kv(float* qkv, int nq, int nkv, int hd, const float* cos_t, const float* sin_t, int interleaved, int pos0, int act,
                                                   KvView kv, int layer, const int* slots) {
  const int s = blockIdx.x, j = blockIdx.y, pos = pos0 + s;
  float* v = qkv + (size_t)s * (nq + 2 * nkv) * hd + (size_t)j * hd;
  const int half = hd / 2;
  if (j < nq + nkv) {
    const float* cr = cos_t + (size_t)pos * half; const float* sr = sin_t + (size_t)pos * half;
    for (int i = threadIdx.x; i < half; i += 64) {
      const int a = interleaved ? 2 * i : i, b = interleaved ? 2 * i + 1 : i + half;
      const float x0 = v[a], x1 = v[b];
      v[a] = pf_round(x0 * cr[i] - x1 * sr[i], act);
      v[b] = pf_round(x1 * cr[i] + x0 * sr[i], act);
    }
  }
  __syncthreads();
  if (j >= nq) {
    const int kvh = (j - nq) % nkv;
    const bool isv = j >= nq + nkv;
    size_t off;
    if (kv.paged) {
      const int slot = slots[s];
      off = (size_t)layer * kv.layer_stride + (((size_t)(slot / kv.bs) * kv.n_kv + kvh) * kv.bs + slot % kv.bs) * kv.hd;
    } else {
      off = (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + pos) * kv.hd;
    }
    void* base = isv ? kv.v : kv.k;
    for (int i = threadIdx.x; i < hd; i += 64) {
      const float x = v[i];
      if (kv.dtype == BZ_F16) ((__half*)base)[off + i] = __float2half_rn(x);
      else if (kv.dtype == BZ_BF16) ((unsigned short*)base)[off + i] = to16<BZ_BF16>(x);
      else ((float*)base)[off + i] = x;
    }
  }
}

__device__ __forceinline__ float kv_at(const KvView& kv, const void* base, int layer, int kvh, int p, int i) {
  size_t off;
  if (kv.paged) { const int blk = kv.block_table[p / kv.bs]; off = (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + p % kv.bs) * kv.hd + i; }
  else off = (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd + i;
  if (kv.dtype == BZ_F16) return __half2float(((const __half*)base)[off]);
  if (kv.dtype == BZ_BF16) return __uint_as_float((unsigned)((const unsigned short*)base)[off] << 16);
  return ((const float*)base)[off];
}

// causal attention of query (s, head) over cache positions [0, pos0 + s]; output rounded and stored as 16-bit (the o_proj GEMM input).
// grid = (S, nq), 256 threads; scores in LDS (len <= max_len)
template <int DT>
__global__ __launch_bounds__(256) void k_pf_attn(const float* qkv, int nq, int nkv, int hd, int pos0, int act, KvView kv, int layer, float scale, unsigned short* out16) {
  extern __shared__ float lds[];
  float* q = lds; float* red = q + hd; float* part = red + 8; float* sc = part + 4 * hd;
  const int s = blockIdx.x, hq = blockIdx.y, kvh = hq / (nq / nkv), len = pos0 + s + 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < hd; i += 256) q[i] = qkv[(size_t)s * (nq + 2 * nkv) * hd + (size_t)hq * hd + i];
  __syncthreads();
  for (int p = tid; p < len; p += 256) {
    float d = 0.f;
    for (int i = 0; i < hd; i++) d += q[i] * kv_at(kv, kv.k, layer, kvh, p, i);
    sc[p] = d * scale;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int p = tid; p < len; p += 256) m = fmaxf(m, sc[p]);
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int p = tid; p < len; p += 256) { const float e = expf(sc[p] - m); sc[p] = e; sum += e; }
  for (int k = 32; k >= 1; k >>= 1) sum += __shfl_xor(sum, k, 64);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  // PV: wave w takes positions w, w+4, ...; lane owns dims lane, lane+64, ...
  for (int i0 = 0; i0 < hd; i0 += 64) {
    const int i = i0 + lane;
    float a = 0.f;
    if (i < hd) for (int p = wave; p < len; p += 4) a += sc[p] * kv_at(kv, kv.v, layer, kvh, p, i);
    if (i < hd) part[wave * hd + i] = a;
  }
  __syncthreads();
  for (int i = tid; i < hd; i += 256)
    out16[(size_t)s * nq * hd + (size_t)hq * hd + i] = to16<DT>(pf_round(((part[i] + part[hd + i]) + (part[2 * hd + i] + part[3 * hd + i])) * inv, act));
}

// a16[s][i] = to16(R(R(silu(g)) * u)),  gu rows = [gate (I) | up (I)]
template <int DT>
__global__ void k_pf_silu(const float* gu, int S, int I, int act, unsigned short* a16) {
  const size_t n = (size_t)S * I;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t s = idx / I, i = idx % I;
    const float g = gu[s * 2 * I + i], u = gu[s * 2 * I + I + i];
    a16[idx] = to16<DT>(pf_round(pf_round(g / (1.0f + expf(-g)), act) * u, act));
  }
}

template <int DT>
__global__ void k_pf_cvt16(const float* x, size_t n, unsigned short* y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = to16<DT>(x[i]);
}
__global__ void k_pf_embed(const void* table, int tdt, const long long* tok, int S, int H, int act, float* out) {
  const int s = blockIdx.x;
  const size_t row = (size_t)tok[s] * H;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float v;
    if (tdt == BZ_F32) v = ((const float*)table)[row + i];
    else if (tdt == BZ_F16) v = __half2float(((const __half*)table)[row + i]);
    else v = __uint_as_float((unsigned)((const unsigned short*)table)[row + i] << 16);
    out[(size_t)s * H + i] = pf_round(v, act);
  }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------------------------------------------------------
int bzk_gemm_nt(hipStream_t s, int dt, const void* x16, const void* w, const float* bias, int S, int N, int K, int act, float* y) {
  if (dt != BZ_F16 && dt != BZ_BF16) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: 16-bit operands only");
  if (K % 64 || K < 64 || S <= 0 || N <= 0) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: K=%d must be a positive multiple of 64", K);
  const int MT = S > 96 ? 4 : (S > 64 ? 3 : (S > 32 ? 2 : 1));
  const dim3 grid((N + 127) / 128, (S + 32 * MT - 1) / (32 * MT));
  const double flops = 2.0 * S * (double)N * K;
#define LAUNCH_GEMM(DT, M) BZ_LAUNCH("gemm_nt_mfma", flops, (k_gemm_nt<DT, M>), grid, dim3(256), 0, s, (const unsigned short*)x16, (const unsigned short*)w, bias, S, N, K, act, y)
#define LAUNCH_GEMM_M(DT) do { if (MT == 4) LAUNCH_GEMM(DT, 4); else if (MT == 3) LAUNCH_GEMM(DT, 3); else if (MT == 2) LAUNCH_GEMM(DT, 2); else LAUNCH_GEMM(DT, 1); } while (0)
  if (dt == BZ_F16) LAUNCH_GEMM_M(BZ_F16); else LAUNCH_GEMM_M(BZ_BF16);
#undef LAUNCH_GEMM_M
#undef LAUNCH_GEMM
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

int bzk_pf_cvt16(hipStream_t s, int dt, const float* x, size_t n, void* y) {
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_cvt16<BZ_F16>, dim3(256), dim3(256), 0, s, x, n, (unsigned short*)y);
  else hipLaunchKernelGGL(k_pf_cvt16<BZ_BF16>, dim3(256), dim3(256), 0, s, x, n, (unsigned short*)y);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_embed(hipStream_t s, const void* table, int tdt, const long long* tok, int S, int H, int act, float* out) {
  hipLaunchKernelGGL(k_pf_embed, dim3(S), dim3(256), 0, s, table, tdt, tok, S, H, act, out);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_norm(hipStream_t s, int dt, float* hbuf, const float* prev, const float* w, int S, int H, float eps, int act, void* x16) {
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_norm<BZ_F16>, dim3(S), dim3(256), 0, s, hbuf, prev, w, H, eps, act, (unsigned short*)x16);
  else hipLaunchKernelGGL(k_pf_norm<BZ_BF16>, dim3(S), dim3(256), 0, s, hbuf, prev, w, H, eps, act, (unsigned short*)x16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_rope_kv(hipStream_t s, float* qkv, int S, int nq, int nkv, int hd, const float* cos_t, const float* sin_t, int interleaved, int pos0, int act,
                   const KvView& kv, int layer, const int* slots) {
  hipLaunchKernelGGL(k_pf_rope_kv, dim3(S, nq + 2 * nkv), dim3(64), 0, s, qkv, nq, nkv, hd, cos_t, sin_t, interleaved, pos0, act, kv, layer, slots);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_attn(hipStream_t s, int dt, const float* qkv, int S, int nq, int nkv, int hd, int pos0, int act, const KvView& kv, int layer, void* out16) {
  const size_t smem = (size_t)(hd * 5 + 8 + pos0 + S) * 4 + 64;
  if (smem > 64 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "prefill attention: context %d too long for this kernel", pos0 + S);
  const float scale = 1.0f / sqrtf((float)hd);
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_attn<BZ_F16>, dim3(S, nq), dim3(256), smem, s, qkv, nq, nkv, hd, pos0, act, kv, layer, scale, (unsigned short*)out16);
  else hipLaunchKernelGGL(k_pf_attn<BZ_BF16>, dim3(S, nq), dim3(256), smem, s, qkv, nq, nkv, hd, pos0, act, kv, layer, scale, (unsigned short*)out16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_silu(hipStream_t s, int dt, const float* gu, int S, int I, int act, void* a16) {
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_silu<BZ_F16>, dim3(512), dim3(256), 0, s, gu, S, I, act, (unsigned short*)a16);
  else hipLaunchKernelGGL(k_pf_silu<BZ_BF16>, dim3(512), dim3(256), 0, s, gu, S, I, act, (unsigned short*)a16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
