// bz_sample.hip -- non-greedy half of SamplingOps::logits_to_token (/root/reference/src/engine/sampling.rs:445-460):
// penalties -> temperature -> softmax -> top-k -> top-p -> min-p -> seeded draw, token stays on the device.
//
// The reference's kernel (and its RNG) is in the absent boostr crate; only the argument list is visible.  This build fixes
// the semantics in oracle/orc_ops.c (orc_logits_to_token) and implements exactly those:
//   l_i = penalised logit / temperature ; p_i = (float)(bz_expf(l_i - max) / sum_double)
//   candidates sorted by p descending, ties by ascending id ; keep = top_k ; cut after the first prefix with mass >= top_p ;
//   cut at the first p_i < min_p * p_0 ; u = splitmix64(seed) * 2^-53 * mass(kept) ; token = first i with u < cumulative mass.
// The descending sort is rocPRIM's stable radix sort on the f32 bit patterns (p >= 0, so the bit order is the value order; stability on
// ids 0..V-1 gives the ascending-id tie order).  Everything else is hand-written: three streaming passes and one single-workgroup pick.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "bz_internal.h"

#define SFAIL BZ_FAIL
#define SHIP BZ_HIP

namespace {
constexpr int NB = 128;   // blocks of the streaming passes

__device__ __forceinline__ float penalised(const float* logits, long long i, const long long* ids, const int* cnts, int n, float rp, float fp, float pp) {
  float x = logits[i];
  for (int j = 0; j < n; j++) {
    if (ids[j] == i) {
      if (rp != 1.0f) x = (x > 0.f) ? x / rp : x * rp;
      x -= fp * (float)cnts[j] + pp;
    }
  }
  return x;
}

__global__ __launch_bounds__(256) void k_samp_scale(const float* logits, long long V, const long long* ids, const int* cnts, int n, float rp, float fp, float pp,
                                                    float temperature, float* l2, float* pmax) {
  __shared__ float red[4];
  float m = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
    const float x = penalised(logits, i, ids, cnts, n, rp, fp, pp) / temperature;
    l2[i] = x;
    m = fmaxf(m, x);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void k_samp_exp(float* l2, long long V, const float* pmax, double* psum) {
  __shared__ double red[4];
  float m = -INFINITY;
  for (int b = 0; b < NB; b++) m = fmaxf(m, pmax[b]);
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
    const float e = bz_expf(l2[i] - m);
    l2[i] = e;
    s += (double)e;
  }
  for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) psum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_samp_norm(const float* l2, long long V, const double* psum, unsigned* keys, int* vals) {
  double sum = 0.0;
  for (int b = 0; b < NB; b++) sum += psum[b];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
    keys[i] = __float_as_uint((float)((double)l2[i] / sum));
    vals[i] = (int)i;
  }
}

// single workgroup over the sorted candidates
__global__ __launch_bounds__(256) void k_samp_pick(const unsigned* keys, const int* vals, long long V, int top_k, float top_p, float min_p, unsigned long long seed,
                                                   long long* tok_out) {
  __shared__ double part[256];
  __shared__ double off[257];
  __shared__ long long cut;
  const int tid = threadIdx.x;
  long long keep = V;
  if (top_k > 0 && top_k < keep) keep = top_k;
  auto scan = [&](long long n) {   // off[t] = mass of [0, t*chunk) ; off[256] = mass of [0, n)
    const long long chunk = (n + 255) / 256;
    const long long b = tid * chunk, e = min(b + chunk, n);
    double s = 0.0;
    for (long long i = b; i < e; i++) s += (double)__uint_as_float(keys[i]);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { double c = 0.0; for (int t = 0; t < 256; t++) { off[t] = c; c += part[t]; } off[256] = c; }
    __syncthreads();
    return chunk;
  };
  auto first_where = [&](long long n, long long chunk, int mode, double thr) {   // mode 0: cum >= thr ; 1: p < thr (i >= 1) ; 2: thr < cum
    if (tid == 0) cut = n;
    __syncthreads();
    const long long b = tid * chunk, e = min(b + chunk, n);
    double c = off[tid];
    for (long long i = b; i < e; i++) {
      const double p = (double)__uint_as_float(keys[i]);
      c += p;
      const bool hit = mode == 0 ? c >= thr : (mode == 1 ? (i >= 1 && p < thr) : thr < c);
      if (hit) { atomicMin((unsigned long long*)&cut, (unsigned long long)i); break; }
    }
    __syncthreads();
    const long long r = cut;
    __syncthreads();
    return r;
  };
  long long chunk = scan(keep);
  if (top_p > 0.0f && top_p < 1.0f) {
    const long long i = first_where(keep, chunk, 0, (double)top_p);
    if (i + 1 < keep) keep = i + 1;
  }
  if (min_p > 0.0f) {
    const float thr = __uint_as_float(keys[0]) * min_p;
    keep = first_where(keep, chunk, 1, (double)thr);   // `cut` starts at keep: no hit leaves it unchanged
  }
  chunk = scan(keep);
  const double tot = off[256];
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) * tot;
  const long long i = first_where(keep, chunk, 2, u);
  if (tid == 0) tok_out[0] = vals[i < keep ? i : keep - 1];
}

struct SampWs { long long V = 0; float* l2 = nullptr; unsigned* k0 = nullptr; unsigned* k1 = nullptr; int* v0 = nullptr; int* v1 = nullptr; float* pmax = nullptr;
                double* psum = nullptr; void* tmp = nullptr; size_t tmp_bytes = 0; };
}  // namespace

// workspace lives with the device handle (bz_device::samp_ws); grown on demand, freed in bz_device_close
int bzk_sample_free(void* ws_) {
  SampWs* w = (SampWs*)ws_;
  if (!w) return BZ_OK;
  hipFree(w->l2); hipFree(w->k0); hipFree(w->k1); hipFree(w->v0); hipFree(w->v1); hipFree(w->pmax); hipFree(w->psum); hipFree(w->tmp);
  delete w;
  return BZ_OK;
}

int bzk_sample(hipStream_t s, void** ws_, const float* logits, long long V, const long long* ids, const int* cnts, int n, float rp, float fp, float pp,
               float temperature, int top_k, float top_p, float min_p, unsigned long long seed, long long* tok_out) {
  if (!(temperature > 0.0f)) SFAIL(BZ_E_INVALID, "logits_to_token: temperature must be >= 0");
  if (V > 0x7fffffffLL) SFAIL(BZ_E_UNSUPPORTED, "logits_to_token: vocab too large");
  SampWs* w = (SampWs*)*ws_;
  if (!w || w->V < V) {
    SHIP(hipStreamSynchronize(s));
    bzk_sample_free(w);
    w = new SampWs(); *ws_ = w;
    w->V = V;
    SHIP(hipMalloc(&w->l2, (size_t)V * 4)); SHIP(hipMalloc(&w->k0, (size_t)V * 4)); SHIP(hipMalloc(&w->k1, (size_t)V * 4));
    SHIP(hipMalloc(&w->v0, (size_t)V * 4)); SHIP(hipMalloc(&w->v1, (size_t)V * 4));
    SHIP(hipMalloc(&w->pmax, NB * 4)); SHIP(hipMalloc(&w->psum, NB * 8));
    SHIP(rocprim::radix_sort_pairs_desc(nullptr, w->tmp_bytes, w->k0, w->k1, w->v0, w->v1, (size_t)V, 0, 32, s));
    SHIP(hipMalloc(&w->tmp, w->tmp_bytes ? w->tmp_bytes : 16));
  }
  hipLaunchKernelGGL(k_samp_scale, dim3(NB), dim3(256), 0, s, logits, V, ids, cnts, n, rp, fp, pp, temperature, w->l2, w->pmax);
  hipLaunchKernelGGL(k_samp_exp, dim3(NB), dim3(256), 0, s, w->l2, V, w->pmax, w->psum);
  hipLaunchKernelGGL(k_samp_norm, dim3(NB), dim3(256), 0, s, w->l2, V, w->psum, w->k0, w->v0);
  size_t tb = w->tmp_bytes;
  SHIP(rocprim::radix_sort_pairs_desc(w->tmp, tb, w->k0, w->k1, w->v0, w->v1, (size_t)V, 0, 32, s));
  hipLaunchKernelGGL(k_samp_pick, dim3(1), dim3(256), 0, s, w->k1, w->v1, V, top_k, top_p, min_p, seed, tok_out);
  SHIP(hipGetLastError());
  return BZ_OK;
}
