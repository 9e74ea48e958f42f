// bz_prefill.hip -- the prompt phase on the matrix cores ("MFMA for prefill only"): batched prefill of the Llama family (dense 16-bit and
// int4 group-quantised weights), Mamba2 and DeepSeek-V2, and the multi-row steps of batched decode.
//
//   Y[S,N] = R(X[S,K] . W[N,K]^T)  on the matrix cores: v_mfma_f32_32x32x16_{bf16,f16}, f32 accumulate.
//
// Both operands are K-contiguous ("NT"), which is exactly the MFMA fragment order: lane (r = l & 31, h = l >> 5) of a 32x32x16 step owns
// 8 consecutive k of row r of A and of column r of B.  K is a reduction index, so any permutation of k applied to both operands is allowed:
// within a 64-k tile, step s of lane half h takes k = 32 h + 8 s + j -- a lane reads 64 contiguous bytes of its row per tile; no transposes.
//
//   k_gemm_nt2        dense 16-bit GEMM, both operands global -> LDS by LDS-DMA (128 x 128 x 64 / 64 x 128 x 64 tiles, split-K, grouped MoE form)
//   k_gemm_nt         the round-1 form (weights straight to registers, A through LDS), kept for A/B runs (BZ_GEMM_NT_WAVE_TILES)
//   k_gemm_q4g_lds    W4A16: raw int4 chunks by LDS-DMA, exact (q - z) f16 fragments rebuilt per wave, f32 group scales (square and 64 x 256 wide form)
//   k_gemm_q4g_mfma   W4A16 single-wave form (<= 16 rows)
//   k_pf_attn_mfma    flash-style causal attention for prompts;  k_pf_attn: scalar form with scores in LDS (decode batches)
//   k_ssm_scan        Mamba2 recurrence over the prompt's tokens (sequential in t: the state is rounded per token as the decode step stores it)
//
// Around them: row-wise residual + RMSNorm (16-bit output = GEMM input), RoPE + KV append, SiLU*up, depthwise conv, gated norm.
// Every tensor is rounded to the activation dtype at the same op boundaries as the decode path and the oracle (oracle/orc_llama.c).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include "bz_internal.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pf_round(float x, int act) {
  if (act == BZ_F16) return f16_round(x);
  if (act == BZ_BF16) return (float)(__bf16)x;     // v_cvt_pk_bf16_f32 (round to nearest even) + shift
  return x;
}
template <int DT> __device__ __forceinline__ unsigned short to16(float x) {
  if (DT == BZ_F16) return __half_as_ushort(f16_cvt(x));
  const __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b);
}
// one element of a GEMM-input row buffer: 16-bit for the 16-bit activation dtypes, plain f32 for f32-activation models (GGUF), whose rows are split into
// 16-bit pieces by k_pf_split3 afterwards
template <int DT> __device__ __forceinline__ void put_x(void* base, size_t i, float v) {
  if constexpr (DT == BZ_F32) ((float*)base)[i] = v;
  else ((unsigned short*)base)[i] = to16<DT>(v);
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

// Workgroup = 4 waves side by side along N; wave tile = (32 MT) rows x (32 NT) columns.  The A tile (32 MT rows x 64 k) is shared by the four
// waves through LDS (row stride 144 B: 16-byte fragment reads of 32 consecutive rows spread over all banks); B fragments have no reuse across
// waves and go straight from global memory to registers.  Double buffered: the loads of tile i+1 (A -> registers -> LDS, B -> registers) are
// in flight under the MFMAs of tile i; one barrier per tile.
// grid = (ceil(N / (128 NT)), ceil(S / (32 MT)))
constexpr int A_STRIDE = 144;   // bytes per LDS row: 64 k x 2 B + 16 B pad
template <int DT, int MT, int NT>
__global__ __launch_bounds__(256) void k_gemm_nt(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, const float* __restrict__ bias,
                                                 int S, int N, int K, int act, float* __restrict__ Y, const int* __restrict__ g_off, const int* __restrict__ g_cnt,
                                                 long long w_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[2][32 * MT * A_STRIDE];
  if (g_cnt) {   // grouped form (MoE experts over gathered token rows): group blockIdx.z owns rows [g_off, g_off + g_cnt) of X / Y and the z-th weight matrix
    const int z = blockIdx.z, off = g_off[z];
    S = g_cnt[z];
    if ((int)blockIdx.y * 32 * MT >= S) return;
    X += (size_t)off * K; Y += (size_t)off * N; W += (size_t)z * w_stride;
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int n0 = (blockIdx.x * 4 + wave) * 32 * NT, m0 = blockIdx.y * 32 * MT;
  // B: lane (r,h) reads 64 contiguous bytes (k = 32 h .. 32 h + 31) of row n per 64-k tile
  unsigned woff[NT];                      // element offsets (N K < 2^32 for every matrix on this path)
#pragma unroll
  for (int j = 0; j < NT; j++) woff[j] = (unsigned)min(n0 + 32 * j + r, N - 1) * (unsigned)K + 32u * h;
  // A staging: 32 MT rows x 128 B per tile = 8 MT pieces of 16 B per row-pair...: piece p = tid + 256 i covers row p / 8, 16-byte column p % 8
  constexpr int APT = MT;                 // 32 MT rows x 8 pieces = 256 MT pieces: exactly MT per thread; piece i of thread t: row (t >> 3) + 32 i, column t & 7
  unsigned xoff[APT];
#pragma unroll
  for (int i = 0; i < APT; i++) xoff[i] = (unsigned)min(m0 + (tid >> 3) + 32 * i, S - 1) * (unsigned)K + 8u * (tid & 7);
  const int adst0 = (tid >> 3) * A_STRIDE + 16 * (tid & 7);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; t++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][j][i] = 0.f;
  uint4 bcur[NT][4], bnext[NT][4], areg[APT];
#define GEMM_GLOAD(K0, B)                                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NT; j++)                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 4; s++)                                                                                  \
      B[j][s] = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const u32x4*)(W + woff[j] + (K0)) + s));                     \
  _Pragma("unroll") for (int i = 0; i < APT; i++) areg[i] = *(const uint4*)(X + xoff[i] + (K0));
#define GEMM_ASTORE(BUF) _Pragma("unroll") for (int i = 0; i < APT; i++) *(uint4*)(&sA[BUF][adst0 + 32 * i * A_STRIDE]) = areg[i];
  GEMM_GLOAD(0, bcur)
  GEMM_ASTORE(0)
  __syncthreads();
  const int nk = K >> 6;
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    const int kn = min(kt + 1, nk - 1) << 6;             // clamped: one redundant reload at the tail, never a branch around a load
    GEMM_GLOAD(kn, bnext)
    const unsigned char* a0 = &sA[buf][r * A_STRIDE + 64 * h];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      uint4 af[MT];
#pragma unroll
      for (int t = 0; t < MT; t++) af[t] = *(const uint4*)(a0 + 32 * t * A_STRIDE + 16 * s);
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[t][j] = mfma16<DT>(af[t], bcur[j][s], acc[t][j]);
    }
    GEMM_ASTORE(buf ^ 1)                                  // the other buffer was last read before the previous barrier
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int s = 0; s < 4; s++) bcur[j][s] = bnext[j][s];
  }
#undef GEMM_GLOAD
#undef GEMM_ASTORE
  // C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int n = n0 + 32 * j + r;
    if (n < N) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int m = m0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (m < S) Y[(size_t)m * N + n] = pf_round(acc[t][j][i] + bv, act);
        }
    }
  }
}

// LDS-DMA form (every plain GEMM of the prefill paths; scripts/gemm_probe.hip holds the same kernel with a reference check and the timings behind
// these choices).  Workgroup tile 128 x 128 x 64, 2 x 2 waves of 64 x 64 (2 x 2 MFMA 32x32x16 tiles each).  BOTH operands go global -> LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write pass): one wave-instruction fills 8 rows x 128 B = 1 KB, lane-linear, so the LDS rows are
// un-padded and the 16-byte pieces of a row are XOR-swizzled by (row & 7) -- on the SOURCE address and on the fragment read (conflict-free
// ds_read_b128).  Two 32 KB buffers: tile i+1 is in flight under the MFMAs of tile i; the wait is a counted vmcnt and the barrier a raw s_barrier, so
// nothing drains early; two workgroups share a CU.  Measured against the register-staged wave-tile kernel above (k_gemm_nt): Mamba2 in_proj at 512 rows
// (10576 x 2560) 176 -> 48 us (572 TFLOP/s), out_proj 114 -> 35 us; 2048 x 4096 x 4096 845 TFLOP/s.
// K can be split over KS workgroups (unrounded f32 partials in `part`, summed in a fixed order by k_q4g_mfma_reduce): a few hundred rows give only tens of
// tiles.  XCD-aware tile order: consecutive workgroups of one XCD (blockIdx.x % 8) walk the row tiles and K splits of ONE column tile, so a weight tile
// crosses HBM -> L2 once.                                grid = 8 ceil(ntiles / 8) * mtiles * KS (1-D); LDS = 64 KB
// One 1 KB LDS-DMA: lane i's 16 bytes at `gsrc` land at LDS byte address ldst + 16 i (ldst wave-uniform).  Inline asm, not __builtin_amdgcn_global_load_lds:
// with the builtin hipcc (ROCm 7.2) put an s_waitcnt vmcnt(0) between the DMAs of tile i+1 and the first fragment read of tile i in these kernels (its
// LDS-DMA alias tracking; the same source in scripts/gemm_probe.hip compiled without it), i.e. one memory round trip per k-step.  An asm DMA is outside its
// bookkeeping: the counted waits in the kernels are the only ones.  M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, const unsigned char* ldst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)ldst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
#define BZ_GLDS16(gsrc, ldst) glds16((gsrc), (ldst))
__device__ __forceinline__ void glds4(const void* gsrc, const unsigned char* ldst) {      // 64 lanes x 4 bytes -> 256 contiguous LDS bytes
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)ldst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// TM = 2: 128-row tiles; TM = 1: 64-row tiles (2 x 2 waves of 32 x 64) for <= 64 rows and for the grouped form, where an expert sees tens of rows
// and a 128-row A tile would spend half of the CU's load slots on clamped duplicates
template <int DT, int TM>
__global__ __launch_bounds__(256) void k_gemm_nt2(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, const float* __restrict__ bias,
                                                  int S, int N, int K, int act, float* __restrict__ Y, float* __restrict__ part, int KS, int mtiles, int ntiles,
                                                  const int* __restrict__ g_off, const int* __restrict__ g_cnt, long long w_stride, const float* __restrict__ rscale) {
  constexpr int BM = 64 * TM, TILE = (BM + 128) * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem2[];   // the ONLY LDS object of this kernel (a second one makes hipcc drain vmcnt per k-step)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt = jj % mtiles, rest = jj / mtiles, ks = rest % KS, nt = (rest / KS) * 8 + xcd;
  if (nt >= ntiles) return;                                  // (whole workgroup: no barrier is skipped by part of it)
  if (g_cnt) {   // grouped form (MoE experts over gathered token rows): group blockIdx.y owns rows [g_off, g_off + g_cnt) of X / Y and the y-th weight matrix
    const int z = blockIdx.y, off = g_off[z];
    S = g_cnt[z];
    if (mt * BM >= S) return;
    X += (size_t)off * K; Y += (size_t)off * N; W += (size_t)z * w_stride;
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = mt * BM, n0 = nt * 128;
  const int nk = K >> 6, k_beg = (int)((long long)ks * nk / KS), k_end = (int)((long long)(ks + 1) * nk / KS), nsteps = k_end - k_beg;
  // staging: wave w fills row groups 4 w .. 4 w + 3 (8 rows x 128 B each) of A and of B; lane -> (row lane / 8, slot lane % 8), the slot holds piece slot ^ row
  const int lrow = lane >> 3, piece = (lane & 7) ^ lrow;
  unsigned xo[2 * TM], wo[4];                                // element offsets (S K and N K < 2^32 on this path); rows past the edge are clamped, their results dropped
#pragma unroll
  for (int i = 0; i < 2 * TM; i++) xo[i] = (unsigned)min(m0 + (wave * 2 * TM + i) * 8 + lrow, S - 1) * (unsigned)K + 8u * piece;
#pragma unroll
  for (int i = 0; i < 4; i++) wo[i] = (unsigned)min(n0 + (wave * 4 + i) * 8 + lrow, N - 1) * (unsigned)K + 8u * piece;
  auto issue = [&](int kt, int buf) {
    unsigned char* base = smem2 + buf * TILE;
#pragma unroll
    for (int i = 0; i < 2 * TM; i++) BZ_GLDS16(X + xo[i] + (size_t)kt * 64, base + (wave * 2 * TM + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; i++) BZ_GLDS16(W + wo[i] + (size_t)kt * 64, base + BM * 128 + (wave * 4 + i) * 1024);
  };
  f32x16 acc[TM][2];
#pragma unroll
  for (int t = 0; t < TM; t++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][j][i] = 0.f;
  if (nsteps > 0) {
    issue(k_beg, 0);
    const int aoff = (wm * 32 * TM + r) * 128, boff = BM * 128 + (wn * 64 + r) * 128, sw = r & 7;
    for (int it = 0; it < nsteps; it++) {
      // this wave's pieces of tile `it` have landed; after the barrier everyone's have, and everyone is done reading the buffer the next issue overwrites
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(min(k_beg + it + 1, k_end - 1), (it + 1) & 1);   // clamped: one redundant reload at the tail, never a branch around a load
      const unsigned char* tb = smem2 + (it & 1) * TILE;
#pragma unroll
      for (int sidx = 0; sidx < 4; sidx++) {
        const int po = ((4 * h + sidx) ^ sw) * 16;           // lane (r, h) takes k = 32 h + 8 sidx .. + 7 of its row for MFMA step sidx (A and B alike)
        uint4 af[TM], bf[2];
#pragma unroll
        for (int t = 0; t < TM; t++) af[t] = *(const uint4*)(tb + aoff + t * 32 * 128 + po);
#pragma unroll
        for (int j = 0; j < 2; j++) bf[j] = *(const uint4*)(tb + boff + j * 32 * 128 + po);
#pragma unroll
        for (int t = 0; t < TM; t++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[t][j] = mfma16<DT>(af[t], bf[j], acc[t][j]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tail's redundant LDS-DMA lands before the LDS allocation is released
  }
  // C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + wn * 64 + 32 * j + r;
    if (n < N) {
      const float bv = (!part && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int t = 0; t < TM; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int m = m0 + wm * 32 * TM + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (m < S) {
            if (part) part[((size_t)ks * S + m) * N + n] = acc[t][j][i];
            else Y[(size_t)m * N + n] = pf_round((rscale ? acc[t][j][i] * rscale[m] : acc[t][j][i]) + bv, act);     // rscale: a power of two per row (split f32 operands)
          }
        }
    }
  }
}

// row s: h <- R(h + prev) (prev optional) ; x16 <- to16(R(w * R(h * rs)))        grid = S
template <int DT>
__global__ __launch_bounds__(256) void k_pf_norm(float* hbuf, const float* prev, const float* w, int H, float eps, int act, void* x16) {
  __shared__ double red[4];
  float* hr = hbuf + (size_t)blockIdx.x * H;
  const float* pr = prev ? prev + (size_t)blockIdx.x * H : nullptr;
  // the oracle's rule (orc_rms_norm): f32 squares summed in double, ONE rounding to f32 -- the decode kernels carry the sum the same way, so a row's 1 / rms is
  // the same bits on both paths (a 1e-7 difference flips f16 roundings of the normalised row)
  double ssd = 0.0;
  for (int i = threadIdx.x; i < H; i += 256) {
    float v = hr[i];
    if (pr) { v = pf_round(v + pr[i], act); hr[i] = v; }
    ssd += (double)(v * v);
  }
  ssd = wave_sum_d(ssd);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssd;
  __syncthreads();
  const float ss = (float)((red[0] + red[1]) + (red[2] + red[3]));
  const float rs = rms_scale(ss, (float)H, eps);
  for (int i = threadIdx.x; i < H; i += 256) put_x<DT>(x16, (size_t)blockIdx.x * H + i, pf_round(w[i] * pf_round(hr[i] * rs, act), act));
}

// row s, head j of [q heads | k heads | v heads]: RoPE on q and k (rounded), k / v appended to the cache at position pos0 + s.   grid = (S, nq + 2 nkv)
__global__ __launch_bounds__(64) void k_pf_rope_kv(float* qkv, int nq, int nkv, int hd, const float* cos_t, const float* sin_t, int interleaved, int pos0, int act,
                                                   KvView kv, int layer, const int* slots, const int* row_pos) {
  const int s = blockIdx.x, j = blockIdx.y, pos = row_pos ? row_pos[s] : pos0 + s;   // row_pos: decode batch (one sequence per row)
  float* v = qkv + (size_t)s * (nq + 2 * nkv) * hd + (size_t)j * hd;
  const int half = hd / 2;
  if (j < nq + nkv) {
    const float* cr = cos_t + (size_t)pos * half; const float* sr = sin_t + (size_t)pos * half;
    for (int i = threadIdx.x; i < half; i += 64) {
      const int a = interleaved ? 2 * i : i, b = interleaved ? 2 * i + 1 : i + half;
      const float x0 = v[a], x1 = v[b];
      v[a] = pf_round(rope_lo(x0, x1, cr[i], sr[i]), act);
      v[b] = pf_round(rope_hi(x0, x1, cr[i], sr[i]), act);
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
      if (kv.dtype == BZ_F16) ((__half*)base)[off + i] = f16_cvt(x);
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

// Causal attention for prefill: one workgroup per (query s, kv head) serves all REP = nq / nkv query heads of the group, so every K / V row
// is loaded once per group (GQA).  PR = hd / 8 lanes share a cache row (16 bytes each, a row is one contiguous 16 * PR-byte read); 256 / PR
// rows per pass.  Scores for the REP heads sit in LDS; output rounded and stored as 16-bit (the o_proj GEMM input).
// grid = (S, nkv), 256 threads.  KVDT: cache dtype (compile time: no dtype switch around the loads).
template <int DT, int KVDT>
__device__ __forceinline__ void ld_row8(const void* base, size_t off, float (&o)[8]) {
  if constexpr (KVDT == BZ_F32) {
    const float4 a = *(const float4*)((const float*)base + off), b = *(const float4*)((const float*)base + off + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    const uint4 rr = *(const uint4*)((const unsigned short*)base + off);
    const unsigned u[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { o[2 * i] = from16<KVDT>((unsigned short)(u[i] & 0xffffu)); o[2 * i + 1] = from16<KVDT>((unsigned short)(u[i] >> 16)); }
  }
}
// EXACT: the oracle's arithmetic (orc_attn_decode), which is also the decode kernels' (k_attn2): score = f32(sum_double q k) * scale, p = exp(score - max),
// l = f32(sum_double p), out = f32(sum_double p v) / l (IEEE division).  Products of 16-bit values (and of an f32 weight with one) are exact in double and the
// sums are order-independent to ~1e-16, so a prompt row's attention output is the same bits as the decode step's.  (Non-exact form: f32 FMAs.)
template <int DT, int KVDT, int REP, bool EXACT>
__global__ __launch_bounds__(256) void k_pf_attn(const float* qkv, int nq, int nkv, int hd, int pos0, int act, KvView kv, int layer, float scale, void* out16,
                                                const int* row_pos, int table_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef typename std::conditional<EXACT, double, float>::type acc_t;
  const int s = blockIdx.x, kvh = blockIdx.y, len = (row_pos ? row_pos[s] : pos0 + s) + 1;
  const int* btab = kv.block_table + (row_pos ? (size_t)s * table_stride : 0);   // decode batch: one block-table row per sequence
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int PR = hd >> 3, RPP = 256 / PR;              // lanes per row, rows per pass
  const int c = tid % PR, g = tid / PR;                // this thread's 8-dim piece and row slot
  acc_t* red = (acc_t*)lds;                            // [REP][4] maxima (as acc_t) + [REP][4] sums
  acc_t* part = red + 8 * REP;                         // [RPP][REP][hd]
  float* sc = (float*)(part + RPP * REP * hd);         // [REP][len]
  float q[REP][8];
#pragma unroll
  for (int h = 0; h < REP; h++)
#pragma unroll
    for (int e = 0; e < 8; e++) q[h][e] = qkv[(size_t)s * (nq + 2 * nkv) * hd + (size_t)(kvh * REP + h) * hd + 8 * c + e];
  auto row_off = [&](int p) -> size_t {
    if (kv.paged) { const int blk = btab[p / kv.bs]; return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + p % kv.bs) * kv.hd + 8 * c; }
    return (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd + 8 * c;
  };
  // ---- scores ----
  for (int p0 = 0; p0 < len; p0 += 2 * RPP) {
    float k0[8], k1[8];
    const int pa = p0 + g, pb = p0 + RPP + g;
    ld_row8<DT, KVDT>(kv.k, row_off(min(pa, len - 1)), k0);
    ld_row8<DT, KVDT>(kv.k, row_off(min(pb, len - 1)), k1);
#pragma unroll
    for (int h = 0; h < REP; h++) {
      acc_t d0 = 0, d1 = 0;
#pragma unroll
      for (int e = 0; e < 8; e++) { d0 += (acc_t)q[h][e] * (acc_t)k0[e]; d1 += (acc_t)q[h][e] * (acc_t)k1[e]; }
      if constexpr (EXACT) {
        if (PR == 16) { d0 = grp_sum_d<16>(d0); d1 = grp_sum_d<16>(d1); }
        else if (PR == 8) { d0 = grp_sum_d<8>(d0); d1 = grp_sum_d<8>(d1); }
        else for (int m = 1; m < PR; m <<= 1) { d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
      } else {
        if (PR == 16) { d0 = grp_reduce<16, OpAdd>(d0); d1 = grp_reduce<16, OpAdd>(d1); }
        else if (PR == 8) { d0 = grp_reduce<8, OpAdd>(d0); d1 = grp_reduce<8, OpAdd>(d1); }
        else for (int m = 1; m < PR; m <<= 1) { d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
      }
      if (c == 0) { if (pa < len) sc[h * len + pa] = (float)d0 * scale; if (pb < len) sc[h * len + pb] = (float)d1 * scale; }
    }
  }
  __syncthreads();
  // ---- softmax per head ----
#pragma unroll
  for (int h = 0; h < REP; h++) {
    float m = -INFINITY;
    for (int p = tid; p < len; p += 256) m = fmaxf(m, sc[h * len + p]);
    m = wave_max(m);
    if (lane == 0) red[h * 4 + wave] = (acc_t)m;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < REP; h++) {
    const float m = fmaxf(fmaxf((float)red[h * 4], (float)red[h * 4 + 1]), fmaxf((float)red[h * 4 + 2], (float)red[h * 4 + 3]));
    acc_t sum = 0;
    for (int p = tid; p < len; p += 256) { const float e = bz_expf(sc[h * len + p] - m); sc[h * len + p] = e; sum += (acc_t)e; }
    if constexpr (EXACT) sum = wave_sum_d(sum); else sum = wave_sum(sum);
    if (lane == 0) red[4 * REP + h * 4 + wave] = sum;
  }
  __syncthreads();
  // ---- PV: thread (c, g) accumulates its 8 dims over rows g, g + RPP, ... ----
  acc_t acc[REP][8];
#pragma unroll
  for (int h = 0; h < REP; h++)
#pragma unroll
    for (int e = 0; e < 8; e++) acc[h][e] = 0;
  for (int p0 = 0; p0 < len; p0 += 2 * RPP) {
    float v0[8], v1[8];
    const int pa = p0 + g, pb = p0 + RPP + g;
    ld_row8<DT, KVDT>(kv.v, row_off(min(pa, len - 1)), v0);
    ld_row8<DT, KVDT>(kv.v, row_off(min(pb, len - 1)), v1);
#pragma unroll
    for (int h = 0; h < REP; h++) {
      const acc_t w0 = pa < len ? (acc_t)sc[h * len + pa] : (acc_t)0, w1 = pb < len ? (acc_t)sc[h * len + pb] : (acc_t)0;
#pragma unroll
      for (int e = 0; e < 8; e++) acc[h][e] += w0 * (acc_t)v0[e] + w1 * (acc_t)v1[e];
    }
  }
#pragma unroll
  for (int h = 0; h < REP; h++)
#pragma unroll
    for (int e = 0; e < 8; e++) part[(g * REP + h) * hd + 8 * c + e] = acc[h][e];
  __syncthreads();
  for (int i = tid; i < REP * hd; i += 256) {
    const int h = i / hd, d = i % hd;
    acc_t a = 0;
    for (int gg = 0; gg < RPP; gg++) a += part[(gg * REP + h) * hd + d];
    const acc_t lsum = (red[4 * REP + h * 4] + red[4 * REP + h * 4 + 1]) + (red[4 * REP + h * 4 + 2] + red[4 * REP + h * 4 + 3]);
    float o;
    if constexpr (EXACT) o = div_rn((float)a, (float)lsum);
    else o = (float)a * (1.0f / (float)lsum);
    put_x<DT>(out16, (size_t)s * nq * hd + (size_t)(kvh * REP + h) * hd + d, pf_round(o, act));
  }
}

// Flash-style causal attention for prompts on the matrix cores (row_pos == nullptr; the scalar kernel above keeps the decode batches).
// Workgroup = 4 waves = 4 units (query head of the GQA group, 32-query tile) of ONE kv head: HW = min(REP, 4) heads x QT = 4 / HW query tiles, so a K / V
// tile staged in LDS serves the whole group.  Keys go by in tiles of 64:
//   S^T = K . Q^T    A = K rows from LDS (16-byte reads, pitch 2 HD + 16), B = the wave's Q fragments (registers, rounded by k_pf_rope_kv: exact in 16 bits);
//                    C[key][query]: a lane holds ONE query's 16 keys per 32-key block, so the running max / sum of the online softmax are per-lane
//                    scalars plus one exchange with lane ^ 32, and rescaling O^T is a per-lane multiply;
//   O^T += V^T . P^T A = V^T from LDS (V is transposed while it is staged: lanes of a wave take 64 different keys, so the 2-byte transposed writes of one
//                    d land in 128 contiguous bytes), B = P straight from the score registers (the C layout's key order is used as the k-slot order on
//                    both operands).  P goes to the MFMA as hi + lo 16-bit halves (two MFMAs per step).
// The next tile's K / V rows are requested before the current tile's arithmetic.      grid = (ceil(S / (32 QT)), nkv, REP / HW), 256 threads
template <int DT> __device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = (unsigned)to16<DT>(v[0]) | ((unsigned)to16<DT>(v[1]) << 16); o.y = (unsigned)to16<DT>(v[2]) | ((unsigned)to16<DT>(v[3]) << 16);
  o.z = (unsigned)to16<DT>(v[4]) | ((unsigned)to16<DT>(v[5]) << 16); o.w = (unsigned)to16<DT>(v[6]) | ((unsigned)to16<DT>(v[7]) << 16);
  return o;
}
template <int DT, int HD, int REP, bool PAGED>
__global__ __launch_bounds__(256) void k_pf_attn_mfma(const float* __restrict__ qkv, int S, int nq, int nkv, int pos0, int act, KvView kv, int layer, float scale,
                                                      unsigned short* __restrict__ out16) {
  constexpr int HW = REP < 4 ? REP : 4, QT = 4 / HW, PR = HD / 8, PK = HD * 2 + 16, PV = 64 * 2 + 8, NC = HD / 16, NDB = HD / 32, KPT = PR / 4;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * PK];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[HD * PV];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int kvh = blockIdx.y, hh = wave % HW, qt = wave / HW, head = kvh * REP + blockIdx.z * HW + hh;
  const int wgt = gridDim.x - 1 - blockIdx.x;                 // the longest (latest) query tiles are dispatched first
  const int q0 = (wgt * QT + qt) * 32, qrow = min(q0 + r, S - 1), qpos = pos0 + q0 + r;
  const int kmax = pos0 + min(S, (wgt + 1) * 32 * QT);        // this workgroup's keys: [0, kmax)
  const int my_last = pos0 + min(q0 + 31, S - 1);             // last position any query of this wave attends to
  uint4 qf[NC];
  {
    const float* qp = qkv + (size_t)qrow * (nq + 2 * nkv) * HD + (size_t)head * HD + 8 * h;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const float4 a = *(const float4*)(qp + 16 * c), b = *(const float4*)(qp + 16 * c + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      qf[c] = pack8<DT>(v);
    }
  }
  auto row_off = [&](int p) -> size_t {     // (PAGED is compile time: a run-time branch around each load made hipcc wait for them one by one and spill the prefetch registers)
    if constexpr (PAGED) { const int blk = kv.block_table[p / kv.bs]; return (size_t)layer * kv.layer_stride + (((size_t)blk * kv.n_kv + kvh) * kv.bs + p % kv.bs) * kv.hd; }
    else return (size_t)layer * kv.layer_stride + ((size_t)kvh * kv.cap + p) * kv.hd;
  };
  // (macros, not lambdas: with the prefetch registers captured by reference hipcc kept one of the two arrays in scratch, reloaded and re-stored every tile)
  uint4 kr[KPT], vr[KPT];
#define PFA_GLOAD(KT0)                                                                                                                        \
  _Pragma("unroll") for (int i = 0; i < KPT; i++) {                                                                                           \
    const int idx = tid + 256 * i, row = idx / PR, col = idx % PR;                                 /* K: a row's pieces on consecutive lanes */ \
    kr[i] = *(const uint4*)((const unsigned short*)kv.k + row_off(min((KT0) + row, kmax - 1)) + 8 * col);                                      \
    vr[i] = *(const uint4*)((const unsigned short*)kv.v + row_off(min((KT0) + lane, kmax - 1)) + 8 * (wave + 4 * i));   /* V: 64 keys on the 64 lanes, piece wave + 4 i */ \
  }
  f32x16 o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; d++)
#pragma unroll
    for (int i = 0; i < 16; i++) o[d][i] = 0.f;
  float m = -INFINITY, l = 0.f;
  PFA_GLOAD(0)
  for (int kt0 = 0; kt0 < kmax; kt0 += 64) {
    __syncthreads();                                          // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < KPT; i++) {
      const int idx = tid + 256 * i, row = idx / PR, col = idx % PR;
      *(uint4*)(Ks + row * PK + col * 16) = kr[i];
      unsigned char* vb = Vt + (8 * (wave + 4 * i)) * PV + lane * 2;
      const uint4 vv = vr[i];
      *(unsigned short*)(vb + 0 * PV) = (unsigned short)(vv.x & 0xffffu); *(unsigned short*)(vb + 1 * PV) = (unsigned short)(vv.x >> 16);
      *(unsigned short*)(vb + 2 * PV) = (unsigned short)(vv.y & 0xffffu); *(unsigned short*)(vb + 3 * PV) = (unsigned short)(vv.y >> 16);
      *(unsigned short*)(vb + 4 * PV) = (unsigned short)(vv.z & 0xffffu); *(unsigned short*)(vb + 5 * PV) = (unsigned short)(vv.z >> 16);
      *(unsigned short*)(vb + 6 * PV) = (unsigned short)(vv.w & 0xffffu); *(unsigned short*)(vb + 7 * PV) = (unsigned short)(vv.w >> 16);
    }
    __syncthreads();
    { const int ktn = kt0 + 64 < kmax ? kt0 + 64 : kt0; PFA_GLOAD(ktn) }                  // lands under this tile's arithmetic (last tile: a redundant reload, never a branch around the loads -- under one, hipcc kept the prefetch registers in scratch)
    if (kt0 <= my_last) {                                     // wave-uniform: later tiles are fully masked for this wave
      f32x16 st[2];
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
#pragma unroll
        for (int i = 0; i < 16; i++) st[kb][i] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const uint4 a = *(const uint4*)(Ks + (32 * kb + r) * PK + (16 * c + 8 * h) * 2);
          st[kb] = mfma16<DT>(a, qf[c], st[kb]);
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int kp = kt0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          const float v = kp > qpos ? -INFINITY : st[kb][i] * scale;
          st[kb][i] = v; mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);                          // finite: key 0 is visible to every query and sits in the first tile
      const float alpha = bz_expf(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int i = 0; i < 16; i++) { const float e = bz_expf(st[kb][i] - mn); st[kb][i] = e; ps += e; }
      ps += __shfl_xor(ps, 32, 64);
      l = l * alpha + ps; m = mn;
#pragma unroll
      for (int d = 0; d < NDB; d++)
#pragma unroll
        for (int i = 0; i < 16; i++) o[d][i] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
          // k-slot e of lane half h = key 32 kb + 16 u + 8 (e >> 2) + 4 h + (e & 3): the order the score tile's registers 8 u .. 8 u + 7 already have
          float pv[8], pl[8];
#pragma unroll
          for (int e = 0; e < 8; e++) pv[e] = st[kb][8 * u + e];
          const uint4 phi = pack8<DT>(pv);
          // p = hi + lo in 16 bits each (two MFMAs): P reaches the matrix cores with ~2^-17 (bf16) / 2^-22 (f16) relative error, i.e. the kernel differs from
          // the scalar one only in summation order (single-rounded f16 P measured 1.095e-3 relative L2 on the logits of awq-h2048 against the 1e-3 bar)
#pragma unroll
          for (int e = 0; e < 8; e++) pl[e] = pv[e] - from16<DT>(to16<DT>(pv[e]));
          const uint4 plo = pack8<DT>(pl);
#pragma unroll
          for (int d = 0; d < NDB; d++) {
            const unsigned char* vp = Vt + (32 * d + r) * PV + (32 * kb + 16 * u + 4 * h) * 2;
            const uint2 a0 = *(const uint2*)vp, a1 = *(const uint2*)(vp + 16);
            const uint4 a = {a0.x, a0.y, a1.x, a1.y};
            o[d] = mfma16<DT>(a, phi, o[d]);
            o[d] = mfma16<DT>(a, plo, o[d]);
          }
        }
    }
  }
  if (q0 + r < S) {
    const float inv = 1.0f / l;
    unsigned short* op = out16 + (size_t)(q0 + r) * nq * HD + (size_t)head * HD;
#pragma unroll
    for (int d = 0; d < NDB; d++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int i0 = 4 * g, dd = 32 * d + 8 * g + 4 * h;
        uint2 w;
        w.x = (unsigned)to16<DT>(pf_round(o[d][i0] * inv, act)) | ((unsigned)to16<DT>(pf_round(o[d][i0 + 1] * inv, act)) << 16);
        w.y = (unsigned)to16<DT>(pf_round(o[d][i0 + 2] * inv, act)) | ((unsigned)to16<DT>(pf_round(o[d][i0 + 3] * inv, act)) << 16);
        *(uint2*)(op + dd) = w;
      }
  }
}

#undef PFA_GLOAD
// a16[s][i] = to16(R(R(silu(g)) * u)),  gu rows = [gate (I) | up (I)]
template <int DT>
__global__ __launch_bounds__(256) void k_pf_silu(const float* __restrict__ gu, int S, int I, int act, void* __restrict__ a16) {   // grid = (ceil(I / 1024), S): 4 columns per thread
  const int s = blockIdx.y, i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= I) return;
  const float* row = gu + (size_t)s * 2 * I;
  const float4 g = *(const float4*)(row + i), u = *(const float4*)(row + I + i);
  const float gv[4] = {g.x, g.y, g.z, g.w}, uv[4] = {u.x, u.y, u.z, u.w};
  float of[4];
#pragma unroll
  for (int e = 0; e < 4; e++) of[e] = pf_round(pf_round(div_rn(gv[e], 1.0f + bz_expf(-gv[e])), act) * uv[e], act);
  if constexpr (DT == BZ_F32) *(float4*)((float*)a16 + (size_t)s * I + i) = make_float4(of[0], of[1], of[2], of[3]);
  else {
    unsigned short o[4];
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = to16<DT>(of[e]);
    uint2 w; w.x = o[0] | ((unsigned)o[1] << 16); w.y = o[2] | ((unsigned)o[3] << 16);
    *(uint2*)((unsigned short*)a16 + (size_t)s * I + i) = w;
  }
}

// ---- f32-activation rows on the 16-bit matrix cores (GGUF-quantised models) ------------------------------------------------------------------------
// A value v is carried as hi + lo, two f16 numbers: hi = f16(v'), lo = f16(v' - hi), v' = v 2^e with a power-of-two e that puts the row's (the matrix's)
// largest magnitude at 2^13 -- every element down to 2^-16 of that maximum keeps 22 significant bits with both pieces NORMAL f16 numbers (nothing relies on
// subnormal operands surviving the MFMA).  x . w = xh wh + xl wh + xh wl (+ xl wl, 2^-22 of the product: dropped) becomes ONE 16-bit GEMM over 3 K:
//   X' row = [ xh | xl 2^11 | xh 2^-11 ],   W' row = [ wh | wh 2^-11 | wl 2^11 ]      (the 2^+-11 keep the small pieces in the normal range; exact)
// and the f32 accumulator is scaled back by 2^-(e_row + e_w) in the GEMM's epilogue (rscale).  Products are exact, sums f32: an error of ~2^-22 per product
// against the f32 arithmetic of the decode kernels (exact integer block sums, f32 scales).
__device__ __forceinline__ void split3(float vs, unsigned short& hi, unsigned short& mid, unsigned short& lo, bool weight) {
  const __half h = f16_cvt(vs);
  const float hf = __half2float(h), l = vs - hf;              // exact (Sterbenz-like: hf is vs rounded to 11 bits)
  hi = __half_as_ushort(h);
  if (!weight) { mid = __half_as_ushort(f16_cvt(l * 2048.0f)); lo = __half_as_ushort(f16_cvt(hf * (1.0f / 2048.0f))); }      // [xh | xl 2^11 | xh 2^-11]
  else { mid = __half_as_ushort(f16_cvt(hf * (1.0f / 2048.0f))); lo = __half_as_ushort(f16_cvt(l * 2048.0f)); }               // [wh | wh 2^-11 | wl 2^11]
}
// power of two that brings `am` into [2^13, 2^14) (1 for a zero / non-finite maximum)
__device__ __forceinline__ float split_scale(float am) {
  const unsigned eb = (__float_as_uint(am) >> 23) & 255u;
  if (eb == 0u || eb == 255u || eb > 240u || eb < 20u) return 1.0f;
  return __uint_as_float((127u + 13u + 127u - eb) << 23);     // 2^(13 - (eb - 127))
}
// x [S][K] f32 -> xs [S][3 K] f16 pieces; rscale[s] = 2^-e_s / wscale (wscale: device scalar, the weight matrix's 2^e_w; nullptr = 1)     grid = S
__global__ __launch_bounds__(256) void k_pf_split3(const float* __restrict__ x, int K, unsigned short* __restrict__ xs, float* __restrict__ rscale, const float* __restrict__ wscale) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * K;
  float am = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) am = fmaxf(am, fabsf(xr[i]));
  am = wave_max(am);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
  __syncthreads();
  am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = split_scale(am);
  unsigned short* o = xs + (size_t)blockIdx.x * 3 * K;
  for (int i = threadIdx.x; i < K; i += 256) {
    unsigned short a, b, c;
    split3(xr[i] * sc, a, b, c, false);
    o[i] = a; o[K + i] = b; o[2 * K + i] = c;
  }
  if (threadIdx.x == 0) rscale[blockIdx.x] = (1.0f / sc) / (wscale ? *wscale : 1.0f);
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


// ---------------------------------------------------------------------------------------------------------------------------------------
// W4A16 GEMM on the matrix cores: Y[S,N] = R(X[S,K] . dequant(W)^T) for the int4 group-quantised layout of the decode kernels
//   W [N/64][K/32][64 lanes][16 B] : lane = column, word j of a chunk = k 8j..8j+7 (byte b: low nibble k 8j+b, high nibble k 8j+4+b, both stored as q ^ 8 = two's-complement q - 8)
// A wave owns 64 rows x 64 columns; no LDS.  Per 32-k chunk it loads ONE 16-byte weight piece per lane and turns it into the four MFMA
// B-fragments (two 32-column halves x two 16-k steps) with two V_PERMLANE32_SWAPs: swap(word0, word1) leaves {columns 0-31: k 0-7 | k 8-15}
// in the first result and {columns 32-63: k 0-7 | k 8-15} in the second -- exactly the (column = lane & 31, k-half = lane >> 5) operand
// layout of v_mfma_f32_32x32x16.  A nibble word becomes 8 f16 values (q - z), exact small integers: V_PERM_B32 drops each nibble under a
// 0x64 exponent byte (1024 + q), one packed subtract of (1024 + z) -- 12 VALU per fragment against >= 2 MFMAs of 16 passes.  The group scale is
// applied in f32 when a 128-k group's partial product is folded into the running sum, so the products are exact and the sums f32, as in the
// decode kernels and the oracle.  A fragments (16 contiguous bytes of an activation row per lane) come straight from global memory / L1.
// grid = (N / 64, ceil(S / 64)) single-wave blocks.
// ---------------------------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 q4_frag_f16(unsigned w, f16x2 mz) {   // mz = -(1024 + z) in both halves
  const unsigned wq = w ^ 0x88888888u;      // stored nibbles are (q - 8) in two's complement (the decode kernels' V_DOT8 operands): back to q
  const unsigned lo = wq & 0x0F0F0F0Fu, hi = (wq >> 4) & 0x0F0F0F0Fu;
  const unsigned p0 = __builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u), p1 = __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u);
  const unsigned p2 = __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u), p3 = __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u);
  uint4 r;
  r.x = __builtin_bit_cast(unsigned, (f16x2)(__builtin_bit_cast(f16x2, p0) + mz));
  r.y = __builtin_bit_cast(unsigned, (f16x2)(__builtin_bit_cast(f16x2, p1) + mz));
  r.z = __builtin_bit_cast(unsigned, (f16x2)(__builtin_bit_cast(f16x2, p2) + mz));
  r.w = __builtin_bit_cast(unsigned, (f16x2)(__builtin_bit_cast(f16x2, p3) + mz));
  return r;
}

template <int WPB, int MT = 2>   // waves per block, side by side along N; MT 32-row tiles per wave (1: decode batches / prompts of <= 32 rows -- half the MFMAs and half the activation loads)
__global__ __launch_bounds__(WPB * 64) void k_gemm_q4g_mfma(const uint4* __restrict__ W, const __half* __restrict__ Sc, const unsigned char* __restrict__ Z,
                                                       const float* __restrict__ bias, int N, int K, const unsigned short* __restrict__ X, int S, int act,
                                                       float* __restrict__ Y, int GPB, float* __restrict__ part) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x * WPB + wave;            // 64-column tile of this wave
  if (tile * 64 >= N) return;
  const int r0 = blockIdx.y * 32 * MT;
  const int G = K >> 7, C32 = K >> 5;
  const uint4* wp = W + (size_t)tile * C32 * 64 + lane;
  const unsigned short* xa0 = X + (size_t)min(r0 + c, S - 1) * K + 8 * h;        // rows r0 + c and r0 + 32 + c (clamped; masked at the store)
  const unsigned short* xa1 = X + (size_t)min(r0 + 32 + c, S - 1) * K + 8 * h;
  const __half* sp = Sc + (size_t)tile * G * 64 + c;
  const unsigned char* zp = Z + (size_t)tile * G * 64 + c;
  f32x16 tot[MT][2], grp[MT][2];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) tot[a][b][i] = 0.f;
  // software pipeline: chunk kc+1's loads are in flight under chunk kc's MFMAs
  const int kc0 = blockIdx.z * GPB * 4;
  u32x4 wn = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)kc0 * 64));
  uint4 an[2][2];
  an[0][0] = *(const uint4*)(xa0 + kc0 * 32); an[0][1] = *(const uint4*)(xa0 + kc0 * 32 + 16);
  if (MT == 2) { an[1][0] = *(const uint4*)(xa1 + kc0 * 32); an[1][1] = *(const uint4*)(xa1 + kc0 * 32 + 16); }
  // K split (small S): block z owns groups [z GPB, (z+1) GPB) and writes an unrounded partial; k_q4g_mfma_reduce sums them in a fixed order
  const int g_beg = blockIdx.z * GPB, g_end = min(G, g_beg + GPB);
  for (int g = g_beg; g < g_end; g++) {
    const float s0 = __half2float(sp[(size_t)g * 64]), s1 = __half2float(sp[(size_t)g * 64 + 32]);
    const _Float16 m0 = (_Float16)(-(1024.0f + (float)zp[(size_t)g * 64])), m1 = (_Float16)(-(1024.0f + (float)zp[(size_t)g * 64 + 32]));
    const f16x2 mz0 = {m0, m0}, mz1 = {m1, m1};
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) grp[a][b][i] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
      const int kc = g * 4 + cc;
      const u32x4 w = wn;
      uint4 a[2][2];
      a[0][0] = an[0][0]; a[0][1] = an[0][1];
      if (MT == 2) { a[1][0] = an[1][0]; a[1][1] = an[1][1]; }
      {
        const int kn = min(kc + 1, C32 - 1);            // last chunk: a harmless re-load
        wn = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)kn * 64));
        an[0][0] = *(const uint4*)(xa0 + kn * 32); an[0][1] = *(const uint4*)(xa0 + kn * 32 + 16);
        if (MT == 2) { an[1][0] = *(const uint4*)(xa1 + kn * 32); an[1][1] = *(const uint4*)(xa1 + kn * 32 + 16); }
      }
      const u32x2 r01 = __builtin_amdgcn_permlane32_swap(w.x, w.y, false, false);   // .x: columns 0-31 {k 0-7 | k 8-15}; .y: columns 32-63
      const u32x2 r23 = __builtin_amdgcn_permlane32_swap(w.z, w.w, false, false);   // same for k 16-31
      const uint4 b00 = q4_frag_f16(r01.x, mz0), b01 = q4_frag_f16(r01.y, mz1);
      const uint4 b10 = q4_frag_f16(r23.x, mz0), b11 = q4_frag_f16(r23.y, mz1);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        grp[mt][0] = mfma16<BZ_F16>(a[mt][0], b00, grp[mt][0]);
        grp[mt][1] = mfma16<BZ_F16>(a[mt][0], b01, grp[mt][1]);
        grp[mt][0] = mfma16<BZ_F16>(a[mt][1], b10, grp[mt][0]);
        grp[mt][1] = mfma16<BZ_F16>(a[mt][1], b11, grp[mt][1]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 16; i++) { tot[mt][0][i] = fmaf(s0, grp[mt][0][i], tot[mt][0][i]); tot[mt][1][i] = fmaf(s1, grp[mt][1][i], tot[mt][1][i]); }
  }
  // C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int T = 0; T < 2; T++) {
    const int n = tile * 64 + 32 * T + c;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int m = r0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < S) {
          if (part) part[((size_t)blockIdx.z * S + m) * N + n] = tot[mt][T][i];
          else Y[(size_t)m * N + n] = pf_round(tot[mt][T][i] + bv, act);
        }
      }
  }
}

// The same arithmetic with both operands shared through LDS (prompts of more than 64 rows): workgroup tile 128 rows x 128 columns (two 64-column weight
// tiles) x 64 k, 2 x 2 waves of 64 x 64.  Activations go global -> LDS by LDS-DMA exactly as in k_gemm_nt2 (128-byte rows, pieces XOR-swizzled by row & 7);
// the int4 weights are DMA'd as they lie in memory -- a (tile, 32-k chunk) is 64 lanes x 16 B = one 1 KB wave-instruction -- and each wave rebuilds its four
// B fragments per chunk from ONE 16-byte LDS read per lane (two V_PERMLANE32_SWAPs + the V_PERM nibble trick above).  The activation rows are fetched once
// per workgroup instead of once per wave (k_gemm_q4g_mfma re-reads them for every 64-column tile: 4 x the weight bytes at f16), the weight chunk once per
// 128 rows.  A group (128 k) is two k-steps: the group accumulator (started from the constant 0 by the group's first MFMA) is folded into the total with the f32
// scale after the second one; the next group's scales / zero points come by LDS-DMA too, a whole group ahead of their use.
// Three 20 KB buffers (prefetch distance two steps; occupancy is bound by registers, not LDS); K may be split in units of groups (partials summed by
// k_q4g_mfma_reduce).     grid = 8 ceil(ntiles / 8) * mtiles * KS, 256 threads
// WIDE (<= 64 rows: decode batches, short prompts): the four waves sit side by side on FOUR column tiles and share ONE 64-row activation tile -- 8 KB of
// activations per 8 KB of weights and step, where the square tile moves 16 KB of (half clamped) activation rows per 4 KB of weights
template <bool WIDE>
__global__ __launch_bounds__(256) void k_gemm_q4g_lds(const uint4* __restrict__ W, const __half* __restrict__ Sc, const unsigned char* __restrict__ Z,
                                                      const float* __restrict__ bias, int N, int K, const unsigned short* __restrict__ X, int S, int act,
                                                      float* __restrict__ Y, float* __restrict__ part, int KS, int mtiles, int ntiles) {
  constexpr int BM = WIDE ? 64 : 128, CT = WIDE ? 4 : 2, ABYTES = BM * 128, WBYTES = CT * 2048, TILE = ABYTES + WBYTES, NA = BM / 32, NW = CT / 2;   // NA / NW: A row groups / weight chunks per wave and step
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3[];   // the only LDS object
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt = jj % mtiles, rest = jj / mtiles, ks = rest % KS, nt = (rest / KS) * 8 + xcd;
  if (nt >= ntiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, wm = WIDE ? 0 : wave >> 1, wn = WIDE ? wave : wave & 1;
  const int m0 = mt * BM;
  const int G = K >> 7, C32 = K >> 5, ncol = N >> 6;
  const int g_beg = (int)((long long)ks * G / KS), g_end = (int)((long long)(ks + 1) * G / KS), nsteps = 2 * (g_end - g_beg);
  const int ctile = min(nt * CT + wn, ncol - 1);              // this wave's 64-column tile (past the edge: a masked duplicate of the last one)
  const bool cols_on = nt * CT + wn < ncol;
  // staging: every wave fills 4 row groups of A (8 rows x 128 B each) and ONE weight chunk: (column tile wave >> 1, chunk wave & 1) of the step
  const int lrow = lane >> 3, piece = (lane & 7) ^ lrow;
  unsigned xo[NA];
#pragma unroll
  for (int i = 0; i < NA; i++) xo[i] = (unsigned)min(m0 + (wave * NA + i) * 8 + lrow, S - 1) * (unsigned)K + 8u * piece;
  // weight chunks: square form -- wave w stages (column tile w >> 1, chunk w & 1); wide form -- wave w stages both chunks of its own column tile
  const uint4* wsrc = W + (size_t)min(nt * CT + (WIDE ? wave : wave >> 1), ncol - 1) * C32 * 64 + lane;
  auto issue = [&](int step, int buf) {                       // step = absolute 64-k step index
    unsigned char* base = smem3 + buf * TILE;
#pragma unroll
    for (int i = 0; i < NA; i++) BZ_GLDS16(X + xo[i] + (size_t)step * 64, base + (wave * NA + i) * 1024);
    if constexpr (WIDE) {
      BZ_GLDS16(wsrc + (size_t)(2 * step) * 64, base + ABYTES + wave * 2048);
      BZ_GLDS16(wsrc + (size_t)(2 * step + 1) * 64, base + ABYTES + wave * 2048 + 1024);
    } else BZ_GLDS16(wsrc + (size_t)(2 * step + (wave & 1)) * 64, base + ABYTES + wave * 1024);
  };
  f32x16 tot[2][2], grp[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) { tot[a][b][i] = 0.f; grp[a][b][i] = 0.f; }
  if (nsteps > 0) {
    // three buffers, prefetch distance two steps.  EVERY load of the loop is an LDS-DMA -- the group scales / zero points too (one 4-byte-per-lane DMA per
    // wave and group into a 256-byte slot: lanes 0-31 two f16 scales each, lanes 32-47 four zero points each) -- because the counted waits rely on loads
    // retiring in issue order, which held among LDS-DMAs but NOT between LDS-DMAs and ordinary VGPR loads (a version with register scale loads returned
    // sparse wrong tiles).  Order of issue: tile 0, group g_beg's scales, tile 1 | step it: tile it + 2 (+ on even steps the scales of step it + 2's group).
    // "tile `it` (and, on even steps, its group's scales) landed" = vmcnt(5) on even steps (only tile it + 1's 5 DMAs may be out), vmcnt(6) on odd ones.
    const int st0 = 2 * g_beg;
    unsigned char* sslot = smem3 + 3 * TILE + wave * 256;      // + 1024 for odd groups
    const unsigned char* ssrc = lane < 32 ? (const unsigned char*)(Sc + (size_t)ctile * G * 64) + 4 * lane : (const unsigned char*)(Z + (size_t)ctile * G * 64) + 4 * (lane & 15);
    const size_t sstep = lane < 32 ? 128 : 64;                  // bytes per group in the scale / zero arrays
    issue(st0, 0);
    glds4(ssrc + (size_t)g_beg * sstep, sslot);
    issue(st0 + min(1, nsteps - 1), 1);
    const int aoff = (wm * 64 + r) * 128, sw = r & 7;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int gi = 0; gi < g_end - g_beg; gi++) {
      float s0 = 0.f, s1 = 0.f;
      f16x2 mz0 = {0, 0}, mz1 = {0, 0};
#pragma unroll
      for (int hs = 0; hs < 2; hs++) {
        const int it = 2 * gi + hs;
        // DMAs per tile and wave: 4 + 1 (square), 2 + 2 (wide)
        if constexpr (WIDE) { if (hs == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else { if (hs == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();                        // everyone's pieces of tile `it` have landed; everyone is done reading the buffer the next issue overwrites
        if (hs == 0) {
          const unsigned char* sb = sslot + (gi & 1) * 1024;
          s0 = __half2float(*(const __half*)(sb + 2 * r)); s1 = __half2float(*(const __half*)(sb + 2 * (r + 32)));
          const _Float16 a0 = (_Float16)(-(1024.0f + (float)sb[128 + r])), a1 = (_Float16)(-(1024.0f + (float)sb[128 + 32 + r]));
          mz0 = f16x2{a0, a0}; mz1 = f16x2{a1, a1};
        }
        issue(st0 + min(it + 2, nsteps - 1), (it + 2) % 3);   // clamped: redundant reloads at the tail, never a branch around a load
        if (hs == 0) glds4(ssrc + (size_t)min(g_beg + gi + 1, g_end - 1) * sstep, sslot + ((gi + 1) & 1) * 1024);   // the next group's scales, a whole group ahead of their use
        const unsigned char* tb = smem3 + (it % 3) * TILE;
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
          const u32x4 w = *(const u32x4*)(tb + ABYTES + (wn * 2 + cc) * 1024 + lane * 16);
          const u32x2 r01 = __builtin_amdgcn_permlane32_swap(w.x, w.y, false, false);   // .x: columns 0-31 {k 0-7 | k 8-15}; .y: columns 32-63
          const u32x2 r23 = __builtin_amdgcn_permlane32_swap(w.z, w.w, false, false);   // same for k 16-31
          const uint4 b00 = q4_frag_f16(r01.x, mz0), b01 = q4_frag_f16(r01.y, mz1);
          const uint4 b10 = q4_frag_f16(r23.x, mz0), b11 = q4_frag_f16(r23.y, mz1);
#pragma unroll
          for (int t = 0; t < 2; t++) {
            // row (wm 64 + 32 t + r), k = 32 cc + 16 step + 8 h .. + 7 of the 64-k tile: piece 4 cc + 2 step + h
            const uint4 a0 = *(const uint4*)(tb + aoff + t * 32 * 128 + (((4 * cc + h) ^ sw) * 16));
            const uint4 a1 = *(const uint4*)(tb + aoff + t * 32 * 128 + (((4 * cc + 2 + h) ^ sw) * 16));
            const bool first = hs == 0 && cc == 0;            // compile time: a group's first MFMA takes the constant 0 as C (no zeroing pass)
            grp[t][0] = mfma16<BZ_F16>(a0, b00, first ? zero16 : grp[t][0]);
            grp[t][1] = mfma16<BZ_F16>(a0, b01, first ? zero16 : grp[t][1]);
            grp[t][0] = mfma16<BZ_F16>(a1, b10, grp[t][0]);
            grp[t][1] = mfma16<BZ_F16>(a1, b11, grp[t][1]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) { tot[t][0][i] = fmaf(s0, grp[t][0][i], tot[t][0][i]); tot[t][1][i] = fmaf(s1, grp[t][1][i], tot[t][1][i]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tail's redundant LDS-DMAs land before the LDS allocation is released
  }
  if (cols_on) {
#pragma unroll
    for (int T = 0; T < 2; T++) {
      const int n = ctile * 64 + 32 * T + r;
      const float bv = (!part && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int m = m0 + wm * 64 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (m < S) {
            if (part) part[((size_t)ks * S + m) * N + n] = tot[t][T][i];
            else Y[(size_t)m * N + n] = pf_round(tot[t][T][i] + bv, act);
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// EXACT W4 GEMM on the integer matrix cores: the decode kernels' arithmetic (bz_dev.h: x = c * code exactly, c a power of two per 128-k group, code a 32-bit
// integer; sum_k (q_k - z) code_k an exact integer) at prompt scale.  The 32-bit code is FOUR balanced int8 digits (code = sum_p 256^p b_p, b_p in [-128, 127]:
// the bytes of (code + 0x80808080) ^ 0x80808080), the stored nibbles are (q - 8) in two's complement, so (w << 4) & 0xF0F0F0F0 / w & 0xF0F0F0F0 ARE the int8
// values 16 (q - 8) of four k each -- six VALU per 32-k chunk and tile, no table, no subtract.  v_mfma_i32_32x32x32_i8 (operand map verified with exact integer
// data: scripts/mfma_i8_probe.hip) sums one digit plane of a 32-k chunk per instruction: |plane sum of a group| <= 128 * 128 * 128 = 2^21, exact in int32.
// Per 128-k group and output the four plane sums are joined (two shifted int32 adds, then one double FMA: < 2^47), the zero point comes in through the row's
// code sum ((8 - z) * sum_k code_k), and the group's term s * c * Int is added into a double per output -- q4_term's value, in q4g_consume's order of groups.
// A lane's column is fixed (C layout: col = lane & 31), so scale and zero point are per-lane scalars of a group; the row parameters (c, code sum) of the
// workgroup's 32 rows sit in LDS for all groups.  A wave owns 32 rows x 64 columns and walks its two 32-column tiles one after the other over the SAME four weight
// chunks (64 accumulator registers instead of 128: two waves per SIMD, one folding while the other multiplies); the activation digits come straight from
// global memory / L1 (16 bytes per lane, plane and chunk; the WPB waves of a workgroup sit side by side on the same rows).
//   Xq: [4 planes][S][K] int8        par: [S][K / 128] { double code sum ; float c ; pad }        grid = (ceil(N / 64 / WPB), ceil(S / 32))
// ---------------------------------------------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((aligned(16))) RowPar { double rsum; float c; float pad; };

template <int XDT>
__global__ __launch_bounds__(256) void k_pf_quant_i8(const unsigned short* __restrict__ X, int K, signed char* __restrict__ Xq, size_t plane_stride, RowPar* __restrict__ par) {
  const int s = blockIdx.x, G = K >> 7;
  const unsigned short* xr = X + (size_t)s * K;
  for (int e0 = threadIdx.x * 8; e0 < K; e0 += 256 * 8) {        // 16 lanes per 128-k group (K % 128 == 0, so a group never straddles the loop)
    const uint4 raw = *(const uint4*)(xr + e0);
    const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = from16<XDT>((unsigned short)(u[i] & 0xffffu)); v[2 * i + 1] = from16<XDT>((unsigned short)(u[i] >> 16)); }
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) am = fmaxf(am, fabsf(v[i]));
    am = grp_reduce<16, OpMax>(am);
    const unsigned eb = (__float_as_uint(am) >> 23) & 255u;               // as xq_split8: am * 2^e in [2^29, 2^30)
    const bool live = eb >= 32u && eb < 255u;
    const float inv = live ? __uint_as_float((283u - eb) << 23) : 0.f, cs = live ? __uint_as_float((eb - 29u) << 23) : 0.f;
    unsigned cb[8]; double rs = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const int code = (int)rintf(v[i] * inv); rs += (double)code; cb[i] = ((unsigned)code + 0x80808080u) ^ 0x80808080u; }
    rs = grp_sum_d<16>(rs);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const unsigned sel = 0x0c0c0000u | ((4u + p) << 8) | (unsigned)p;   // [lo.byte p, hi.byte p, 0, 0]
      uint2 w;
      w.x = __builtin_amdgcn_perm(cb[1], cb[0], sel) | (__builtin_amdgcn_perm(cb[3], cb[2], sel) << 16);
      w.y = __builtin_amdgcn_perm(cb[5], cb[4], sel) | (__builtin_amdgcn_perm(cb[7], cb[6], sel) << 16);
      *(uint2*)(Xq + (size_t)p * plane_stride + (size_t)s * K + e0) = w;
    }
    if ((threadIdx.x & 15) == 0) { RowPar rp; rp.rsum = rs; rp.c = cs; rp.pad = 0.f; par[(size_t)s * G + (e0 >> 7)] = rp; }
  }
}

template <int WPB, int DBG = 0, bool SEQ = true>     // DBG (timing experiments, BZ_I8_DBG): 1 = no fold, 2 = no MFMAs, 3 = no LDS fragment reads, 4 = interleaved tiles
__global__ __launch_bounds__(WPB * 64) void k_gemm_q4g_i8(const uint4* __restrict__ W, const __half* __restrict__ Sc, const unsigned char* __restrict__ Z, const float* __restrict__ bias,
                                                           int N, int K, const signed char* __restrict__ Xq, size_t plane_stride, const RowPar* __restrict__ par, int S, int act,
                                                           float* __restrict__ Y) {
  // LDS, two buffers: [4 planes][32 rows][128 B] digits of one 128-k group (16-byte pieces XOR-swizzled by row & 7: conflict-free ds_read_b128 of a 32-row column
  // of pieces) + the 32 rows' parameters.  The workgroup's waves share the rows, both column tiles of a wave share them again: one global read per group and workgroup.
  constexpr int NT = WPB * 64, PL = 4 * 32 * 128, BUF = PL + 32 * (int)sizeof(RowPar), LPT = PL / 16 / NT;   // LPT: 16-byte pieces per thread and group (4 or 8)
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * BUF];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  // XCD-aware order (1-D grid): consecutive workgroups of one XCD (blockIdx.x % 8) walk the ROW tiles of one column group, so a group's weights cross HBM -> L2 once
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, mt = (S + 31) / 32, cg = (jj / mt) * 8 + xcd;
  if (cg * WPB * 64 >= N) return;                              // (whole workgroup: no barrier is skipped by part of it)
  const int G = K >> 7, C32 = K >> 5, r0 = (jj % mt) * 32;
  const int tile = min(cg * WPB + wave, N / 64 - 1);           // 64-column tile of this wave (past the edge: a masked duplicate -- every wave stays for the barriers)
  const bool cols_on = (cg * WPB + wave) * 64 < N;
  const uint4* wp = W + (size_t)tile * C32 * 64 + lane;
  const __half* sp = Sc + (size_t)tile * G * 64 + c;
  const unsigned char* zp = Z + (size_t)tile * G * 64 + c;
  // staging map: piece i of this thread = (plane, row, 16-byte column) of the group's digit tile
  const signed char* src[LPT]; int dst[LPT];
#pragma unroll
  for (int i = 0; i < LPT; i++) {
    const int idx = tid + NT * i, p = idx >> 8, row = (idx >> 3) & 31, pc = idx & 7;
    src[i] = Xq + (size_t)p * plane_stride + (size_t)min(r0 + row, S - 1) * K + pc * 16;
    dst[i] = p * 4096 + row * 128 + ((pc ^ (row & 7)) << 4);
  }
  const RowPar* psrc = par + (size_t)min(r0 + (tid & 31), S - 1) * G;
  uint4 st[LPT]; RowPar stp;
#pragma unroll
  for (int i = 0; i < LPT; i++) st[i] = *(const uint4*)(src[i]);
  stp = psrc[0];
  double tot[2][16];
#pragma unroll
  for (int T = 0; T < 2; T++)
#pragma unroll
    for (int i = 0; i < 16; i++) tot[T][i] = 0.0;
  u32x4 wn[4], wn2[4];                                         // the weights of groups g + 1 and g + 2: a cold 16-byte load takes longer than one group's arithmetic
#pragma unroll
  for (int cc = 0; cc < 4; cc++) wn[cc] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)cc * 64));
#pragma unroll
  for (int cc = 0; cc < 4; cc++) wn2[cc] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)(min(1, G - 1) * 4 + cc) * 64));
  const int aoff = c * 128, sw = c & 7;
  for (int g = 0; g < G; g++) {
    unsigned char* buf = sm + (g & 1) * BUF;
#pragma unroll
    for (int i = 0; i < LPT; i++) *(uint4*)(buf + dst[i]) = st[i];
    if (tid < 32) ((RowPar*)(buf + PL))[tid] = stp;
    u32x4 w[4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) { w[cc] = wn[cc]; wn[cc] = wn2[cc]; }
    {
      const int gn = min(g + 1, G - 1), gn2 = min(g + 2, G - 1);   // last groups: harmless re-loads
#pragma unroll
      for (int i = 0; i < LPT; i++) st[i] = *(const uint4*)(src[i] + (size_t)gn * 128);
      stp = psrc[gn];
#pragma unroll
      for (int cc = 0; cc < 4; cc++) wn2[cc] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)(gn2 * 4 + cc) * 64));
    }
    const float sT[2] = {__half2float(sp[(size_t)g * 64]), __half2float(sp[(size_t)g * 64 + 32])};
    const int zT[2] = {(int)zp[(size_t)g * 64], (int)zp[(size_t)g * 64 + 32]};
    __syncthreads();                                          // the group's tile is in LDS; everyone is done with the buffer the next iteration overwrites
    const RowPar* rp = (const RowPar*)(buf + PL);
#pragma unroll
    for (int T = 0; T < 2; T++) {
      i32x16 acc[4];
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[p][i] = 0;
      i32x4 an[4];                                                                           // the next chunk's digit fragments: read from LDS under this chunk's MFMAs
#pragma unroll
      for (int p = 0; p < 4; p++) an[p] = *(const i32x4*)(buf + p * 4096 + aoff + ((h ^ sw) << 4));
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
        i32x4 a[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
          if (DBG == 3) a[p] = i32x4{(int)w[cc].x, p, cc, lane};
          else a[p] = an[p];                                                                 // row c, k 32 cc + 16 h .. + 15
        }
        if (cc < 3) {
#pragma unroll
          for (int p = 0; p < 4; p++) an[p] = *(const i32x4*)(buf + p * 4096 + aoff + (((2 * (cc + 1) + h) ^ sw) << 4));
        }
        // this tile's nibble words: lanes (c, h) need k 16 h .. 16 h + 15 of column 32 T + c
        const u32x2 rxz = __builtin_amdgcn_permlane32_swap(w[cc].x, w[cc].z, false, false);   // .x: {cols 0-31: k 0-7 | k 16-23}   .y: the same for cols 32-63
        const u32x2 ryw = __builtin_amdgcn_permlane32_swap(w[cc].y, w[cc].w, false, false);   // .x: {cols 0-31: k 8-15 | k 24-31}  .y: cols 32-63
        const unsigned n0 = T == 0 ? rxz.x : rxz.y, n1 = T == 0 ? ryw.x : ryw.y;
        i32x4 b;
        b.x = (int)((n0 << 4) & 0xF0F0F0F0u); b.y = (int)(n0 & 0xF0F0F0F0u); b.z = (int)((n1 << 4) & 0xF0F0F0F0u); b.w = (int)(n1 & 0xF0F0F0F0u);
#pragma unroll
        for (int p = 0; p < 4; p++) {
          if (DBG == 2) acc[p][cc] += a[p].x ^ b.x;
          else acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b, acc[p], 0, 0, 0);
        }
      }
      const double zz = (double)(8 - zT[T]), sd = (double)sT[T];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (DBG == 1) { tot[T][i] += (double)(acc[0][i] ^ acc[1][i] ^ acc[2][i] ^ acc[3][i]); continue; }
        const RowPar q = rp[(i & 3) + 8 * (i >> 2) + 4 * h];
        const int lo = acc[0][i] + acc[1][i] * 256, hi = acc[2][i] + acc[3][i] * 256;                 // 16 x the digit-weighted sums: < 2^21 + 2^29
        const double in16 = fma((double)hi, 65536.0, (double)lo);                                    // 16 * sum_k (q_k - 8) code_k
        const double in = fma(zz, q.rsum, in16 * 0.0625);                                            // sum_k (q_k - z) code_k, exact
        tot[T][i] = fma(in * (double)q.c, sd, tot[T][i]);                                            // + s c Int (the product is exact in double)
      }
      if (SEQ) __builtin_amdgcn_sched_barrier(0);     // SEQ: the two tiles one after the other (64 accumulator registers, two waves per SIMD); else the compiler interleaves them (128, one wave per SIMD)
    }
  }
  if (!cols_on) return;
#pragma unroll
  for (int T = 0; T < 2; T++) {
    const int n = tile * 64 + 32 * T + c;
    const double bv = bias ? (double)bias[n] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int m = r0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (m < S) Y[(size_t)m * N + n] = pf_round((float)(tot[T][i] + bv), act);
    }
  }
}

__global__ void k_q4g_mfma_reduce(const float* __restrict__ part, int KS, size_t SN, int N, const float* __restrict__ bias, int act, float* __restrict__ Y,
                                  const float* __restrict__ rscale = nullptr) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < SN; i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < KS; z++) v += part[(size_t)z * SN + i];      // fixed order: deterministic
    if (rscale) v *= rscale[i / (size_t)N];
    Y[i] = pf_round(v + (bias ? bias[i % N] : 0.f), act);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Mamba2 batched prefill: the prompt's rows go through the GEMMs together (in_proj / out_proj on the matrix cores); the depthwise conv is
// a parallel map over (token, channel); the SSM recurrence is a scan INSIDE one kernel -- one workgroup per head keeps its state in
// registers and walks the tokens, so a prompt costs one pass over the state instead of a read-modify-write of it per token.
// Arithmetic and rounding points are those of the decode-step kernels (k_ssm_step with its in-launch conv1d step, the GATED2 prologue).
// ---------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pf_silu(float x) { return div_rn(x, 1.0f + bz_expf(-x)); }
__device__ __forceinline__ float pf_softplus(float x) { return x > 20.0f ? x : (float)log1p((double)bz_expf(x)); }   // as the step kernel (and the oracle): log1p in double

// out[t][ch] = R(silu(R(conv window + bias))): window = the kc-1 inputs before t (from the rows, or from the carried conv state) and in[t]
__global__ void k_pf_conv(const float* __restrict__ zx, int ld, int x_off, int conv_dim, int kc, const float* __restrict__ w, const float* __restrict__ b,
                          const float* __restrict__ cs, int S, int act, float* __restrict__ out) {
  const int ch = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  if (ch >= conv_dim) return;
  float a = 0.f;
  for (int j = 0; j < kc - 1; j++) {
    const int tau = t - (kc - 1) + j;
    const float v = tau >= 0 ? zx[(size_t)tau * ld + x_off + ch] : cs[(size_t)ch * (kc - 1) + (tau + kc - 1)];
    a += v * w[(size_t)ch * kc + j];
  }
  a += zx[(size_t)t * ld + x_off + ch] * w[(size_t)ch * kc + kc - 1];
  a = pf_round(a + b[ch], act);
  out[(size_t)t * conv_dim + ch] = pf_round(pf_silu(a), act);
}
// conv state after the prompt: the last kc-1 inputs
__global__ void k_pf_conv_state(const float* __restrict__ zx, int ld, int x_off, int conv_dim, int kc, float* cs, int S) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= conv_dim) return;
  float nv[8];
  for (int j = 0; j < kc - 1; j++) {
    const int tau = S - (kc - 1) + j;
    nv[j] = tau >= 0 ? zx[(size_t)tau * ld + x_off + ch] : cs[(size_t)ch * (kc - 1) + (tau + kc - 1)];
  }
  for (int j = 0; j < kc - 1; j++) cs[(size_t)ch * (kc - 1) + j] = nv[j];
}

struct SsmScanArgs {
  const float* xbc; int conv_dim;          // rows [S][conv_dim]: x (d_inner) | B (groups x d_state) | C (groups x d_state)
  const float* zx; int ld; int dt_off;     // rows [S][ld]: z at 0, dt at dt_off
  const float* dt_bias; const float* A_log; const float* D;
  void* state;                             // this layer's [n_heads][head_dim][d_state]
  int n_heads, head_dim, d_state, n_groups, d_inner, act, S;
  float* y;                                // rows [S][d_inner]: R(y * R(silu z))
  float* vss;                              // rows [S][n_heads]: sum of squares of the gated y per head
};

template <int SDT> __device__ __forceinline__ float st_load(const void* p, size_t i) {
  if (SDT == BZ_F32) return ((const float*)p)[i];
  if (SDT == BZ_F16) return __half2float(((const __half*)p)[i]);
  return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16);
}
template <int SDT> __device__ __forceinline__ void st_store(void* p, size_t i, float v) {
  if (SDT == BZ_F32) ((float*)p)[i] = v;
  else if (SDT == BZ_F16) ((__half*)p)[i] = f16_cvt(v);
  else ((unsigned short*)p)[i] = to16<BZ_BF16>(v);
}
template <int SDT> __device__ __forceinline__ float st_round(float v) {     // what a store + load of the state dtype does to a value
  if (SDT == BZ_F32) return v;
  if (SDT == BZ_F16) return f16_round((v));
  return from16<BZ_BF16>(to16<BZ_BF16>(v));
}

// compile-time rounding to the activation dtype (bf16: v_cvt_pk_bf16_f32 + shift, two instructions on gfx950)
template <int ACT> __device__ __forceinline__ float rnd(float x) {
  if constexpr (ACT == BZ_F16) return f16_round(x);
  else if constexpr (ACT == BZ_BF16) return (float)(__bf16)x;
  else return x;
}

// The recurrence is sequential in t (the state is rounded at every token, as the decode step stores it), so the kernel is bound by the instructions one
// CU issues per token.  What keeps that count low: rounding is compile-time (ACT); a head's rows are split over ceil(head_dim / 16) workgroups; B / C
// come out of LDS as 16-byte reads; per token only the state update, the C dot and its 16-lane reduction run (dt * x comes staged from LDS) -- the
// gate / SiLU / y epilogue runs once per CHUNK with lane q of a row serving token q (it used to run per token on one lane in 16); the next chunk's
// rows are requested before the current chunk's arithmetic, and a chunk's stores are issued one chunk late, ahead of that request.
// grid = (n_heads, ceil(head_dim / 16)); 256 threads = 16 rows (p) x 16 state parts (q); NQ = d_state / 16 state values per thread in registers;
// tokens in chunks of 8.  vss rows are [S][n_heads][gridDim.y] (k_pf_gnorm adds the pieces)
template <int SDT, int ACT, int NQ>
__global__ __launch_bounds__(256) void k_ssm_scan(SsmScanArgs a) {
  constexpr int TC = 8, PARTS = 16, NS = NQ * PARTS, NP = NQ == 8 ? 12 : NQ, NTH = 256, NWV = 4;   // NP: part stride in LDS (16-byte aligned for NQ >= 4)
  constexpr int NBC = (TC * NS + NTH - 1) / NTH;     // B and C elements per thread per chunk
  __shared__ __attribute__((aligned(16))) float sB[TC][PARTS * NP], sC[TC][PARTS * NP];
  __shared__ float sdA[TC], sdtx[TC][16], sred[TC][NWV];
  const int hd = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HD = a.head_dim;
  const int g = hd / (a.n_heads / a.n_groups);
  const int pl = tid / PARTS, p = blockIdx.y * 16 + pl, q = tid % PARTS, tq = q & (TC - 1);   // tq: the token of a chunk this lane serves in the epilogue
  const bool on = p < HD;
  const int pc = on ? p : 0;
  const float Dh = a.D[hd], Aneg = -bz_expf(a.A_log[hd]), dtb = a.dt_bias[hd];
  const size_t soff = ((size_t)hd * HD + pc) * NS + q * NQ;
  float h[NQ];
#pragma unroll
  for (int n = 0; n < NQ; n++) h[n] = on ? st_load<SDT>(a.state, soff + n) : 0.f;
  // every global load of a chunk is unconditional (clamped rows / channels): a load under a divergent branch drains vmcnt each time
  float bv[NBC], cv[NBC], dtraw, xn, zn;
  auto request = [&](int t0) {
#pragma unroll
    for (int k = 0; k < NBC; k++) {
      const int i = min(k * NTH + tid, TC * NS - 1), tt = i / NS, n = i % NS;
      const float* row = a.xbc + (size_t)min(t0 + tt, a.S - 1) * a.conv_dim;
      bv[k] = row[a.d_inner + g * NS + n];
      cv[k] = row[a.d_inner + a.n_groups * NS + g * NS + n];
    }
    const size_t t = (size_t)min(t0 + tq, a.S - 1);
    dtraw = a.zx[t * a.ld + a.dt_off + hd];
    xn = a.xbc[t * a.conv_dim + hd * HD + pc];
    zn = a.zx[t * a.ld + hd * HD + pc];
  };
  request(0);
  float yprev = 0.f, vssv = 0.f;
  int tprev = -1, ntprev = 0;
  auto flush = [&]() {     // (stores and loads share vmcnt on gfx9: a store right before the wait for the prefetched rows would put its round trip into every chunk)
    if (tprev < 0) return;
    if (on && q < ntprev) a.y[(size_t)(tprev + q) * a.d_inner + hd * HD + p] = yprev;
    if (tid < ntprev) a.vss[((size_t)(tprev + tid) * a.n_heads + hd) * gridDim.y + blockIdx.y] = vssv;
  };
  for (int t0 = 0; t0 < a.S; t0 += TC) {
    const int nt = min(TC, a.S - t0);
    // (no barrier here: every wave passed the one after the sred writes, i.e. is done with the previous chunk's sB / sC / sdA / sdtx)
#pragma unroll
    for (int k = 0; k < NBC; k++) {
      const int i = k * NTH + tid, tt = i / NS, n = i % NS;
      if (i < TC * NS) { sB[tt][(n / NQ) * NP + n % NQ] = bv[k]; sC[tt][(n / NQ) * NP + n % NQ] = cv[k]; }
    }
    const float xc = xn, zc = zn;
    {
      const float dt = rnd<ACT>(pf_softplus(rnd<ACT>(dtraw + dtb)));     // token tq
      if (q < TC) sdtx[q][pl] = dt * xc;
      if (tid < TC) sdA[tid] = bz_expf(dt * Aneg);
    }
    __syncthreads();
    flush();
    if (t0 + TC < a.S) request(t0 + TC);   // uniform branch; the loads land under this chunk's arithmetic
    float accs[TC];
#pragma unroll
    for (int tt = 0; tt < TC; tt++) {
      accs[tt] = 0.f;
      if (tt < nt) {
        const float dA = sdA[tt], dtx = sdtx[tt][pl];
        float Bq[NQ], Cq[NQ];
        if constexpr (NQ % 4 == 0) {
#pragma unroll
          for (int n = 0; n < NQ; n += 4) {
            const float4 b4 = *(const float4*)&sB[tt][q * NP + n], c4 = *(const float4*)&sC[tt][q * NP + n];
            Bq[n] = b4.x; Bq[n + 1] = b4.y; Bq[n + 2] = b4.z; Bq[n + 3] = b4.w; Cq[n] = c4.x; Cq[n + 1] = c4.y; Cq[n + 2] = c4.z; Cq[n + 3] = c4.w;
          }
        } else {
#pragma unroll
          for (int n = 0; n < NQ; n++) { Bq[n] = sB[tt][q * NP + n]; Cq[n] = sC[tt][q * NP + n]; }
        }
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < NQ; n++) {
          const float hn = rnd<ACT>(h[n] * dA + dtx * Bq[n]);
          acc += hn * Cq[n];                   // (the step kernel multiplies the activation-rounded value and stores the state-dtype one)
          h[n] = ACT == SDT ? hn : st_round<SDT>(hn);
        }
        accs[tt] = grp_reduce<PARTS, OpAdd>(acc);
      }
    }
    // epilogue, once per chunk: lane q < nt of row p finishes token t0 + q
    float am = accs[0];
#pragma unroll
    for (int tt = 1; tt < TC; tt++) am = tq == tt ? accs[tt] : am;
    float y = 0.f;
    if (on && q < nt) {
      const float y0 = rnd<ACT>(am + Dh * xc);
      y = rnd<ACT>(y0 * rnd<ACT>(pf_silu(zc)));
    }
    yprev = y;
    // per-token sums of squares over this workgroup's rows: the wave's 4 rows by two lane exchanges, the 4 waves through LDS (fixed tree)
    float sq = y * y;
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < TC) sred[lane][wave] = sq;
    __syncthreads();
    if (tid < TC) vssv = (sred[tid][0] + sred[tid][1]) + (sred[tid][2] + sred[tid][3]);
    tprev = t0; ntprev = nt;
  }
  flush();
  if (on) {
#pragma unroll
    for (int n = 0; n < NQ; n++) st_store<SDT>(a.state, soff + n, h[n]);
  }
}

// row t: x16 = to16(R(w * R(v * rs_group))), rs_group = rsqrt(sum over the group's heads of vss / group size + eps)     grid = S
template <int DT>
__global__ __launch_bounds__(256) void k_pf_gnorm(const float* __restrict__ v, const float* __restrict__ vss, const float* __restrict__ w, int DI, int G, int NH,
                                                  float eps, int act, unsigned short* __restrict__ x16) {
  __shared__ float rs[64];
  const int t = blockIdx.x, gsz = DI / G, hpg = NH / G;
  for (int g = threadIdx.x >> 6; g < G; g += 4) {     // one wave per group: lanes stride over the group's pieces, fixed tree
    float ss = 0.f;
    for (int h = threadIdx.x & 63; h < hpg; h += 64) ss += vss[(size_t)t * NH + g * hpg + h];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) rs[g] = rms_scale(ss, (float)gsz, eps);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < DI; i += 256)
    x16[(size_t)t * DI + i] = to16<DT>(pf_round(w[i] * pf_round(v[(size_t)t * DI + i] * rs[i / gsz], act), act));
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------------
int bzk_gemm_nt(hipStream_t s, int dt, const void* x16, const void* w, const float* bias, int S, int N, int K, int act, float* y, float* ws, size_t ws_bytes, const float* rscale) {
  if (dt != BZ_F16 && dt != BZ_BF16) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: 16-bit operands only");
  if (K % 64 || K < 64 || S <= 0 || N <= 0) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: K=%d must be a positive multiple of 64", K);
  if ((unsigned long long)S * K >= (1ull << 32) || (unsigned long long)N * K >= (1ull << 32)) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: operand of 2^32 elements or more");
  static const bool old_kernel = getenv("BZ_GEMM_NT_WAVE_TILES") != nullptr;     // the round-1 kernel (weights straight to registers), kept for A/B runs and the grouped form
  const double flops = 2.0 * S * (double)N * K;
  if (!old_kernel) {
    // 128 x 128 tiles (64 x 128 for <= 64 rows); K split (a power of two) so that the chip sees about 256-320 workgroups, partials summed in a fixed order
    const int TM = S <= 64 ? 1 : 2, BM = 64 * TM;
    const int mtiles = (S + BM - 1) / BM, ntiles = (N + 127) / 128, nk = K / 64;
    const long long tiles = (long long)mtiles * ntiles;
    int KS = 1;
    if (ws) {
      while (KS * 2 * tiles <= 320 && KS * 2 <= nk / 2 && KS < 16 && (size_t)KS * 2 * S * N * 4 <= ws_bytes) KS *= 2;
    }
    float* part = KS > 1 ? ws : nullptr;
    const unsigned grid = 8u * (unsigned)((ntiles + 7) / 8) * (unsigned)mtiles * (unsigned)KS;
#define LAUNCH_G2(DT, M) do { \
      static bool attr_done = false; \
      if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_gemm_nt2<DT, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr_done = true; } \
      BZ_LAUNCH("gemm_nt_mfma", flops, (k_gemm_nt2<DT, M>), dim3(grid), dim3(256), 2 * (64 * M + 128) * 128, s, (const unsigned short*)x16, (const unsigned short*)w, bias, S, N, K, act, y, part, KS, \
                mtiles, ntiles, (const int*)nullptr, (const int*)nullptr, 0LL, rscale); } while (0)
    if (dt == BZ_F16) { if (TM == 1) LAUNCH_G2(BZ_F16, 1); else LAUNCH_G2(BZ_F16, 2); }
    else { if (TM == 1) LAUNCH_G2(BZ_BF16, 1); else LAUNCH_G2(BZ_BF16, 2); }
#undef LAUNCH_G2
    BZ_HIP(hipGetLastError());
    if (KS > 1) {
      const size_t SN = (size_t)S * N;
      hipLaunchKernelGGL(k_q4g_mfma_reduce, dim3((unsigned)std::min<size_t>((SN + 255) / 256, 2048)), dim3(256), 0, s, (const float*)ws, KS, SN, N, bias, act, y, rscale);
      BZ_HIP(hipGetLastError());
    }
    return BZ_OK;
  }
  if (rscale) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt: row scales need the LDS-DMA kernel (unset BZ_GEMM_NT_WAVE_TILES)");
  // wave tile (32 MT) x (32 NT): as large as the problem allows while still giving the chip >= 512 waves
  int MT = S > 96 ? 4 : (S > 64 ? 3 : (S > 32 ? 2 : 1)), NT = 2;
  auto waves = [&](int mt, int nt) { return (long long)((S + 32 * mt - 1) / (32 * mt)) * ((N + 32 * nt - 1) / (32 * nt)); };
  if (waves(MT, NT) < 512) NT = 1;
  while (MT > 1 && waves(MT, NT) < 512) MT = MT == 3 ? 2 : MT / 2;
  const dim3 grid((N + 128 * NT - 1) / (128 * NT), (S + 32 * MT - 1) / (32 * MT));
#define LAUNCH_GEMM(DT, M, NN) BZ_LAUNCH("gemm_nt_mfma", flops, (k_gemm_nt<DT, M, NN>), grid, dim3(256), 0, s, (const unsigned short*)x16, (const unsigned short*)w, bias, S, N, K, act, y, \
                                         (const int*)nullptr, (const int*)nullptr, 0LL)
#define LAUNCH_GEMM_N(DT, M) do { if (NT == 2) LAUNCH_GEMM(DT, M, 2); else LAUNCH_GEMM(DT, M, 1); } while (0)
#define LAUNCH_GEMM_M(DT) do { if (MT == 4) LAUNCH_GEMM_N(DT, 4); else if (MT == 3) LAUNCH_GEMM_N(DT, 3); else if (MT == 2) LAUNCH_GEMM_N(DT, 2); else LAUNCH_GEMM_N(DT, 1); } while (0)
  if (dt == BZ_F16) LAUNCH_GEMM_M(BZ_F16); else LAUNCH_GEMM_M(BZ_BF16);
#undef LAUNCH_GEMM_M
#undef LAUNCH_GEMM_N
#undef LAUNCH_GEMM
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

// Grouped form: G weight matrices [N][K] (w + g * w_stride elements), group g multiplies the rows [g_off[g], g_off[g] + g_cnt[g]) of x16 / y (device arrays);
// max_rows = the largest group's row count (host copy), total_rows for the FLOP accounting.  One launch for all the experts of a MoE layer.
int bzk_gemm_nt_grouped(hipStream_t s, int dt, const void* x16, const void* w, long long w_stride, int G, const int* g_off, const int* g_cnt, int max_rows, long long total_rows,
                        int N, int K, int act, float* y) {
  if (dt != BZ_F16 && dt != BZ_BF16) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt_grouped: 16-bit operands only");
  if (K % 64 || K < 64 || N <= 0 || G <= 0) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_nt_grouped: K=%d must be a positive multiple of 64", K);
  if (max_rows <= 0) return BZ_OK;
  static const bool old_kernel = getenv("BZ_GEMM_NT_WAVE_TILES") != nullptr;
  if (!old_kernel) {   // LDS-DMA kernel, one grid row per group; experts see tens of rows each, so the launch is bound by the weight stream, not by the padded row tiles
    const int mtiles = (max_rows + 63) / 64, ntiles = (N + 127) / 128;
    const dim3 grid2(8u * (unsigned)((ntiles + 7) / 8) * (unsigned)mtiles, G);
    const double flops2 = 2.0 * (double)total_rows * N * K;
#define LAUNCH_GG2(DT) do { \
      static bool attr_done = false; \
      if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_gemm_nt2<DT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr_done = true; } \
      BZ_LAUNCH("gemm_nt_mfma<grouped>", flops2, (k_gemm_nt2<DT, 1>), grid2, dim3(256), 2 * (64 + 128) * 128, s, (const unsigned short*)x16, (const unsigned short*)w, (const float*)nullptr, 0, N, K, act, y, \
                (float*)nullptr, 1, mtiles, ntiles, g_off, g_cnt, w_stride, (const float*)nullptr); } while (0)
    if (dt == BZ_F16) LAUNCH_GG2(BZ_F16); else LAUNCH_GG2(BZ_BF16);
#undef LAUNCH_GG2
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  const int MT = max_rows > 32 ? 2 : 1;        // experts see tens of rows each: small row tiles, the grid is filled by the groups
  const dim3 grid((N + 127) / 128, (max_rows + 32 * MT - 1) / (32 * MT), G);
  const double flops = 2.0 * (double)total_rows * N * K;
#define LAUNCH_GG(DT, M) BZ_LAUNCH("gemm_nt_mfma<grouped>", flops, (k_gemm_nt<DT, M, 1>), grid, dim3(256), 0, s, (const unsigned short*)x16, (const unsigned short*)w, (const float*)nullptr, \
                                   0, N, K, act, y, g_off, g_cnt, w_stride)
  if (dt == BZ_F16) { if (MT == 2) LAUNCH_GG(BZ_F16, 2); else LAUNCH_GG(BZ_F16, 1); }
  else { if (MT == 2) LAUNCH_GG(BZ_BF16, 2); else LAUNCH_GG(BZ_BF16, 1); }
#undef LAUNCH_GG
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}

bool bzk_gemm_q4g_mfma_ok(const LinearDev& L, int xdt, int rows) {
  static const bool off = getenv("BZ_NO_Q4G_MFMA") != nullptr;
  static const int min_rows = getenv("BZ_Q4G_MFMA_MIN") ? atoi(getenv("BZ_Q4G_MFMA_MIN")) : 5;   // measured on batched decode steps (8B AWQ, one graph per step): 5 / 6 / 8 sequences 4.46 / 4.77 / 5.41 ms
                                                                                                  // through the multi-row dot8 kernel, 4.19 / 4.23 / 4.26 ms here; 3-4 sequences equal or better there
  return !off && L.kind == LK_Q4G && !L.perm && L.K % 128 == 0 && L.N % 64 == 0 && xdt == BZ_F16 && rows >= min_rows;
}
// Y[S][N] (f32, rounded to act) = X16[S][K] . dequant(W)^T on the matrix cores (f16 activations)
int bzk_gemm_q4g_mfma(hipStream_t s, const LinearDev& L, const void* x16, int S, int act, float* y, float* ws, size_t ws_bytes) {
  if (!bzk_gemm_q4g_mfma_ok(L, BZ_F16, S)) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_q4g_mfma: unsupported weight / activation format");
  // one wave per block (measured faster than four at every prompt length, 32..2000 tokens): the A rows are shared through L1 / L2 either
  // way, and single-wave blocks spread a small grid (o_proj / down at a few hundred rows: 64 x 8 tiles) over the whole chip
  static const bool no_ks = getenv("BZ_Q4G_MFMA_NO_KSPLIT") != nullptr;
  const double flops = 2.0 * S * (double)L.N * L.K;
  static const bool no_lds = getenv("BZ_Q4G_MFMA_NO_LDS") != nullptr;
  static const int lds_min = getenv("BZ_Q4G_LDS_MIN") ? atoi(getenv("BZ_Q4G_LDS_MIN")) : 17;   // measured with the wide form (batched decode steps): 24 rows 4.49 -> 4.28 ms, 32 rows 4.66 -> 4.39 ms, 9-16 rows equal
  if (!no_lds && S >= lds_min) {                 // prompts: both operands through LDS (k_gemm_q4g_lds)
    const bool wide = S <= 64;
    const int mtiles = wide ? 1 : (S + 127) / 128, ntiles = wide ? (L.N + 255) / 256 : (L.N + 127) / 128, Gn = L.K / 128;
    const long long tiles2 = (long long)mtiles * ntiles;
    int KS2 = 1;
    if (!no_ks && ws) while (KS2 * 2 * tiles2 <= 320 && KS2 * 2 <= Gn / 2 && KS2 < 16 && (size_t)KS2 * 2 * S * L.N * 4 <= ws_bytes) KS2 *= 2;
    float* part2 = KS2 > 1 ? ws : nullptr;
    const unsigned grid2 = 8u * (unsigned)((ntiles + 7) / 8) * (unsigned)mtiles * (unsigned)KS2;
    if (wide) BZ_LAUNCH("gemm_q4g_mfma<lds 64x256>", flops, k_gemm_q4g_lds<true>, dim3(grid2), dim3(256), 3 * (8192 + 8192) + 2048, s, (const uint4*)L.w, (const __half*)L.scales,
                        (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const unsigned short*)x16, S, act, y, part2, KS2, mtiles, ntiles);
    else BZ_LAUNCH("gemm_q4g_mfma<lds>", flops, k_gemm_q4g_lds<false>, dim3(grid2), dim3(256), 3 * (16384 + 4096) + 2048, s, (const uint4*)L.w, (const __half*)L.scales,
                   (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const unsigned short*)x16, S, act, y, part2, KS2, mtiles, ntiles);
    BZ_HIP(hipGetLastError());
    if (KS2 > 1) {
      const size_t SN = (size_t)S * L.N;
      hipLaunchKernelGGL(k_q4g_mfma_reduce, dim3((unsigned)std::min<size_t>((SN + 255) / 256, 2048)), dim3(256), 0, s, (const float*)ws, KS2, SN, L.N, L.bias, act, y);
      BZ_HIP(hipGetLastError());
    }
    return BZ_OK;
  }
  const bool small = S <= 32;                    // decode batches / short prompts: one 32-row tile per wave
  const int G = L.K / 128, rt = small ? 1 : (S + 63) / 64;
  // short prompts / decode batches: too few 64 x 64 tiles to fill the chip -> split K over blockIdx.z into partials (summed in a fixed order)
  const long long tiles = (long long)(L.N / 64) * rt;
  int KS = 1, GPB = G;
  if (!no_ks && ws && tiles < 768) {
    const int want = (int)std::min<long long>(G, (1024 + tiles - 1) / tiles);
    GPB = (G + want - 1) / want;
    KS = (G + GPB - 1) / GPB;
    if (KS < 2 || (size_t)KS * S * L.N * 4 > ws_bytes) { KS = 1; GPB = G; }
  }
  float* part = KS > 1 ? ws : nullptr;
  {
    const dim3 grid(L.N / 64, rt, KS);
    if (small) BZ_LAUNCH("gemm_q4g_mfma<32 rows>", flops, (k_gemm_q4g_mfma<1, 1>), grid, dim3(64), 0, s, (const uint4*)L.w, (const __half*)L.scales,
                         (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const unsigned short*)x16, S, act, y, GPB, part);
    else BZ_LAUNCH("gemm_q4g_mfma", flops, (k_gemm_q4g_mfma<1, 2>), grid, dim3(64), 0, s, (const uint4*)L.w, (const __half*)L.scales,
                   (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const unsigned short*)x16, S, act, y, GPB, part);
  }
  BZ_HIP(hipGetLastError());
  if (KS > 1) {
    const size_t SN = (size_t)S * L.N;
    hipLaunchKernelGGL(k_q4g_mfma_reduce, dim3((unsigned)std::min<size_t>((SN + 255) / 256, 2048)), dim3(256), 0, s, (const float*)ws, KS, SN, L.N, L.bias, act, y);
    BZ_HIP(hipGetLastError());
  }
  return BZ_OK;
}

int bzk_ssm_scan_pieces(int head_dim) { return (head_dim + 15) / 16; }     // workgroups per head = vss pieces per (token, head)
bool bzk_ssm_scan_ok(int head_dim, int d_state, int n_groups, int kc) {
  return head_dim <= 64 && (d_state == 16 || d_state == 64 || d_state == 128) && n_groups <= 64 && kc >= 2 && kc <= 9;
}
int bzk_pf_conv(hipStream_t s, const float* zx, int ld, int x_off, int conv_dim, int kc, const float* w, const float* b, float* cs, int S, int act, float* out) {
  hipLaunchKernelGGL(k_pf_conv, dim3((conv_dim + 255) / 256, S), dim3(256), 0, s, zx, ld, x_off, conv_dim, kc, w, b, (const float*)cs, S, act, out);
  hipLaunchKernelGGL(k_pf_conv_state, dim3((conv_dim + 255) / 256), dim3(256), 0, s, zx, ld, x_off, conv_dim, kc, cs, S);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_ssm_scan(hipStream_t s, const BzSsmScan& b, int sdt) {
  if (!bzk_ssm_scan_ok(b.head_dim, b.d_state, b.n_groups, 4)) BZ_FAIL(BZ_E_UNSUPPORTED, "ssm_scan: unsupported state shape");
  SsmScanArgs a;
  a.xbc = b.xbc; a.conv_dim = b.conv_dim; a.zx = b.zx; a.ld = b.ld; a.dt_off = b.dt_off; a.dt_bias = b.dt_bias; a.A_log = b.A_log; a.D = b.D; a.state = b.state;
  a.n_heads = b.n_heads; a.head_dim = b.head_dim; a.d_state = b.d_state; a.n_groups = b.n_groups; a.d_inner = b.d_inner; a.act = b.act; a.S = b.S; a.y = b.y; a.vss = b.vss;
#define LAUNCH_SCAN(SDT, ACT, NQ) BZ_LAUNCH("mamba2_ssm_scan", 0.0, (k_ssm_scan<SDT, ACT, NQ>), dim3(b.n_heads, bzk_ssm_scan_pieces(b.head_dim)), dim3(256), 0, s, a)
#define LAUNCH_SCAN_N(SDT, ACT) do { if (b.d_state == 16) LAUNCH_SCAN(SDT, ACT, 1); else if (b.d_state == 64) LAUNCH_SCAN(SDT, ACT, 4); else LAUNCH_SCAN(SDT, ACT, 8); } while (0)
#define LAUNCH_SCAN_A(SDT) do { if (b.act == BZ_F32) LAUNCH_SCAN_N(SDT, BZ_F32); else if (b.act == BZ_F16) LAUNCH_SCAN_N(SDT, BZ_F16); else LAUNCH_SCAN_N(SDT, BZ_BF16); } while (0)
  if (sdt == BZ_F32) LAUNCH_SCAN_A(BZ_F32); else if (sdt == BZ_F16) LAUNCH_SCAN_A(BZ_F16); else LAUNCH_SCAN_A(BZ_BF16);
#undef LAUNCH_SCAN_A
#undef LAUNCH_SCAN_N
#undef LAUNCH_SCAN
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_gnorm(hipStream_t s, int dt, const float* v, const float* vss, const float* w, int S, int DI, int G, int NH, float eps, int act, void* x16) {
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_gnorm<BZ_F16>, dim3(S), dim3(256), 0, s, v, vss, w, DI, G, NH, eps, act, (unsigned short*)x16);
  else hipLaunchKernelGGL(k_pf_gnorm<BZ_BF16>, dim3(S), dim3(256), 0, s, v, vss, w, DI, G, NH, eps, act, (unsigned short*)x16);
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
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_norm<BZ_F16>, dim3(S), dim3(256), 0, s, hbuf, prev, w, H, eps, act, x16);
  else if (dt == BZ_F32) hipLaunchKernelGGL(k_pf_norm<BZ_F32>, dim3(S), dim3(256), 0, s, hbuf, prev, w, H, eps, act, x16);
  else hipLaunchKernelGGL(k_pf_norm<BZ_BF16>, dim3(S), dim3(256), 0, s, hbuf, prev, w, H, eps, act, x16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_rope_kv(hipStream_t s, float* qkv, int S, int nq, int nkv, int hd, const float* cos_t, const float* sin_t, int interleaved, int pos0, int act,
                   const KvView& kv, int layer, const int* slots, const int* row_pos) {
  hipLaunchKernelGGL(k_pf_rope_kv, dim3(S, nq + 2 * nkv), dim3(64), 0, s, qkv, nq, nkv, hd, cos_t, sin_t, interleaved, pos0, act, kv, layer, slots, row_pos);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
bool bzk_pf_attn_mfma_ok(int hd, int rep) {     // prompts: the MFMA flash kernel (any context length)
  static const bool off = getenv("BZ_NO_PF_ATTN_MFMA") != nullptr;
  return !off && (hd == 64 || hd == 128) && (rep == 1 || rep == 2 || rep == 4 || rep == 8);
}
size_t bzk_pf_attn_smem(int nq, int nkv, int hd, int len, bool exact) {
  const int REP = nq / nkv, RPP = 256 / (hd / 8);
  return (size_t)(8 * REP + RPP * REP * hd) * (exact ? 8 : 4) + (size_t)REP * len * 4 + 64;
}
int bzk_pf_attn(hipStream_t s, int dt, const float* qkv, int S, int nq, int nkv, int hd, int pos0, int act, const KvView& kv, int layer, void* out16,
                const int* row_pos, int table_stride, int max_len, bool exact) {
  const int REP = nq / nkv;
  if (hd % 8 || hd > 256 || (256 % (hd / 8)) || (REP != 1 && REP != 2 && REP != 4 && REP != 8) || kv.dtype != dt)
    BZ_FAIL(BZ_E_UNSUPPORTED, "prefill attention: head_dim %d / group size %d / cache dtype unsupported", hd, REP);
  if (!row_pos && dt != BZ_F32 && !exact && bzk_pf_attn_mfma_ok(hd, REP)) {
    const float scale_m = div_rn(1.0f, sqrt_rn((float)hd));
    const int HW = REP < 4 ? REP : 4, QT = 4 / HW;
    const dim3 grid((S + 32 * QT - 1) / (32 * QT), nkv, REP / HW);
    const double flops = 4.0 * nq * hd * ((double)S * pos0 + 0.5 * (double)S * S);
#define LAUNCH_PFM(DT, HD, R) do { if (kv.paged) BZ_LAUNCH("pf_attn_mfma", flops, (k_pf_attn_mfma<DT, HD, R, true>), grid, dim3(256), 0, s, qkv, S, nq, nkv, pos0, act, kv, layer, scale_m, (unsigned short*)out16); \
                                   else BZ_LAUNCH("pf_attn_mfma", flops, (k_pf_attn_mfma<DT, HD, R, false>), grid, dim3(256), 0, s, qkv, S, nq, nkv, pos0, act, kv, layer, scale_m, (unsigned short*)out16); } while (0)
#define LAUNCH_PFM_R(DT, HD) do { if (REP == 1) LAUNCH_PFM(DT, HD, 1); else if (REP == 2) LAUNCH_PFM(DT, HD, 2); else if (REP == 4) LAUNCH_PFM(DT, HD, 4); else LAUNCH_PFM(DT, HD, 8); } while (0)
#define LAUNCH_PFM_H(DT) do { if (hd == 64) LAUNCH_PFM_R(DT, 64); else LAUNCH_PFM_R(DT, 128); } while (0)
    if (dt == BZ_F16) LAUNCH_PFM_H(BZ_F16); else LAUNCH_PFM_H(BZ_BF16);
#undef LAUNCH_PFM_H
#undef LAUNCH_PFM_R
#undef LAUNCH_PFM
    BZ_HIP(hipGetLastError());
    return BZ_OK;
  }
  const int ctx = row_pos ? max_len : pos0 + S;
  const size_t smem = bzk_pf_attn_smem(nq, nkv, hd, ctx, exact);
  if (smem > 160 * 1024) BZ_FAIL(BZ_E_UNSUPPORTED, "prefill attention: context %d too long for this kernel", ctx);
  const float scale = div_rn(1.0f, sqrt_rn((float)hd));
#define LAUNCH_PFA(DT, R, EX) do { \
    static bool attr_done = false; \
    if (!attr_done) { BZ_HIP(hipFuncSetAttribute((const void*)k_pf_attn<DT, DT, R, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; } \
    hipLaunchKernelGGL((k_pf_attn<DT, DT, R, EX>), dim3(S, nkv), dim3(256), smem, s, qkv, nq, nkv, hd, pos0, act, kv, layer, scale, out16, row_pos, table_stride); } while (0)
#define LAUNCH_PFA_R(DT, EX) do { if (REP == 1) LAUNCH_PFA(DT, 1, EX); else if (REP == 2) LAUNCH_PFA(DT, 2, EX); else if (REP == 4) LAUNCH_PFA(DT, 4, EX); else LAUNCH_PFA(DT, 8, EX); } while (0)
  if (exact) { if (dt == BZ_F16) LAUNCH_PFA_R(BZ_F16, true); else if (dt == BZ_F32) LAUNCH_PFA_R(BZ_F32, true); else LAUNCH_PFA_R(BZ_BF16, true); }
  else { if (dt == BZ_F16) LAUNCH_PFA_R(BZ_F16, false); else if (dt == BZ_F32) LAUNCH_PFA_R(BZ_F32, false); else LAUNCH_PFA_R(BZ_BF16, false); }
#undef LAUNCH_PFA_R
#undef LAUNCH_PFA
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
bool bzk_gemm_q4g_i8_ok(const LinearDev& L, int xdt) {
  return L.kind == LK_Q4G && !L.perm && L.K % 128 == 0 && L.N % 64 == 0 && xdt == BZ_F16;
}
size_t bzk_pf_quant_i8_bytes(int S, int K, size_t* par_off) { const size_t planes = (size_t)4 * S * K; *par_off = (planes + 255) / 256 * 256; return *par_off + (size_t)S * (K / 128) * sizeof(RowPar); }
// x16 rows -> four int8 digit planes + row parameters in `xq` (bzk_pf_quant_i8_bytes(S, K) bytes)
int bzk_pf_quant_i8(hipStream_t s, const void* x16, int S, int K, void* xq) {
  if (K % 128) BZ_FAIL(BZ_E_UNSUPPORTED, "pf_quant_i8: K=%d is not a multiple of 128", K);
  size_t po; bzk_pf_quant_i8_bytes(S, K, &po);
  hipLaunchKernelGGL(k_pf_quant_i8<BZ_F16>, dim3(S), dim3(256), 0, s, (const unsigned short*)x16, K, (signed char*)xq, (size_t)S * K, (RowPar*)((char*)xq + po));
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
// Y[S][N] = R(X . dequant(W)^T), every group sum an exact integer (xq from bzk_pf_quant_i8 for the same S, K)
int bzk_gemm_q4g_i8(hipStream_t s, const LinearDev& L, const void* xq, int S, int act, float* y) {
  if (!bzk_gemm_q4g_i8_ok(L, BZ_F16)) BZ_FAIL(BZ_E_UNSUPPORTED, "gemm_q4g_i8: unsupported weight format");
  size_t po; bzk_pf_quant_i8_bytes(S, L.K, &po);
  const int ntile = L.N / 64, mt = (S + 31) / 32;
  static const int wpb_env = getenv("BZ_I8_WPB") ? atoi(getenv("BZ_I8_WPB")) : 0;
  int WPB = (long long)((ntile + 3) / 4) * mt >= 256 ? 4 : 2;      // a few hundred workgroups when the problem has them
  if ((long long)((ntile + 7) / 8) * mt >= 512) WPB = 8;          // large problems: 32 x 512 tiles halve the activation re-reads from L2
  if (wpb_env) WPB = wpb_env;
  const double flops = 2.0 * S * (double)L.N * L.K;
#define LAUNCH_I8X(W_) BZ_LAUNCH("gemm_q4g_i8_mfma", flops, (k_gemm_q4g_i8<W_, 0, false>), dim3(8u * (unsigned)(((ntile + W_ - 1) / W_ + 7) / 8) * (unsigned)mt), dim3(W_ * 64), 0, s, (const uint4*)L.w, (const __half*)L.scales, \
              (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const signed char*)xq, (size_t)S * L.K, (const RowPar*)((const char*)xq + po), S, act, y)
#define LAUNCH_I8D(W_, D_) BZ_LAUNCH("gemm_q4g_i8_mfma", flops, (k_gemm_q4g_i8<W_, D_>), dim3(8u * (unsigned)(((ntile + W_ - 1) / W_ + 7) / 8) * (unsigned)mt), dim3(W_ * 64), 0, s, (const uint4*)L.w, (const __half*)L.scales, \
              (const unsigned char*)L.zeros, L.bias, L.N, L.K, (const signed char*)xq, (size_t)S * L.K, (const RowPar*)((const char*)xq + po), S, act, y)
#define LAUNCH_I8(W_) do { if (dbg == 1) LAUNCH_I8D(W_, 1); else if (dbg == 2) LAUNCH_I8D(W_, 2); else if (dbg == 3) LAUNCH_I8D(W_, 3); else if (dbg == 4) LAUNCH_I8X(W_); else LAUNCH_I8D(W_, 0); } while (0)
  static const int dbg = getenv("BZ_I8_DBG") ? atoi(getenv("BZ_I8_DBG")) : 0;     // timing experiments only: the results are wrong
  if (WPB == 8) LAUNCH_I8(8); else if (WPB == 4) LAUNCH_I8(4); else LAUNCH_I8(2);
#undef LAUNCH_I8D
#undef LAUNCH_I8
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_silu(hipStream_t s, int dt, const float* gu, int S, int I, int act, void* a16) {
  if (I % 4) BZ_FAIL(BZ_E_UNSUPPORTED, "pf_silu: intermediate size %d is not a multiple of 4", I);
  const dim3 grid((I + 1023) / 1024, S);
  if (dt == BZ_F16) hipLaunchKernelGGL(k_pf_silu<BZ_F16>, grid, dim3(256), 0, s, gu, S, I, act, a16);
  else if (dt == BZ_F32) hipLaunchKernelGGL(k_pf_silu<BZ_F32>, grid, dim3(256), 0, s, gu, S, I, act, a16);
  else hipLaunchKernelGGL(k_pf_silu<BZ_BF16>, grid, dim3(256), 0, s, gu, S, I, act, a16);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
int bzk_pf_split3(hipStream_t s, const float* x, int S, int K, void* xs, float* rscale, const float* wscale) {
  hipLaunchKernelGGL(k_pf_split3, dim3(S), dim3(256), 0, s, x, K, (unsigned short*)xs, rscale, wscale);
  BZ_HIP(hipGetLastError());
  return BZ_OK;
}
