/*
 * oracle/orc_quant.c -- number formats and weight-format restatements (TEST INFRASTRUCTURE; see orc.h).
 *
 * Reference anchors:
 *   AWQ  : /root/reference/src/loader/safetensors/awq.rs:3-6 (shapes), :29-32 (AWQ_SHIFTS), :239-263 (unpack_awq_zeros)
 *   GPTQ : /root/reference/src/loader/safetensors/gptq.rs:3-8 (shapes), :198-247 (what is kept packed)
 *   GGUF : /root/reference/src/loader/gguf.rs:20-44 (blocks are opaque to blazr; formats = public GGML spec)
 * The dequant FORMULAS are ASSUMPTIONS restating AutoAWQ / AutoGPTQ-v1 / GGML (the reference
 * delegates them to boostr, which is absent): parity unpinned.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------- f16 / bf16 */
float orc_f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      man &= 0x3FFu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f; memcpy(&f, &bits, 4); return f;
}

uint16_t orc_f32_to_f16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) /* inf / nan */
    return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0));
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* rounds to inf (>= 65520) */
  if (ax < 0x33000001u) return (uint16_t)sign;             /* < 2^-25 (or == 2^-25: ties-to-even -> 0) */
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
  uint32_t shift, half_bits;
  if (e < -14) { /* subnormal result */
    shift = (uint32_t)(13 + (-14 - e));
    half_bits = 0;
  } else {
    shift = 13;
    half_bits = (uint32_t)(e + 15) << 10;
    m &= 0x7FFFFFu;
  }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  return (uint16_t)(sign | (half_bits + q)); /* carry into exponent is correct by construction */
}

float orc_bf16_to_f32(uint16_t h) {
  uint32_t bits = (uint32_t)h << 16; float f; memcpy(&f, &bits, 4); return f;
}

uint16_t orc_f32_to_bf16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x40u); /* keep NaN a NaN */
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7FFFu + lsb;
  return (uint16_t)(x >> 16);
}

float orc_round(float x, int dtype) {
  if (dtype == ORC_F16) return orc_f16_to_f32(orc_f32_to_f16(x));
  if (dtype == ORC_BF16) return orc_bf16_to_f32(orc_f32_to_bf16(x));
  return x;
}

void orc_round_vec(float* x, size_t n, int dtype) {
  if (dtype == ORC_F32) return;
  for (size_t i = 0; i < n; i++) x[i] = orc_round(x[i], dtype);
}

/* ---------------------------------------------------------------- AWQ */
/* awq.rs:29-32 */
static const uint32_t AWQ_SHIFTS[8] = {0, 16, 4, 20, 8, 24, 12, 28};

/* awq.rs:239-263 */
void orc_awq_unpack_zeros(const uint32_t* packed, int G, int N, float* out) {
  int n8 = N / 8;
  for (int g = 0; g < G; g++)
    for (int j = 0; j < n8; j++) {
      uint32_t pv = packed[(size_t)g * n8 + j];
      for (int k = 0; k < 8; k++) out[(size_t)g * N + j * 8 + k] = (float)((pv >> AWQ_SHIFTS[k]) & 0xFu);
    }
}

static inline float awq_w(const orc_linear* L, int k, int n) {
  const uint32_t* qw = (const uint32_t*)L->w;
  int g = k / L->group_size;
  uint32_t word = qw[(size_t)k * (L->N / 8) + n / 8];
  float q = (float)((word >> AWQ_SHIFTS[n % 8]) & 0xFu);
  /* ASSUMPTION (AutoAWQ GEMM): w = (q - zero) * scale */
  return (q - L->zeros_f[(size_t)g * L->N + n]) * L->scales[(size_t)g * L->N + n];
}

/* ---------------------------------------------------------------- GPTQ */
static inline int gptq_group(const orc_linear* L, int k) { return L->g_idx ? L->g_idx[k] : k / L->group_size; }

static inline float gptq_w(const orc_linear* L, int k, int n) {
  const uint32_t* qw = (const uint32_t*)L->w;
  int g = gptq_group(L, k);
  uint32_t word = qw[(size_t)(k / 8) * L->N + n];
  float q = (float)((word >> (4 * (k % 8))) & 0xFu);
  uint32_t zw = L->qzeros[(size_t)g * (L->N / 8) + n / 8];
  /* ASSUMPTION (AutoGPTQ v1 checkpoints): stored zero is (zero - 1) */
  float z = (float)(((zw >> (4 * (n % 8))) & 0xFu) + 1u);
  return (q - z) * L->scales[(size_t)g * L->N + n];
}

/* ---------------------------------------------------------------- GGML blocks (public spec) */
#define QK8_0 32
#define QK_K 256
typedef struct { uint16_t d; int8_t qs[QK8_0]; } __attribute__((packed)) blk_q8_0;                       /* 34 B  */
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qs[QK_K / 2]; } __attribute__((packed)) blk_q4_K; /* 144 B */
typedef struct { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; uint16_t d; } __attribute__((packed)) blk_q6_K; /* 210 B */

size_t orc_ggml_row_bytes(int type, size_t K) {
  switch (type) {
    case ORC_GGML_F32: return K * 4;
    case ORC_GGML_F16: case ORC_GGML_BF16: return K * 2;
    case ORC_GGML_Q8_0: return K / QK8_0 * sizeof(blk_q8_0);
    case ORC_GGML_Q4_K: return K / QK_K * sizeof(blk_q4_K);
    case ORC_GGML_Q6_K: return K / QK_K * sizeof(blk_q6_K);
    default: return 0;
  }
}

static inline void q4k_scale_min(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
  if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
  else { *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}

void orc_ggml_dequant(int type, const void* blocks, size_t n, float* y) {
  if (type == ORC_GGML_F32) { memcpy(y, blocks, n * 4); return; }
  if (type == ORC_GGML_F16) { const uint16_t* h = blocks; for (size_t i = 0; i < n; i++) y[i] = orc_f16_to_f32(h[i]); return; }
  if (type == ORC_GGML_BF16) { const uint16_t* h = blocks; for (size_t i = 0; i < n; i++) y[i] = orc_bf16_to_f32(h[i]); return; }
  if (type == ORC_GGML_Q8_0) {
    const blk_q8_0* b = blocks;
    for (size_t i = 0; i < n / QK8_0; i++) {
      float d = orc_f16_to_f32(b[i].d);
      for (int j = 0; j < QK8_0; j++) y[i * QK8_0 + j] = d * (float)b[i].qs[j];
    }
    return;
  }
  if (type == ORC_GGML_Q4_K) {
    const blk_q4_K* b = blocks;
    for (size_t i = 0; i < n / QK_K; i++) {
      float d = orc_f16_to_f32(b[i].d), dmin = orc_f16_to_f32(b[i].dmin);
      const uint8_t* q = b[i].qs; float* yy = y + i * QK_K; int is = 0;
      for (int j = 0; j < QK_K; j += 64) {
        uint8_t sc, m;
        q4k_scale_min(is + 0, b[i].scales, &sc, &m); float d1 = d * sc, m1 = dmin * m;
        q4k_scale_min(is + 1, b[i].scales, &sc, &m); float d2 = d * sc, m2 = dmin * m;
        for (int l = 0; l < 32; l++) *yy++ = d1 * (float)(q[l] & 0xF) - m1;
        for (int l = 0; l < 32; l++) *yy++ = d2 * (float)(q[l] >> 4) - m2;
        q += 32; is += 2;
      }
    }
    return;
  }
  if (type == ORC_GGML_Q6_K) {
    const blk_q6_K* b = blocks;
    for (size_t i = 0; i < n / QK_K; i++) {
      float d = orc_f16_to_f32(b[i].d);
      const uint8_t* ql = b[i].ql; const uint8_t* qh = b[i].qh; const int8_t* sc = b[i].scales; float* yy = y + i * QK_K;
      for (int nn = 0; nn < QK_K; nn += 128) {
        for (int l = 0; l < 32; l++) {
          int is = l / 16;
          int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
          int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
          int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
          int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
          yy[l + 0] = d * (float)sc[is + 0] * (float)q1;
          yy[l + 32] = d * (float)sc[is + 2] * (float)q2;
          yy[l + 64] = d * (float)sc[is + 4] * (float)q3;
          yy[l + 96] = d * (float)sc[is + 6] * (float)q4;
        }
        yy += 128; ql += 64; qh += 32; sc += 8;
      }
    }
    return;
  }
  memset(y, 0, n * sizeof(float));
}

/* ---------------------------------------------------------------- whole-matrix dequant [N][K] */
static inline float dense_at(const orc_linear* L, size_t idx) {
  if (L->w_dtype == ORC_F32) return ((const float*)L->w)[idx];
  if (L->w_dtype == ORC_F16) return orc_f16_to_f32(((const uint16_t*)L->w)[idx]);
  return orc_bf16_to_f32(((const uint16_t*)L->w)[idx]);
}

void orc_linear_dequant(const orc_linear* L, float* out) {
  int N = L->N, K = L->K;
  if (L->kind == ORC_LIN_GGUF) {
    size_t rb = orc_ggml_row_bytes(L->ggml_type, (size_t)K);
    for (int n = 0; n < N; n++) orc_ggml_dequant(L->ggml_type, (const char*)L->w + (size_t)n * rb, (size_t)K, out + (size_t)n * K);
    return;
  }
  for (int n = 0; n < N; n++)
    for (int k = 0; k < K; k++) {
      float w;
      if (L->kind == ORC_LIN_DENSE) w = dense_at(L, (size_t)n * K + k);
      else if (L->kind == ORC_LIN_AWQ) w = awq_w(L, k, n);
      else w = gptq_w(L, k, n);
      out[(size_t)n * K + k] = w;
    }
}

/* ---------------------------------------------------------------- y = x W^T (+bias) */
/* Summation (the oracle's definition): the EXACTLY ROUNDED dot product.  Every product x[k] * w[k] of two f32 values is exact in double
 * (24 + 24 significant bits), the sum over k is carried in double (relative error ~1e-16 K, nine orders of magnitude under an f16 ulp),
 * and the result is rounded to f32 once.  This is the one definition that does not depend on a summation order, so it is what "f32
 * accumulate" implementations with different orders (boostr's CPU and CUDA kernels, the HIP path) all approximate; round 1's f32-blocked
 * sums carried ~1e-6 of order-dependent error of their own, which flipped 0.3 % of the f16 roundings downstream and put the comparison at
 * its noise floor (tests/test_gpu_parity_truth.py).  The HIP int4 path computes the same quantity exactly (integer group sums, double
 * across groups, 2^-32 fixed point across workgroups).
 *   AWQ / GPTQ : w[k] = (q - z) * s, exact in f32 (5-bit integer times an f16 scale)                                              */
static inline float dot8(const float* a, const float* b, int K) {
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = 0;
  for (; k + 8 <= K; k += 8)
    for (int j = 0; j < 8; j++) p[j] += (double)a[k + j] * (double)b[k + j];
  for (int j = 0; k < K; k++, j++) p[j] += (double)a[k] * (double)b[k];
  return (float)(((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7])));
}

#define NB 64 /* columns per work item */
#define PAR_MIN ((size_t)1 << 21) /* weights below which a GEMV stays on one thread */

static void awq_forward(const orc_linear* L, const float* x, int S, float* y) {
  const int N = L->N, K = L->K, gs = L->group_size, n8 = N / 8;
  const uint32_t* qw = (const uint32_t*)L->w;
#pragma omp parallel for schedule(static) if ((size_t)N * K >= PAR_MIN)
  for (int nb = 0; nb < N / NB; nb++) {
    float wrow[NB]; double tot[8][NB];
    const int n0 = nb * NB;
    for (int s0 = 0; s0 < S; s0 += 8) {
      const int sc = (S - s0) < 8 ? (S - s0) : 8;
      memset(tot, 0, sizeof(tot));
      for (int k = 0; k < K; k++) {
        const float* srow = L->scales + (size_t)(k / gs) * N + n0;
        const float* zrow = L->zeros_f + (size_t)(k / gs) * N + n0;
        const uint32_t* wp = qw + (size_t)k * n8 + n0 / 8;
        for (int j = 0; j < NB / 8; j++) {
          uint32_t word = wp[j];
          for (int i = 0; i < 8; i++) {
            float q = (float)((word >> AWQ_SHIFTS[i]) & 0xFu);
            wrow[j * 8 + i] = (q - zrow[j * 8 + i]) * srow[j * 8 + i];
          }
        }
        for (int s = 0; s < sc; s++) {
          const double xv = (double)x[(size_t)(s0 + s) * K + k];
          for (int c = 0; c < NB; c++) tot[s][c] += xv * (double)wrow[c];
        }
      }
      for (int s = 0; s < sc; s++)
        for (int c = 0; c < NB; c++) y[(size_t)(s0 + s) * N + n0 + c] = (float)tot[s][c] + (L->bias ? L->bias[n0 + c] : 0.0f);
    }
  }
}

static void gptq_forward(const orc_linear* L, const float* x, int S, float* y) {
  const int N = L->N, K = L->K;
  const uint32_t* qw = (const uint32_t*)L->w;
#pragma omp parallel for schedule(static) if ((size_t)N * K >= PAR_MIN)
  for (int nb = 0; nb < N / NB; nb++) {
    float wrow[NB]; double tot[8][NB];
    const int n0 = nb * NB;
    for (int s0 = 0; s0 < S; s0 += 8) {
      const int sc = (S - s0) < 8 ? (S - s0) : 8;
      memset(tot, 0, sizeof(tot));
      for (int k = 0; k < K; k++) {
        const int g = gptq_group(L, k);
        const float* srow = L->scales + (size_t)g * N + n0;
        const uint32_t* zp = L->qzeros + (size_t)g * (N / 8) + n0 / 8;
        const uint32_t* wp = qw + (size_t)(k / 8) * N + n0;
        const int sh = 4 * (k % 8);
        for (int c = 0; c < NB; c++) {
          float q = (float)((wp[c] >> sh) & 0xFu);
          float z = (float)(((zp[c / 8] >> (4 * (c % 8))) & 0xFu) + 1u);
          wrow[c] = (q - z) * srow[c];
        }
        for (int s = 0; s < sc; s++) {
          const double xv = (double)x[(size_t)(s0 + s) * K + k];
          for (int c = 0; c < NB; c++) tot[s][c] += xv * (double)wrow[c];
        }
      }
      for (int s = 0; s < sc; s++)
        for (int c = 0; c < NB; c++) y[(size_t)(s0 + s) * N + n0 + c] = (float)tot[s][c] + (L->bias ? L->bias[n0 + c] : 0.0f);
    }
  }
}

static void rows_forward(const orc_linear* L, const float* x, int S, float* y) {
  const int N = L->N, K = L->K;
  const size_t rb = (L->kind == ORC_LIN_GGUF) ? orc_ggml_row_bytes(L->ggml_type, (size_t)K) : 0;
#pragma omp parallel if ((size_t)N * K >= PAR_MIN)
  {
    float* wrow = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
    for (int n = 0; n < N; n++) {
      const float* wr;
      if (L->kind == ORC_LIN_GGUF) { orc_ggml_dequant(L->ggml_type, (const char*)L->w + (size_t)n * rb, (size_t)K, wrow); wr = wrow; }
      else if (L->w_dtype == ORC_F32) wr = (const float*)L->w + (size_t)n * K;
      else { for (int k = 0; k < K; k++) wrow[k] = dense_at(L, (size_t)n * K + k); wr = wrow; }
      for (int s = 0; s < S; s++) y[(size_t)s * N + n] = dot8(x + (size_t)s * K, wr, K) + (L->bias ? L->bias[n] : 0.0f);
    }
    free(wrow);
  }
}

void orc_linear_forward(const orc_linear* L, const float* x, int S, float* y) {
  if (L->kind == ORC_LIN_AWQ && L->N % NB == 0) awq_forward(L, x, S, y);
  else if (L->kind == ORC_LIN_GPTQ && L->N % NB == 0) gptq_forward(L, x, S, y);
  else if (L->kind == ORC_LIN_DENSE || L->kind == ORC_LIN_GGUF) rows_forward(L, x, S, y);
  else { /* ragged N: slow generic path, same definition */
    for (int s = 0; s < S; s++)
      for (int n = 0; n < L->N; n++) {
        double tot = 0.0;
        for (int k = 0; k < L->K; k++) tot += (double)x[(size_t)s * L->K + k] * (double)(L->kind == ORC_LIN_AWQ ? awq_w(L, k, n) : gptq_w(L, k, n));
        y[(size_t)s * L->N + n] = (float)tot + (L->bias ? L->bias[n] : 0.0f);
      }
  }
}
