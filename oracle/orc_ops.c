/*
 * oracle/orc_ops.c -- elementwise / attention / sampling restatements (TEST INFRASTRUCTURE; see orc.h).
 *
 * Reference anchors (call sites; bodies are in the absent boostr crate => formulas are ASSUMPTIONS
 * restating HF transformers Llama semantics):
 *   RMSNorm  : NormalizationOps bound, /root/reference/src/engine/executor.rs:72 ; eps default gguf.rs:157-160
 *   RoPE     : rope_caches() [max_pos, head_dim/2] f32, /root/reference/src/engine/cuda_graphs.rs:81-93 ;
 *              scaling fields /root/reference/src/loader/safetensors/config.rs:83-95
 *   sampling : /root/reference/src/engine/sampling.rs:169-191 (window), :445-460 (logits_to_token args)
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* HF LlamaRMSNorm: y = w * round(x * rsqrt(mean(x^2) + eps)); both products rounded to the activation dtype */
void orc_rms_norm(const float* x, const float* w, int n, float eps, int act, float* out) {
  double ssd = 0.0;   /* exact-ish sum of squares, one rounding */
  for (int i = 0; i < n; i++) ssd += (double)(x[i] * x[i]);
  float ss = (float)ssd;
  float rs = 1.0f / sqrtf(ss / (float)n + eps);
  for (int i = 0; i < n; i++) out[i] = orc_round(w[i] * orc_round(x[i] * rs, act), act);
}

void orc_rope_tables(const orc_rope_cfg* c, float* cos_t, float* sin_t) {
  const int half = c->head_dim / 2;
  const double PI2 = 6.283185307179586476925286766559;
  /* yarn: HF modeling_rope_utils._compute_yarn_parameters (the reference carries beta_fast / beta_slow / attention_factor as None = defaults,
     loader/safetensors/config.rs:83-95): correction range [low, high] in dims, linear ramp, blend of interpolated and original inv_freq */
  double ylow = 0.0, yhigh = 0.0; float af = 1.0f;
  if (c->scaling_type == 3) {
    const double bf = c->beta_fast > 0.f ? c->beta_fast : 32.0, bs = c->beta_slow > 0.f ? c->beta_slow : 1.0;
    const double dim = c->head_dim, lb = log((double)c->theta), omax = c->original_max_pos;
    ylow = floor(dim * log(omax / (bf * PI2)) / (2.0 * lb));
    yhigh = ceil(dim * log(omax / (bs * PI2)) / (2.0 * lb));
    if (ylow < 0.0) ylow = 0.0;
    if (yhigh > dim - 1.0) yhigh = dim - 1.0;
    if (ylow == yhigh) yhigh += 0.001;
    af = c->attn_factor > 0.f ? c->attn_factor : (c->factor <= 1.f ? 1.0f : (float)(0.1 * log((double)c->factor) + 1.0));
  }
  for (int i = 0; i < half; i++) {
    double inv = 1.0 / pow((double)c->theta, (double)(2 * i) / (double)c->head_dim);
    if (c->scaling_type == 3) {
      double ramp = ((double)i - ylow) / (yhigh - ylow);
      if (ramp < 0.0) ramp = 0.0;
      if (ramp > 1.0) ramp = 1.0;
      const double ext = 1.0 - ramp;                      /* weight of the original (extrapolated) frequency */
      inv = (inv / (double)c->factor) * (1.0 - ext) + inv * ext;
    }
    if (c->scaling_type == 1) {
      inv /= (double)c->factor;
    } else if (c->scaling_type == 2) { /* llama3 (config.rs:83-95 field mapping; HF _compute_llama3_parameters) */
      double low_wl = (double)c->original_max_pos / (double)c->low_freq_factor;
      double high_wl = (double)c->original_max_pos / (double)c->high_freq_factor;
      double wl = PI2 / inv;
      if (wl > low_wl) inv = inv / (double)c->factor;
      else if (wl >= high_wl) {
        double smooth = ((double)c->original_max_pos / wl - (double)c->low_freq_factor) /
                        ((double)c->high_freq_factor - (double)c->low_freq_factor);
        inv = (1.0 - smooth) * inv / (double)c->factor + smooth * inv;
      }
    }
    const float invf = (float)inv;
    for (int p = 0; p < c->max_pos; p++) {
      float ang = (float)p * invf;
      cos_t[(size_t)p * half + i] = (float)cos((double)ang) * af;
      sin_t[(size_t)p * half + i] = (float)sin((double)ang) * af;
    }
  }
}

/* RoPE of one head.  The products x * cos are exact in double (x carries <= 24 significant bits, the table entries 24), so each output is the
   double-rounded difference / sum of two exact products, rounded once more to f32: the same bits whatever the compiler contracts. */
void orc_rope_apply(float* v, int head_dim, int rot_dim, const float* c, const float* s, int interleaved) {
  const int half = rot_dim / 2;
  (void)head_dim;
  for (int i = 0; i < half; i++) {
    int a = interleaved ? 2 * i : i, b = interleaved ? 2 * i + 1 : i + half;
    const double x0 = v[a], x1 = v[b], ci = c[i], si = s[i];
    v[a] = (float)(x0 * ci - x1 * si);
    v[b] = (float)(x1 * ci + x0 * si);
  }
}

/* exp, SPECIFIED (ASSUMPTION: the reference's expf lives in boostr / libm and is not visible; any faithful expf is "the" exp).  The oracle and
   the HIP kernels evaluate this one sequence of IEEE operations (Cephes expf: Cody-Waite reduction by ln2 in two fmaf steps, degree-5 polynomial
   in Horner form with fmaf, exact scaling by 2^n), so a comparison of the two sides measures structure and rounding points, not two libms.
   Range: 0 below -86 (the smallest result stays a normal f32), +inf above 88; NaN propagates.  Max error vs the true exp: < 1 ulp. */
float orc_expf(float x) {
  if (x != x) return x;
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return INFINITY;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693145751953125f, x);
  r = fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  float y = fmaf(p, r2, r);
  y = y + 1.0f;
  union { uint32_t u; float f; } sc;
  sc.u = (uint32_t)((int)n + 127) << 23;      /* 2^n, n in [-125, 127] */
  return y * sc.f;
}

/* One kv head's group of query heads over `len` cached positions.  Every sum is DEFINED as the exactly rounded sum (the linear layers' rule,
   orc_quant.c): products of 16-bit values are exact in double, so are products of an f32 probability and a 16-bit value; the sums are carried in
   double (order-independent to ~1e-16) and rounded to f32 once.  score = f32(sum q k) * scale; p = exp(score - max) (orc_expf);
   out = f32(sum p v) / f32(sum p).  An implementation that accumulates in f32 in some order approximates exactly this. */
void orc_attn_decode(const float* q, int n_q_per_kv, int head_dim, const float* kc, const float* vc,
                     size_t stride, int len, float scale, float* out) {
  float* sc = (float*)malloc(sizeof(float) * (size_t)len);
  double* od = (double*)malloc(sizeof(double) * (size_t)head_dim);
  for (int h = 0; h < n_q_per_kv; h++) {
    const float* qh = q + (size_t)h * head_dim;
    float m = -INFINITY;
    for (int p = 0; p < len; p++) {
      const float* kr = kc + (size_t)p * stride;
      double d = 0.0;
      for (int i = 0; i < head_dim; i++) d += (double)qh[i] * (double)kr[i];
      sc[p] = (float)d * scale;
      if (sc[p] > m) m = sc[p];
    }
    double sum = 0.0;
    for (int p = 0; p < len; p++) { sc[p] = orc_expf(sc[p] - m); sum += (double)sc[p]; }
    const float l = (float)sum;
    float* o = out + (size_t)h * head_dim;
    for (int i = 0; i < head_dim; i++) od[i] = 0.0;
    for (int p = 0; p < len; p++) {
      const float* vr = vc + (size_t)p * stride;
      const double e = (double)sc[p];
      for (int i = 0; i < head_dim; i++) od[i] += e * (double)vr[i];
    }
    for (int i = 0; i < head_dim; i++) o[i] = (float)od[i] / l;
  }
  free(sc); free(od);
}

float orc_silu(float x) { return x / (1.0f + orc_expf(-x)); }

int64_t orc_argmax(const float* v, int64_t n) {
  int64_t best = 0; float bv = v[0];
  for (int64_t i = 1; i < n; i++) if (v[i] > bv) { bv = v[i]; best = i; }
  return best;
}

/* sampling.rs:169-191 */
int orc_penalty_window(const uint32_t* recent, int n_recent, int repeat_last_n, int64_t* ids, int32_t* cnts) {
  const uint32_t* w = recent; int wn = n_recent;
  if (repeat_last_n > 0 && repeat_last_n < n_recent) { w = recent + (n_recent - repeat_last_n); wn = repeat_last_n; }
  int n = 0;
  for (int i = 0; i < wn; i++) {
    int j = 0;
    for (; j < n; j++) if (ids[j] == (int64_t)w[i]) { cnts[j]++; break; }
    if (j == n) { ids[n] = (int64_t)w[i]; cnts[n] = 1; n++; }
  }
  /* ascending id order (any order is equivalent: ids are unique) */
  for (int i = 1; i < n; i++) {
    int64_t id = ids[i]; int32_t c = cnts[i]; int j = i - 1;
    while (j >= 0 && ids[j] > id) { ids[j + 1] = ids[j]; cnts[j + 1] = cnts[j]; j--; }
    ids[j + 1] = id; cnts[j + 1] = c;
  }
  return n;
}

static uint64_t splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

typedef struct { float p; int64_t id; } cand;
static int cand_cmp(const void* a, const void* b) {
  const cand* x = a; const cand* y = b;
  if (x->p > y->p) return -1;
  if (x->p < y->p) return 1;
  return (x->id > y->id) - (x->id < y->id);
}

int64_t orc_logits_to_token(const float* logits, int64_t V, const int64_t* ids, const int32_t* cnts, int n,
                            float rp, float fp, float pp, float temperature, int top_k, float top_p, float min_p,
                            uint64_t seed) {
  float* l = (float*)malloc(sizeof(float) * (size_t)V);
  memcpy(l, logits, sizeof(float) * (size_t)V);
  for (int i = 0; i < n; i++) {
    int64_t id = ids[i];
    if (id < 0 || id >= V) continue;
    float x = l[id];
    if (rp != 1.0f) x = (x > 0.0f) ? x / rp : x * rp; /* ASSUMPTION: llama.cpp sign rule */
    x -= fp * (float)cnts[i] + pp;                     /* generation.rs:40-47: freq/presence subtract */
    l[id] = x;
  }
  int64_t tok;
  if (temperature == 0.0f) { /* generation.rs:262-264 greedy == temperature 0 */
    tok = orc_argmax(l, V);
  } else {
    cand* c = (cand*)malloc(sizeof(cand) * (size_t)V);
    float m = -INFINITY;
    for (int64_t i = 0; i < V; i++) { l[i] /= temperature; if (l[i] > m) m = l[i]; }
    double sum = 0.0;
    for (int64_t i = 0; i < V; i++) { c[i].p = orc_expf(l[i] - m); c[i].id = i; sum += c[i].p; }
    for (int64_t i = 0; i < V; i++) c[i].p = (float)(c[i].p / sum);
    qsort(c, (size_t)V, sizeof(cand), cand_cmp);
    int64_t keep = V;
    if (top_k > 0 && top_k < keep) keep = top_k;
    if (top_p > 0.0f && top_p < 1.0f) {
      double cum = 0.0; int64_t i = 0;
      for (; i < keep; i++) { cum += c[i].p; if (cum >= top_p) { i++; break; } }
      if (i < keep) keep = i;
    }
    if (min_p > 0.0f) {
      float thr = c[0].p * min_p; int64_t i = 1;
      for (; i < keep; i++) if (c[i].p < thr) break;
      keep = i;
    }
    double tot = 0.0;
    for (int64_t i = 0; i < keep; i++) tot += c[i].p;
    uint64_t st = seed;
    double u = (double)(splitmix64(&st) >> 11) * (1.0 / 9007199254740992.0) * tot;
    double cum = 0.0; tok = c[keep - 1].id;
    for (int64_t i = 0; i < keep; i++) { cum += c[i].p; if (u < cum) { tok = c[i].id; break; } }
    free(c);
  }
  free(l);
  return tok;
}
