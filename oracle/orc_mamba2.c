/*
 * oracle/orc_mamba2.c -- Mamba2 decode/prefill step (TEST INFRASTRUCTURE; see orc.h).
 *
 * Reference anchors:
 *   LayeredSsmState::new(layers, batch, mamba_config, dtype, device)   /root/reference/src/engine/executor_generate.rs:131-133
 *   forward_with_ssm_state(&input, &mut ssm)                           /root/reference/src/engine/executor_generate.rs:137,148
 *   state shapes ssm [B, n_heads, head_dim, d_state], conv [B, conv_dim, k-1]   /root/reference/docs/architecture.md:52-54
 *   SsmConfig fields (num_heads, head_dim, state_size, n_groups, conv_kernel, chunk 256)  /root/reference/src/loader/gguf.rs:219-262
 * The recurrence itself is in the absent boostr crate: this restates the public Mamba2 reference (HF Mamba2Mixer, single-token
 * path), every tensor rounded to the activation dtype at op boundaries (ASSUMPTION; parity unpinned):
 *   xn      = rmsnorm(h, norm)
 *   zxbcdt  = R(W_in xn)                       -> z [d_inner] | xBC [d_inner + 2 G d_state] | dt [n_heads]
 *   xBC     = R(silu(R(conv_state . w[:, :k-1] + xBC . w[:, k-1] + b)))     ; conv_state <- shift in raw xBC
 *   dt      = R(softplus(R(dt + dt_bias))) ; dA = exp(dt * -exp(A_log))     (f32)
 *   state   = R(state * dA + (dt * x) * B)     [head, p, n]   (stored in the activation dtype)
 *   y       = R(sum_n state * C + D * x)
 *   y       = R(y * R(silu(z))) ; y = R(w_norm * R(y * rsqrt(mean_group(y^2) + eps)))
 *   h       = R(h + R(W_out y))
 * prefill = the same step applied token by token (the reference's chunked scan is an optimisation of this recurrence).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* log1p through double (glibc here, ocml on the GPU: both faithful in double, one rounding to f32 -- the same float unless the double result sits within 1e-16 of a
   rounding boundary); the f32 log1pf of the two libraries differ in the last bit often enough to matter */
static float softplus_f(float x) { return x > 20.0f ? x : (float)log1p((double)orc_expf(x)); }

orc_mamba2* orc_mamba2_new(const orc_mamba2_cfg* cfg) {
  orc_mamba2* m = (orc_mamba2*)calloc(1, sizeof(orc_mamba2));
  m->cfg = *cfg;
  m->layers = (orc_mamba2_layer*)calloc((size_t)cfg->n_layers, sizeof(orc_mamba2_layer));
  return m;
}
void orc_mamba2_free(orc_mamba2* m) { if (!m) return; free(m->layers); free(m); }

orc_ssm_state* orc_ssm_state_new(const orc_mamba2_cfg* c) {
  orc_ssm_state* s = (orc_ssm_state*)calloc(1, sizeof(orc_ssm_state));
  const int conv_dim = c->d_inner + 2 * c->n_groups * c->d_state;
  s->ssm = (float*)calloc((size_t)c->n_layers * c->n_heads * c->head_dim * c->d_state, sizeof(float));
  s->conv = (float*)calloc((size_t)c->n_layers * conv_dim * (c->conv_kernel - 1), sizeof(float));
  return s;
}
void orc_ssm_state_free(orc_ssm_state* s) { if (!s) return; free(s->ssm); free(s->conv); free(s); }

/* The two halves of the mixer between in_proj and the gated norm, as separate entry points (op-level parity tests call them; the forward below is built
   from them).  zx = the rounded in_proj row [z (d_inner) | x B C (conv_dim) | dt (n_heads)].
   conv1d step: depthwise causal convolution over the window [state (k-1 values) | this token's value], + bias, R, SiLU, R; the window moves on. */
void orc_mamba2_conv1d_step(const orc_mamba2_cfg* c, const orc_mamba2_layer* L, const float* zx, float* conv_state, float* xbc) {
  const int DI = c->d_inner, NS = c->d_state, G = c->n_groups, KC = c->conv_kernel, act = c->act_dtype;
  const int conv_dim = DI + 2 * G * NS;
  const float* xraw = zx + DI;
  float* cs = conv_state;
  for (int ch = 0; ch < conv_dim; ch++) {
    /* the window's dot product is the exactly rounded one (products exact in double, one rounding to f32), like every other sum of the oracle: an f32 chain would make
       the result depend on whether an implementation fuses its multiply-adds */
    double ad = 0.0;
    for (int j = 0; j < KC - 1; j++) ad += (double)cs[(size_t)ch * (KC - 1) + j] * (double)L->conv_w[(size_t)ch * KC + j];
    ad += (double)xraw[ch] * (double)L->conv_w[(size_t)ch * KC + KC - 1];
    float a = (float)ad;
    a = orc_round(a + L->conv_b[ch], act);
    xbc[ch] = orc_round(orc_silu(a), act);
    for (int j = 0; j + 1 < KC - 1; j++) cs[(size_t)ch * (KC - 1) + j] = cs[(size_t)ch * (KC - 1) + j + 1];
    cs[(size_t)ch * (KC - 1) + KC - 2] = xraw[ch];
  }
}
/* SSM step (HF Mamba2 single-token recurrence): h = R(h exp(dt A) + (dt x) B), y = R(C.h + D x), then the gate y = R(y R(silu z)); ssm = this layer's
   [n_heads][head_dim][d_state] */
void orc_mamba2_ssm_step(const orc_mamba2_cfg* c, const orc_mamba2_layer* L, const float* zx, const float* xbc, float* ssm, float* y) {
  const int DI = c->d_inner, NH = c->n_heads, HD = c->head_dim, NS = c->d_state, G = c->n_groups, act = c->act_dtype;
  const int conv_dim = DI + 2 * G * NS;
  const float* z = zx; const float* dtr = zx + DI + conv_dim;
  const float* x = xbc; const float* Bm = xbc + DI; const float* Cm = xbc + DI + G * NS;
  for (int hd = 0; hd < NH; hd++) {
    const int g = hd / (NH / G);
    const float dt = orc_round(softplus_f(orc_round(dtr[hd] + L->dt_bias[hd], act)), act);
    const float dA = orc_expf(dt * -orc_expf(L->A_log[hd]));
    for (int p = 0; p < HD; p++) {
      const float xv = x[hd * HD + p];
      float* hs = ssm + ((size_t)hd * HD + p) * NS;
      /* state update: both products are exact in double, their sum is rounded once to double, once to f32, once to the state dtype; readout: the exactly rounded
         dot product C . h (double), then + D x in f32 */
      const float dtx = dt * xv;
      double acc = 0.0;
      for (int n = 0; n < NS; n++) {
        hs[n] = orc_round((float)((double)hs[n] * (double)dA + (double)dtx * (double)Bm[g * NS + n]), act);
        acc += (double)hs[n] * (double)Cm[g * NS + n];
      }
      y[hd * HD + p] = orc_round((float)acc + L->D[hd] * xv, act);
    }
  }
  for (int i = 0; i < DI; i++) y[i] = orc_round(y[i] * orc_round(orc_silu(z[i]), act), act);
}

int orc_mamba2_forward(const orc_mamba2* m, const int64_t* tokens, int S, orc_ssm_state* st, float* logits, int all_logits) {
  const orc_mamba2_cfg* c = &m->cfg;
  const int D = c->hidden, DI = c->d_inner, NH = c->n_heads, HD = c->head_dim, NS = c->d_state, G = c->n_groups, KC = c->conv_kernel;
  const int conv_dim = DI + 2 * G * NS, d_in = 2 * DI + 2 * G * NS + NH, act = c->act_dtype, V = c->vocab;
  float* h = (float*)malloc(sizeof(float) * D); float* xn = (float*)malloc(sizeof(float) * D);
  float* zx = (float*)malloc(sizeof(float) * d_in); float* xbc = (float*)malloc(sizeof(float) * conv_dim);
  float* y = (float*)malloc(sizeof(float) * DI); float* out = (float*)malloc(sizeof(float) * D);
  for (int s = 0; s < S; s++) {
    const size_t row = (size_t)tokens[s] * D;
    for (int i = 0; i < D; i++) {
      float v = m->embed_dtype == ORC_F32 ? ((const float*)m->embed)[row + i]
              : (m->embed_dtype == ORC_F16 ? orc_f16_to_f32(((const uint16_t*)m->embed)[row + i]) : orc_bf16_to_f32(((const uint16_t*)m->embed)[row + i]));
      h[i] = orc_round(v, act);
    }
    for (int l = 0; l < c->n_layers; l++) {
      const orc_mamba2_layer* L = &m->layers[l];
      orc_rms_norm(h, L->norm, D, c->rms_eps, act, xn);
      orc_linear_forward(&L->in_proj, xn, 1, zx); orc_round_vec(zx, (size_t)d_in, act);
      float* cs = st->conv + (size_t)l * conv_dim * (KC - 1);
      float* ss = st->ssm + (size_t)l * NH * HD * NS;
      orc_mamba2_conv1d_step(c, L, zx, cs, xbc);
      orc_mamba2_ssm_step(c, L, zx, xbc, ss, y);
      const int gsz = DI / G;
      for (int g = 0; g < G; g++) {
        double ssd = 0.0;
        for (int i = 0; i < gsz; i++) ssd += (double)(y[g * gsz + i] * y[g * gsz + i]);
        const float rs = 1.0f / sqrtf((float)ssd / (float)gsz + c->rms_eps);
        for (int i = 0; i < gsz; i++) y[g * gsz + i] = orc_round(L->gnorm[g * gsz + i] * orc_round(y[g * gsz + i] * rs, act), act);
      }
      orc_linear_forward(&L->out_proj, y, 1, out); orc_round_vec(out, (size_t)D, act);
      for (int i = 0; i < D; i++) h[i] = orc_round(h[i] + out[i], act);
    }
    if (all_logits || s == S - 1) {
      orc_rms_norm(h, m->final_norm, D, c->rms_eps, act, xn);
      float* lo = logits + (size_t)(all_logits ? s : 0) * V;
      orc_linear_forward(&m->lm_head, xn, 1, lo);
      orc_round_vec(lo, (size_t)V, act);
    }
  }
  free(h); free(xn); free(zx); free(xbc); free(y); free(out);
  return 0;
}

/* executor_generate.rs:123-181 (Mamba2 branch), greedy */
int orc_mamba2_generate(const orc_mamba2* m, const int64_t* prompt, int n_prompt, int max_tokens, int64_t eos_id, int64_t* out_tokens,
                        float* logits_trace) {
  const int V = m->cfg.vocab;
  orc_ssm_state* st = orc_ssm_state_new(&m->cfg);
  float* logits = (float*)malloc(sizeof(float) * V);
  int n_out = 0;
  orc_mamba2_forward(m, prompt, n_prompt, st, logits, 0);
  for (int i = 0; i < max_tokens; i++) {
    int64_t tok = orc_argmax(logits, V);
    if (logits_trace) memcpy(logits_trace + (size_t)i * V, logits, sizeof(float) * V);
    out_tokens[n_out++] = tok;
    if (tok == eos_id || i + 1 == max_tokens) break;
    orc_mamba2_forward(m, &tok, 1, st, logits, 0);
  }
  free(logits); orc_ssm_state_free(st);
  return n_out;
}
