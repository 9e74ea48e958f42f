/*
 * oracle/orc_dsv2.c -- DeepSeek-V2 family decode step: MLA attention over a compressed-latent cache + MoE FFN
 * (TEST INFRASTRUCTURE; see orc.h).
 *
 * Reference anchors:
 *   MLA config fields kv_latent_dim / q_latent_dim / d_rope            /root/reference/src/loader/gguf.rs:188-196
 *   MLA components (kv_a_proj_with_mqa, kv_b_proj, q_a/q_b, o_proj), "stores compressed latents ... expands on the fly"
 *                                                                      /root/reference/docs/architecture.md:65-93
 *   MoE routing: scores = softmax(gate(x)); top-k; sum(score * expert(x)); + shared expert
 *                                                                      /root/reference/docs/architecture.md:108-119
 *   MoeConfig (expert_count, expert_used_count)                        /root/reference/src/loader/gguf.rs:271-283
 *   stacked expert weights [E, in, out], ExpertWeights{gate,up,down}   /root/reference/src/engine/executor_cache.rs:218-219,344-348
 * The arithmetic is in the absent boostr crate: this restates the public DeepSeek-V2 model (HF modeling_deepseek.py semantics,
 * q_lora_rank optional, decoupled RoPE on interleaved pairs, softmax scale 1/sqrt(nope+rope), router in f32, greedy top-k,
 * norm_topk_prob / routed_scaling_factor) in the weight-absorbed decode form, every tensor rounded to the activation dtype
 * at op boundaries (ASSUMPTION; parity unpinned):
 *   xn   = rmsnorm(h, attn_norm)
 *   q    = R(Wq xn)                                   [n_heads][nope + rope]      (q_lora: q = R(Wqb rmsnorm(R(Wqa xn))))
 *   kva  = R(Wkva xn) -> c = rmsnorm(kva[:rank], kv_norm) ; kpe = R(rope(kva[rank:]))  -> cache[pos] = (c, kpe)
 *   per head: qpe = R(rope(q_pe)); qabs = R(Wuk_h^T q_nope)      (Wuk_h = kv_b rows [h(nope+v), +nope))
 *             s_t = (qabs . c_t + qpe . kpe_t) * scale ; p = softmax(s) (f32)
 *             olat = R(sum_t p_t c_t) ; out_h = R(Wuv_h olat)    (Wuv_h = kv_b rows [h(nope+v)+nope, +v))
 *   h    = R(h + R(Wo out))
 *   xn2  = rmsnorm(h, ffn_norm)
 *   dense layer (< first_dense): h = R(h + R(Wd R(R(silu(R(Wg xn2))) * R(Wu xn2))))
 *   MoE layer: s = softmax_f32(Wr xn2) (f32, unrounded) ; top-k greedy (ties -> lowest index) ; w_k = s_k * routed_scale
 *              (/ sum of the selected s when norm_topk) ; routed = R(sum_k w_k * mlp_{e_k}(xn2)) (f32 sum in selection order) ;
 *              h = R(h + R(routed + mlp_shared(xn2)))
 * prefill = the same step token by token.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

orc_dsv2* orc_dsv2_new(const orc_dsv2_cfg* cfg) {
  orc_dsv2* m = (orc_dsv2*)calloc(1, sizeof(orc_dsv2));
  m->cfg = *cfg;
  m->layers = (orc_dsv2_layer*)calloc((size_t)cfg->n_layers, sizeof(orc_dsv2_layer));
  orc_rope_cfg rc = cfg->rope;
  rc.head_dim = cfg->rope_dim; rc.max_pos = cfg->max_seq_len;
  m->cos_t = (float*)malloc(sizeof(float) * (size_t)cfg->max_seq_len * (cfg->rope_dim / 2));
  m->sin_t = (float*)malloc(sizeof(float) * (size_t)cfg->max_seq_len * (cfg->rope_dim / 2));
  orc_rope_tables(&rc, m->cos_t, m->sin_t);
  return m;
}

/* call once after the layer structs are filled: dequantised copy of kv_b for the absorbed products */
void orc_dsv2_prepare(orc_dsv2* m) {
  for (int l = 0; l < m->cfg.n_layers; l++) {
    orc_dsv2_layer* L = &m->layers[l];
    free(L->kv_b_f32);
    L->kv_b_f32 = (float*)malloc(sizeof(float) * (size_t)L->kv_b.N * L->kv_b.K);
    orc_linear_dequant(&L->kv_b, L->kv_b_f32);
  }
}

void orc_dsv2_free(orc_dsv2* m) {
  if (!m) return;
  for (int l = 0; l < m->cfg.n_layers; l++) free(m->layers[l].kv_b_f32);
  free(m->layers); free(m->cos_t); free(m->sin_t); free(m);
}

orc_mla_cache* orc_mla_cache_new(const orc_dsv2_cfg* c, int capacity) {
  orc_mla_cache* k = (orc_mla_cache*)calloc(1, sizeof(orc_mla_cache));
  k->n_layers = c->n_layers; k->width = c->kv_lora_rank + c->rope_dim; k->capacity = capacity;
  k->lat = (float*)calloc((size_t)c->n_layers * capacity * k->width, sizeof(float));
  return k;
}
void orc_mla_cache_free(orc_mla_cache* k) { if (!k) return; free(k->lat); free(k); }

static void mlp(const orc_linear* g, const orc_linear* u, const orc_linear* d, const float* x, int act, float* tg, float* tu, float* out) {
  orc_linear_forward(g, x, 1, tg); orc_round_vec(tg, (size_t)g->N, act);
  orc_linear_forward(u, x, 1, tu); orc_round_vec(tu, (size_t)u->N, act);
  for (int i = 0; i < g->N; i++) tg[i] = orc_round(orc_round(orc_silu(tg[i]), act) * tu[i], act);
  orc_linear_forward(d, tg, 1, out); orc_round_vec(out, (size_t)d->N, act);
}

/* router: f32 softmax + greedy top-k.  sel/w in selection order (descending score, ties -> lowest index) */
void orc_moe_route(const float* logits, int E, int top_k, float routed_scale, int norm_topk, int* sel, float* w) {
  float m = -INFINITY, sum = 0.0f;
  float* s = (float*)malloc(sizeof(float) * (size_t)E);
  for (int e = 0; e < E; e++) if (logits[e] > m) m = logits[e];
  for (int e = 0; e < E; e++) { s[e] = orc_expf(logits[e] - m); sum += s[e]; }
  for (int e = 0; e < E; e++) s[e] = s[e] / sum;
  float tsum = 0.0f;
  for (int k = 0; k < top_k; k++) {
    int best = -1;
    for (int e = 0; e < E; e++) {
      int taken = 0;
      for (int j = 0; j < k; j++) if (sel[j] == e) taken = 1;
      if (!taken && (best < 0 || s[e] > s[best])) best = e;
    }
    sel[k] = best; w[k] = s[best]; tsum += s[best];
  }
  for (int k = 0; k < top_k; k++) w[k] = norm_topk ? w[k] / (tsum + 1e-20f) * routed_scale : w[k] * routed_scale;
  free(s);
}

int orc_dsv2_forward(const orc_dsv2* m, const int64_t* tokens, int S, orc_mla_cache* kc, int position, float* logits, int all_logits) {
  const orc_dsv2_cfg* c = &m->cfg;
  const int H = c->hidden, NH = c->n_heads, R = c->kv_lora_rank, DN = c->nope_dim, DR = c->rope_dim, DV = c->v_dim, act = c->act_dtype, V = c->vocab;
  const int QH = DN + DR, W = R + DR;
  if (position + S > kc->capacity || position + S > c->max_seq_len) return -1;
  const float msc = c->softmax_mscale > 0.f ? c->softmax_mscale : 1.0f;
  const float scale = 1.0f / sqrtf((float)QH) * msc * msc;
  const int imax = c->inter > c->n_shared * c->moe_inter ? c->inter : c->n_shared * c->moe_inter;
  float* h = (float*)malloc(sizeof(float) * H); float* xn = (float*)malloc(sizeof(float) * H); float* o = (float*)malloc(sizeof(float) * H);
  float* q = (float*)malloc(sizeof(float) * (size_t)NH * QH); float* kva = (float*)malloc(sizeof(float) * W);
  float* qa = (float*)malloc(sizeof(float) * (size_t)(c->q_lora_rank > 0 ? c->q_lora_rank : 1));
  float* att = (float*)malloc(sizeof(float) * (size_t)NH * DV);
  float* qabs = (float*)malloc(sizeof(float) * R); float* olat = (float*)malloc(sizeof(float) * R); double* qabsd = (double*)malloc(sizeof(double) * R);
  float* sc = (float*)malloc(sizeof(float) * (size_t)(position + S));
  float* tg = (float*)malloc(sizeof(float) * (size_t)(imax > c->moe_inter ? imax : c->moe_inter)); float* tu = (float*)malloc(sizeof(float) * (size_t)(imax > c->moe_inter ? imax : c->moe_inter));
  float* ye = (float*)malloc(sizeof(float) * H); float* routed = (float*)malloc(sizeof(float) * H);
  float* rl = (float*)malloc(sizeof(float) * (size_t)(c->n_experts > 0 ? c->n_experts : 1));
  int* sel = (int*)malloc(sizeof(int) * (size_t)(c->top_k > 0 ? c->top_k : 1)); float* sw = (float*)malloc(sizeof(float) * (size_t)(c->top_k > 0 ? c->top_k : 1));
  for (int s = 0; s < S; s++) {
    const int pos = position + s, len = pos + 1;
    const size_t row = (size_t)tokens[s] * H;
    for (int i = 0; i < H; i++) {
      float v = m->embed_dtype == ORC_F32 ? ((const float*)m->embed)[row + i]
              : (m->embed_dtype == ORC_F16 ? orc_f16_to_f32(((const uint16_t*)m->embed)[row + i]) : orc_bf16_to_f32(((const uint16_t*)m->embed)[row + i]));
      h[i] = orc_round(v, act);
    }
    const float* cr = m->cos_t + (size_t)pos * (DR / 2); const float* sr = m->sin_t + (size_t)pos * (DR / 2);
    for (int l = 0; l < c->n_layers; l++) {
      const orc_dsv2_layer* L = &m->layers[l];
      orc_rms_norm(h, L->attn_norm, H, c->rms_eps, act, xn);
      if (c->q_lora_rank > 0) {
        orc_linear_forward(&L->q_proj, xn, 1, qa); orc_round_vec(qa, (size_t)c->q_lora_rank, act);
        orc_rms_norm(qa, L->q_norm, c->q_lora_rank, c->rms_eps, act, qa);
        orc_linear_forward(&L->q_b, qa, 1, q);
      } else {
        orc_linear_forward(&L->q_proj, xn, 1, q);
      }
      orc_round_vec(q, (size_t)NH * QH, act);
      orc_linear_forward(&L->kv_a, xn, 1, kva); orc_round_vec(kva, (size_t)W, act);
      float* crow = kc->lat + ((size_t)l * kc->capacity + pos) * W;
      orc_rms_norm(kva, L->kv_norm, R, c->rms_eps, act, crow);
      orc_rope_apply(kva + R, DR, DR, cr, sr, 1); orc_round_vec(kva + R, (size_t)DR, act);
      memcpy(crow + R, kva + R, sizeof(float) * DR);
      const float* lat = kc->lat + (size_t)l * kc->capacity * W;
      for (int hd = 0; hd < NH; hd++) {
        float* qh = q + (size_t)hd * QH;
        orc_rope_apply(qh + DN, DR, DR, cr, sr, 1); orc_round_vec(qh + DN, (size_t)DR, act);
        const float* Wuk = L->kv_b_f32 + (size_t)hd * (DN + DV) * R;
        const float* Wuv = Wuk + (size_t)DN * R;
        /* every sum is the exactly rounded one (round 3, as in orc_ops.c / orc_quant.c: products of f32 values are exact in double, the sum is carried in double and
           rounded to f32 once), so that an implementation's result does not depend on its order of summation */
        for (int r = 0; r < R; r++) qabsd[r] = 0.0;
        for (int d = 0; d < DN; d++) { const double qd = (double)qh[d]; const float* wr = Wuk + (size_t)d * R; for (int r = 0; r < R; r++) qabsd[r] += qd * (double)wr[r]; }
        for (int r = 0; r < R; r++) qabs[r] = orc_round((float)qabsd[r], act);
        float mx = -INFINITY;
        for (int t = 0; t < len; t++) {
          const float* ct = lat + (size_t)t * W;
          double d0 = 0.0;
          for (int r = 0; r < R; r++) d0 += (double)qabs[r] * (double)ct[r];
          for (int j = 0; j < DR; j++) d0 += (double)qh[DN + j] * (double)ct[R + j];
          sc[t] = (float)d0 * scale;
          if (sc[t] > mx) mx = sc[t];
        }
        double sumd = 0.0;
        for (int t = 0; t < len; t++) { sc[t] = orc_expf(sc[t] - mx); sumd += (double)sc[t]; }
        const float inv = 1.0f / (float)sumd;
        for (int r = 0; r < R; r++) qabsd[r] = 0.0;
        for (int t = 0; t < len; t++) { const float* ct = lat + (size_t)t * W; const double p = (double)sc[t]; for (int r = 0; r < R; r++) qabsd[r] += p * (double)ct[r]; }
        for (int r = 0; r < R; r++) olat[r] = orc_round((float)qabsd[r] * inv, act);
        for (int d = 0; d < DV; d++) {
          const float* wr = Wuv + (size_t)d * R;
          double a = 0.0;
          for (int r = 0; r < R; r++) a += (double)wr[r] * (double)olat[r];
          att[(size_t)hd * DV + d] = orc_round((float)a, act);
        }
      }
      orc_linear_forward(&L->o, att, 1, o); orc_round_vec(o, (size_t)H, act);
      for (int i = 0; i < H; i++) h[i] = orc_round(h[i] + o[i], act);
      orc_rms_norm(h, L->ffn_norm, H, c->rms_eps, act, xn);
      if (!L->is_moe) {
        mlp(&L->gate, &L->up, &L->down, xn, act, tg, tu, o);
      } else {
        orc_linear_forward(&L->router, xn, 1, rl);
        orc_moe_route(rl, c->n_experts, c->top_k, c->routed_scale, c->norm_topk, sel, sw);
        for (int i = 0; i < H; i++) routed[i] = 0.0f;
        for (int k = 0; k < c->top_k; k++) {
          const int e = sel[k];
          mlp(&L->e_gate[e], &L->e_up[e], &L->e_down[e], xn, act, tg, tu, ye);
          for (int i = 0; i < H; i++) routed[i] += sw[k] * ye[i];
        }
        orc_round_vec(routed, (size_t)H, act);
        if (c->n_shared > 0) {
          mlp(&L->s_gate, &L->s_up, &L->s_down, xn, act, tg, tu, ye);
          for (int i = 0; i < H; i++) o[i] = orc_round(routed[i] + ye[i], act);
        } else {
          memcpy(o, routed, sizeof(float) * H);
        }
      }
      for (int i = 0; i < H; i++) h[i] = orc_round(h[i] + o[i], act);
    }
    if (all_logits || s == S - 1) {
      orc_rms_norm(h, m->final_norm, H, c->rms_eps, act, xn);
      float* lo = logits + (size_t)(all_logits ? s : 0) * V;
      orc_linear_forward(&m->lm_head, xn, 1, lo);
      orc_round_vec(lo, (size_t)V, act);
    }
  }
  if (kc->seq_len < position + S) kc->seq_len = position + S;
  free(h); free(xn); free(o); free(q); free(kva); free(qa); free(att); free(qabs); free(olat); free(qabsd); free(sc); free(tg); free(tu); free(ye); free(routed);
  free(rl); free(sel); free(sw);
  return 0;
}

/* executor_generate.rs:341-410 (contiguous branch), greedy */
int orc_dsv2_generate(const orc_dsv2* m, const int64_t* prompt, int n_prompt, int max_tokens, int64_t eos_id, int64_t* out_tokens, float* logits_trace) {
  const int V = m->cfg.vocab;
  if (max_tokens > m->cfg.max_seq_len - n_prompt) max_tokens = m->cfg.max_seq_len - n_prompt;
  if (max_tokens < 0) max_tokens = 0;
  orc_mla_cache* kc = orc_mla_cache_new(&m->cfg, n_prompt + max_tokens + 1);
  float* logits = (float*)malloc(sizeof(float) * V);
  int n_out = 0;
  orc_dsv2_forward(m, prompt, n_prompt, kc, 0, logits, 0);
  for (int i = 0; i < max_tokens; i++) {
    int64_t tok = orc_argmax(logits, V);
    if (logits_trace) memcpy(logits_trace + (size_t)i * V, logits, sizeof(float) * V);
    out_tokens[n_out++] = tok;
    if (tok == eos_id || i + 1 == max_tokens) break;
    orc_dsv2_forward(m, &tok, 1, kc, n_prompt + i, logits, 0);
  }
  free(logits); orc_mla_cache_free(kc);
  return n_out;
}
