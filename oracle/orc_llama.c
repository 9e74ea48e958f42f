/*
 * oracle/orc_llama.c -- Llama-family decode/prefill forward and the greedy loop (TEST INFRASTRUCTURE; see orc.h).
 *
 * Reference anchors:
 *   forward_with_kv_cache call sites      /root/reference/src/engine/executor_generate.rs:357,372
 *   forward_with_paged_kv_cache           /root/reference/src/engine/executor_generate.rs:259-262,289-292
 *   embed / layers_range / head pieces    /root/reference/src/cli/swarm_forward.rs:205,239-263
 *   (hidden, prev_mlp) deferred residual  /root/reference/src/cli/swarm_forward.rs:239-252
 *   KV cache layout [B, n_kv, cap, hd]    /root/reference/docs/architecture.md:137
 *   greedy loop                           /root/reference/src/engine/executor_generate.rs:341-410
 * The layer body itself is in the absent boostr crate; it restates HF LlamaDecoderLayer with every tensor
 * rounded to the activation dtype at op boundaries (ASSUMPTION; parity unpinned):
 *   h  = R(h + prev_mlp)                       (deferred residual from the previous layer)
 *   xn = rmsnorm(h, attn_norm)
 *   q,k,v = R(W xn + b) ; q,k = R(rope(q,k)) ; cache <- k,v
 *   o  = R(softmax(q k^T * hd^-0.5) v)         (f32 scores and probabilities)
 *   h  = R(h + R(Wo o))
 *   xn = rmsnorm(h, ffn_norm)
 *   prev_mlp = R(Wd R(R(silu(R(Wg xn))) * R(Wu xn)))
 * head: h = R(h + prev_mlp); logits = R(W_lm rmsnorm(h, final_norm))
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

orc_llama* orc_llama_new(const orc_llama_cfg* cfg) {
  orc_llama* m = (orc_llama*)calloc(1, sizeof(orc_llama));
  m->cfg = *cfg;
  m->layers = (orc_llama_layer*)calloc((size_t)cfg->n_layers, sizeof(orc_llama_layer));
  size_t tn = (size_t)cfg->max_seq_len * (size_t)(cfg->head_dim / 2);
  m->cos_t = (float*)malloc(sizeof(float) * tn);
  m->sin_t = (float*)malloc(sizeof(float) * tn);
  orc_rope_cfg rc = cfg->rope; rc.head_dim = cfg->head_dim; rc.max_pos = cfg->max_seq_len;
  orc_rope_tables(&rc, m->cos_t, m->sin_t);
  return m;
}

void orc_llama_free(orc_llama* m) { if (!m) return; free(m->layers); free(m->cos_t); free(m->sin_t); free(m); }

orc_kv* orc_kv_new(int L, int n_kv, int hd, int cap) {
  orc_kv* kv = (orc_kv*)calloc(1, sizeof(orc_kv));
  kv->n_layers = L; kv->n_kv_heads = n_kv; kv->head_dim = hd; kv->capacity = cap; kv->seq_len = 0;
  size_t n = (size_t)L * n_kv * cap * hd;
  kv->k = (float*)calloc(n, sizeof(float)); kv->v = (float*)calloc(n, sizeof(float));
  return kv;
}
void orc_kv_free(orc_kv* kv) { if (!kv) return; free(kv->k); free(kv->v); free(kv); }

orc_paged_kv* orc_paged_kv_new(int L, int num_blocks, int block_size, int n_kv, int hd) {
  orc_paged_kv* kv = (orc_paged_kv*)calloc(1, sizeof(orc_paged_kv));
  kv->n_layers = L; kv->n_kv_heads = n_kv; kv->head_dim = hd; kv->num_blocks = num_blocks; kv->block_size = block_size;
  size_t n = (size_t)L * num_blocks * n_kv * block_size * hd;
  kv->k = (float*)calloc(n, sizeof(float)); kv->v = (float*)calloc(n, sizeof(float));
  return kv;
}
void orc_paged_kv_free(orc_paged_kv* kv) { if (!kv) return; free(kv->k); free(kv->v); free(kv); }

void orc_llama_embed(const orc_llama* m, const int64_t* tokens, int S, float* hidden) {
  const int H = m->cfg.hidden;
  for (int s = 0; s < S; s++) {
    size_t row = (size_t)tokens[s] * H;
    for (int i = 0; i < H; i++) {
      float v;
      if (m->embed_dtype == ORC_F32) v = ((const float*)m->embed)[row + i];
      else if (m->embed_dtype == ORC_F16) v = orc_f16_to_f32(((const uint16_t*)m->embed)[row + i]);
      else v = orc_bf16_to_f32(((const uint16_t*)m->embed)[row + i]);
      hidden[(size_t)s * H + i] = orc_round(v, m->cfg.act_dtype); /* awq.rs:93-103 casts BF16 tensors to F16 */
    }
  }
}

/* kv access abstraction: contiguous or paged */
typedef struct {
  orc_kv* c; orc_paged_kv* p;
  const int32_t* slot_mapping; const int32_t* block_table; int n_table;
} kv_view;

static void kv_store(const kv_view* kvw, int layer, int kvh, int pos, int s_idx, const float* k, const float* v, int hd) {
  if (kvw->c) {
    orc_kv* c = kvw->c;
    size_t off = (((size_t)layer * c->n_kv_heads + kvh) * c->capacity + pos) * hd;
    memcpy(c->k + off, k, sizeof(float) * hd); memcpy(c->v + off, v, sizeof(float) * hd);
  } else {
    orc_paged_kv* p = kvw->p;
    int slot = kvw->slot_mapping[s_idx]; /* batch_decode.rs:81-88 : slot = block*block_size + offset */
    int blk = slot / p->block_size, o = slot % p->block_size;
    size_t off = ((((size_t)layer * p->num_blocks + blk) * p->n_kv_heads + kvh) * p->block_size + o) * hd;
    memcpy(p->k + off, k, sizeof(float) * hd); memcpy(p->v + off, v, sizeof(float) * hd);
  }
}

/* gather rows [0,len) of (layer,kvh) into contiguous tmp (paged) or return direct pointer (contiguous) */
static void kv_rows(const kv_view* kvw, int layer, int kvh, int len, int hd, float* tk, float* tv,
                    const float** ok, const float** ov) {
  if (kvw->c) {
    orc_kv* c = kvw->c;
    size_t off = (((size_t)layer * c->n_kv_heads + kvh) * c->capacity) * hd;
    *ok = c->k + off; *ov = c->v + off;
  } else {
    orc_paged_kv* p = kvw->p;
    for (int t = 0; t < len; t++) {
      int blk = kvw->block_table[t / p->block_size], o = t % p->block_size;
      size_t off = ((((size_t)layer * p->num_blocks + blk) * p->n_kv_heads + kvh) * p->block_size + o) * hd;
      memcpy(tk + (size_t)t * hd, p->k + off, sizeof(float) * hd);
      memcpy(tv + (size_t)t * hd, p->v + off, sizeof(float) * hd);
    }
    *ok = tk; *ov = tv;
  }
}

static int layers_range(const orc_llama* m, float* hidden, float* prev_mlp, int* has_prev, int S, const kv_view* kvw,
                        int start, int end, int position) {
  const orc_llama_cfg* c = &m->cfg;
  const int H = c->hidden, nq = c->n_heads, nkv = c->n_kv_heads, hd = c->head_dim, I = c->inter, act = c->act_dtype;
  const int half = hd / 2, rep = nq / nkv;
  const float scale = 1.0f / sqrtf((float)hd);
  if (position + S > c->max_seq_len) return -1;
  if (kvw->c && position + S > kvw->c->capacity) return -2;
  float* xn = (float*)malloc(sizeof(float) * (size_t)S * H);
  float* q = (float*)malloc(sizeof(float) * (size_t)S * nq * hd);
  float* k = (float*)malloc(sizeof(float) * (size_t)S * nkv * hd);
  float* v = (float*)malloc(sizeof(float) * (size_t)S * nkv * hd);
  float* ao = (float*)malloc(sizeof(float) * (size_t)S * nq * hd);
  float* t1 = (float*)malloc(sizeof(float) * (size_t)S * H);
  float* g = (float*)malloc(sizeof(float) * (size_t)S * I);
  float* u = (float*)malloc(sizeof(float) * (size_t)S * I);
  float* tk = NULL; float* tv = NULL;
  if (kvw->p) { tk = (float*)malloc(sizeof(float) * (size_t)(position + S) * hd); tv = (float*)malloc(sizeof(float) * (size_t)(position + S) * hd); }

  for (int l = start; l < end; l++) {
    const orc_llama_layer* L = &m->layers[l];
    if (*has_prev) for (size_t i = 0; i < (size_t)S * H; i++) hidden[i] = orc_round(hidden[i] + prev_mlp[i], act);
    for (int s = 0; s < S; s++) orc_rms_norm(hidden + (size_t)s * H, L->attn_norm, H, c->rms_eps, act, xn + (size_t)s * H);
    orc_linear_forward(&L->q, xn, S, q); orc_round_vec(q, (size_t)S * nq * hd, act);
    orc_linear_forward(&L->k, xn, S, k); orc_round_vec(k, (size_t)S * nkv * hd, act);
    orc_linear_forward(&L->v, xn, S, v); orc_round_vec(v, (size_t)S * nkv * hd, act);
    for (int s = 0; s < S; s++) {
      const int pos = position + s;
      const float* cr = m->cos_t + (size_t)pos * half; const float* sr = m->sin_t + (size_t)pos * half;
      for (int h = 0; h < nq; h++) { float* p = q + ((size_t)s * nq + h) * hd; orc_rope_apply(p, hd, hd, cr, sr, c->rope_interleaved); orc_round_vec(p, hd, act); }
      for (int h = 0; h < nkv; h++) { float* p = k + ((size_t)s * nkv + h) * hd; orc_rope_apply(p, hd, hd, cr, sr, c->rope_interleaved); orc_round_vec(p, hd, act); }
      for (int h = 0; h < nkv; h++) kv_store(kvw, l, h, pos, s, k + ((size_t)s * nkv + h) * hd, v + ((size_t)s * nkv + h) * hd, hd);
    }
    for (int s = 0; s < S; s++) {
      const int len = position + s + 1; /* causal */
      for (int h = 0; h < nkv; h++) {
        const float *kr, *vr;
        kv_rows(kvw, l, h, len, hd, tk, tv, &kr, &vr);
        orc_attn_decode(q + ((size_t)s * nq + (size_t)h * rep) * hd, rep, hd, kr, vr, (size_t)hd, len, scale,
                        ao + ((size_t)s * nq + (size_t)h * rep) * hd);
      }
    }
    orc_round_vec(ao, (size_t)S * nq * hd, act);
    orc_linear_forward(&L->o, ao, S, t1); orc_round_vec(t1, (size_t)S * H, act);
    for (size_t i = 0; i < (size_t)S * H; i++) hidden[i] = orc_round(hidden[i] + t1[i], act);
    for (int s = 0; s < S; s++) orc_rms_norm(hidden + (size_t)s * H, L->ffn_norm, H, c->rms_eps, act, xn + (size_t)s * H);
    orc_linear_forward(&L->gate, xn, S, g); orc_round_vec(g, (size_t)S * I, act);
    orc_linear_forward(&L->up, xn, S, u); orc_round_vec(u, (size_t)S * I, act);
    for (size_t i = 0; i < (size_t)S * I; i++) g[i] = orc_round(orc_round(orc_silu(g[i]), act) * u[i], act);
    orc_linear_forward(&L->down, g, S, prev_mlp); orc_round_vec(prev_mlp, (size_t)S * H, act);
    *has_prev = 1;
  }
  free(xn); free(q); free(k); free(v); free(ao); free(t1); free(g); free(u); free(tk); free(tv);
  return 0;
}

int orc_llama_layers_range(const orc_llama* m, float* hidden, float* prev_mlp, int* has_prev, int S, orc_kv* kv,
                           int start, int end, int position) {
  kv_view w = {kv, NULL, NULL, NULL, 0};
  int rc = layers_range(m, hidden, prev_mlp, has_prev, S, &w, start, end, position);
  if (rc == 0 && end == m->cfg.n_layers) kv->seq_len = position + S;
  return rc;
}

void orc_llama_head(const orc_llama* m, const float* hidden, const float* prev_mlp, int has_prev, int S, float* logits,
                    int all_logits) {
  const int H = m->cfg.hidden, V = m->cfg.vocab, act = m->cfg.act_dtype;
  const int s0 = all_logits ? 0 : S - 1;
  float* h = (float*)malloc(sizeof(float) * H); float* xn = (float*)malloc(sizeof(float) * H);
  for (int s = s0; s < S; s++) {
    for (int i = 0; i < H; i++) {
      float x = hidden[(size_t)s * H + i];
      h[i] = has_prev ? orc_round(x + prev_mlp[(size_t)s * H + i], act) : x;
    }
    orc_rms_norm(h, m->final_norm, H, m->cfg.rms_eps, act, xn);
    float* out = logits + (size_t)(all_logits ? s : 0) * V;
    orc_linear_forward(&m->lm_head, xn, 1, out);
    orc_round_vec(out, (size_t)V, act);
  }
  free(h); free(xn);
}

int orc_llama_forward_kv(const orc_llama* m, const int64_t* tokens, int S, orc_kv* kv, int position, float* logits,
                         int all_logits) {
  const int H = m->cfg.hidden;
  float* hidden = (float*)malloc(sizeof(float) * (size_t)S * H);
  float* prev = (float*)malloc(sizeof(float) * (size_t)S * H);
  int has_prev = 0;
  orc_llama_embed(m, tokens, S, hidden);
  int rc = orc_llama_layers_range(m, hidden, prev, &has_prev, S, kv, 0, m->cfg.n_layers, position);
  if (rc == 0) orc_llama_head(m, hidden, prev, has_prev, S, logits, all_logits);
  free(hidden); free(prev);
  return rc;
}

int orc_llama_forward_paged(const orc_llama* m, const int64_t* tokens, int S, orc_paged_kv* kv,
                            const int32_t* slot_mapping, const int32_t* block_table, int n_table, int seq_len_k,
                            int start_pos, float* logits, int all_logits) {
  const int H = m->cfg.hidden;
  if (start_pos + S != seq_len_k) return -3;
  float* hidden = (float*)malloc(sizeof(float) * (size_t)S * H);
  float* prev = (float*)malloc(sizeof(float) * (size_t)S * H);
  int has_prev = 0;
  orc_llama_embed(m, tokens, S, hidden);
  kv_view w = {NULL, kv, slot_mapping, block_table, n_table};
  int rc = layers_range(m, hidden, prev, &has_prev, S, &w, 0, m->cfg.n_layers, start_pos);
  if (rc == 0) { kv->seq_len = seq_len_k; orc_llama_head(m, hidden, prev, has_prev, S, logits, all_logits); }
  free(hidden); free(prev);
  return rc;
}

/* executor_generate.rs:341-410 (contiguous branch), sampling via logits_to_token (sampling.rs:445-460) */
int orc_llama_generate(const orc_llama* m, const int64_t* prompt, int n_prompt, int max_tokens, float repeat_penalty,
                       int repeat_last_n, int64_t eos_id, int64_t* out_tokens, float* logits_trace) {
  const int V = m->cfg.vocab;
  int cap = n_prompt + max_tokens; if (cap > m->cfg.max_seq_len) cap = m->cfg.max_seq_len;
  if (max_tokens > m->cfg.max_seq_len - n_prompt) max_tokens = m->cfg.max_seq_len - n_prompt; /* :80-82 */
  orc_kv* kv = orc_kv_new(m->cfg.n_layers, m->cfg.n_kv_heads, m->cfg.head_dim, cap);
  float* logits = (float*)malloc(sizeof(float) * V);
  uint32_t* hist = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_prompt + max_tokens + 1));
  int nh = 0;
  for (int i = 0; i < n_prompt; i++) hist[nh++] = (uint32_t)prompt[i];
  int64_t ids[4096]; int32_t cnts[4096];
  int n_out = 0;
  if (orc_llama_forward_kv(m, prompt, n_prompt, kv, 0, logits, 0) != 0) { n_out = -1; goto done; }
  for (int i = 0; i < max_tokens; i++) {
    int nw = 0;
    if (repeat_penalty != 1.0f) nw = orc_penalty_window(hist, nh, repeat_last_n, ids, cnts);
    int64_t tok = orc_logits_to_token(logits, V, ids, cnts, nw, repeat_penalty, 0.0f, 0.0f, 0.0f, 0, 1.0f, 0.0f, 0);
    if (logits_trace) memcpy(logits_trace + (size_t)i * V, logits, sizeof(float) * V);
    out_tokens[n_out++] = tok; hist[nh++] = (uint32_t)tok;
    if (tok == eos_id) break;
    if (i + 1 == max_tokens) break; /* reference runs one more forward whose logits are discarded (:372) */
    if (orc_llama_forward_kv(m, &tok, 1, kv, kv->seq_len, logits, 0) != 0) { n_out = -1; break; }
  }
done:
  free(logits); free(hist); orc_kv_free(kv);
  return n_out;
}
