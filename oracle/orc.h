/*
 * oracle/orc.h -- CPU restatement of blazr's quantised forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under blazr_amd/ (the product) may
 * include, link or dlopen anything in oracle/.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the crates
 * boostr 0.1.0 / numr 0.5.0 (reference Cargo.lock:352-353, 1940-1941), which
 * are absent from /root/reference and cannot be fetched; the reference holds
 * no golden vector, fixture or known-answer test for the path.
 * What IS pinned by reference source is structural and is followed to the
 * letter, each function citing the file:line it follows:
 *   - AWQ nibble order / tensor shapes  (src/loader/safetensors/awq.rs:3-6,29-32,239-263)
 *   - GPTQ tensor shapes / packed zeros (src/loader/safetensors/gptq.rs:3-8,198-247)
 *   - decode loop call order            (src/engine/executor_generate.rs:341-410)
 *   - penalty window                    (src/engine/sampling.rs:169-191)
 *   - paged slot = block*block_size+off (src/engine/batch_decode.rs:81-88)
 *   - RoPE llama3 scaling fields        (src/loader/safetensors/config.rs:83-95)
 * Everything else (HF Llama semantics, AutoAWQ/AutoGPTQ dequant formulas, GGML
 * block formats, Mamba2 recurrence, DeepSeek-V2 MLA/MoE) restates the public
 * format definitions the reference claims compatibility with and is flagged
 * "ASSUMPTION" where boostr could have chosen otherwise.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation / storage dtypes (values mirror include/blazr_hip.h BZ_DTYPE_*) */
enum { ORC_F32 = 0, ORC_F16 = 1, ORC_BF16 = 2 };

/* linear-layer storage kinds */
enum {
  ORC_LIN_DENSE = 0, /* w: [N][K] row-major, dtype in .w_dtype                      */
  ORC_LIN_AWQ   = 1, /* qweight u32 [K][N/8], scales f32 [G][N], zeros f32 [G][N]  */
  ORC_LIN_GPTQ  = 2, /* qweight u32 [K/8][N], scales f32 [G][N], qzeros u32 [G][N/8], g_idx i32[K]? */
  ORC_LIN_GGUF  = 3  /* raw ggml blocks, rows of K weights, N rows                   */
};

/* ggml type ids (public GGML enum values) */
enum { ORC_GGML_F32 = 0, ORC_GGML_F16 = 1, ORC_GGML_Q8_0 = 8, ORC_GGML_Q4_K = 12, ORC_GGML_Q6_K = 14, ORC_GGML_BF16 = 30 };

typedef struct {
  int kind;
  int N, K;             /* logical shape [N, K] = out x in (awq.rs:215-216, gptq.rs:249-250) */
  int group_size;       /* AWQ/GPTQ */
  int w_dtype;          /* DENSE: ORC_F32/F16/BF16 */
  int ggml_type;        /* GGUF */
  const void*     w;        /* DENSE weights / AWQ,GPTQ qweight / GGUF blocks */
  const float*    scales;   /* AWQ/GPTQ */
  const float*    zeros_f;  /* AWQ (already unpacked to f32, awq.rs:208-213) */
  const uint32_t* qzeros;   /* GPTQ packed (gptq.rs:209-219) */
  const int32_t*  g_idx;    /* GPTQ optional (gptq.rs:221-228) */
  const float*    bias;     /* optional f32 [N] (gptq.rs:230-242) */
} orc_linear;

/* ---- scalar helpers ---------------------------------------------------- */
float    orc_f16_to_f32(uint16_t h);
uint16_t orc_f32_to_f16(float f);      /* round-to-nearest-even */
float    orc_bf16_to_f32(uint16_t h);
uint16_t orc_f32_to_bf16(float f);     /* round-to-nearest-even, NaN preserved */
float    orc_round(float x, int dtype);/* round x to the nearest value representable in dtype */
void     orc_round_vec(float* x, size_t n, int dtype);

/* ---- quant formats ----------------------------------------------------- */
/* awq.rs:239-263 : packed qzeros [G][N/8] -> f32 [G][N] */
void orc_awq_unpack_zeros(const uint32_t* packed, int G, int N, float* out);
/* dequantise a whole linear to f32 [N][K] (row = output feature) */
void orc_linear_dequant(const orc_linear* L, float* out);
/* GGML block dequant: nblocks blocks of `type` -> f32 */
void   orc_ggml_dequant(int type, const void* blocks, size_t nweights, float* out);
size_t orc_ggml_row_bytes(int type, size_t K);
/* y[S][N] = x[S][K] . W^T (+bias), f32 accumulate, sequential over k */
void orc_linear_forward(const orc_linear* L, const float* x, int S, float* y);

/* ---- ops --------------------------------------------------------------- */
void orc_rms_norm(const float* x, const float* w, int n, float eps, int act, float* out);
typedef struct {
  int   head_dim, max_pos;
  float theta;
  int   scaling_type;        /* 0 none, 1 linear, 2 llama3, 3 yarn */
  float factor, low_freq_factor, high_freq_factor;
  int   original_max_pos;
  float beta_fast, beta_slow, attn_factor;   /* yarn: 0 = defaults (32, 1, 0.1 ln(factor) + 1); cos / sin are multiplied by attn_factor */
} orc_rope_cfg;
/* cos/sin tables [max_pos][head_dim/2] f32 */
void orc_rope_tables(const orc_rope_cfg* c, float* cos_t, float* sin_t);
/* in-place rotate one head vector; interleaved=0: HF half-split pairs (i, i+hd/2); 1: pairs (2i,2i+1) */
void orc_rope_apply(float* v, int head_dim, int rot_dim, const float* cos_row, const float* sin_row, int interleaved);
/* single-query attention over a cache of `len` positions.
   k/v rows are fetched through the callback-free strided view: kv[pos*stride + d] */
void orc_attn_decode(const float* q, int n_q_per_kv, int head_dim, const float* kcache, const float* vcache,
                     size_t pos_stride, int len, float scale, float* out /*[n_q_per_kv][head_dim]*/);
float orc_silu(float x);
float orc_expf(float x);   /* the specified exp (orc_ops.c) */
int64_t orc_argmax(const float* v, int64_t n); /* lowest index wins ties */

/* sampling.rs:445-460 logits_to_token. ASSUMPTION: llama.cpp-style repeat penalty
   (logit>0 ? logit/p : logit*p), then -= freq*count + presence. temperature==0 -> argmax. */
int64_t orc_logits_to_token(const float* logits, int64_t vocab, const int64_t* ids, const int32_t* cnts, int n,
                            float repeat_penalty, float freq_penalty, float presence_penalty,
                            float temperature, int top_k, float top_p, float min_p, uint64_t seed);
/* sampling.rs:169-191 penalty_window; ids returned in ascending id order (reference order is HashMap order) */
int orc_penalty_window(const uint32_t* recent, int n_recent, int repeat_last_n, int64_t* ids, int32_t* cnts);

/* ---- Llama-family model ------------------------------------------------ */
typedef struct {
  int hidden, n_layers, n_heads, n_kv_heads, head_dim, inter, vocab;
  float rms_eps;
  int   act_dtype;          /* ORC_F16 for AWQ/GPTQ (awq.rs:69-71), ORC_F32 for GGUF (gguf.rs:305) */
  int   rope_interleaved;
  int   max_seq_len;
  orc_rope_cfg rope;
} orc_llama_cfg;

typedef struct {
  const float* attn_norm; const float* ffn_norm;   /* f32 [hidden] (values representable in act dtype) */
  orc_linear q, k, v, o, gate, up, down;
} orc_llama_layer;

typedef struct {
  orc_llama_cfg cfg;
  const void*  embed; int embed_dtype;   /* [vocab][hidden] */
  const float* final_norm;
  orc_linear   lm_head;                  /* may alias embed (tied) */
  orc_llama_layer* layers;
  float *cos_t, *sin_t;                  /* owned */
} orc_llama;

typedef struct {
  int n_layers, n_kv_heads, head_dim, capacity, seq_len;
  float* k; float* v;                    /* [layer][kv_head][capacity][head_dim] (values rounded to act dtype) */
} orc_kv;

typedef struct {               /* inference.rs:189-191 block_size default 16 */
  int n_layers, n_kv_heads, head_dim, num_blocks, block_size, seq_len;
  float* k; float* v;          /* [layer][block][kv_head][block_size][head_dim] */
} orc_paged_kv;

orc_llama* orc_llama_new(const orc_llama_cfg* cfg);
void       orc_llama_free(orc_llama* m);
orc_kv*    orc_kv_new(int n_layers, int n_kv_heads, int head_dim, int capacity);
void       orc_kv_free(orc_kv* kv);
orc_paged_kv* orc_paged_kv_new(int n_layers, int num_blocks, int block_size, int n_kv_heads, int head_dim);
void       orc_paged_kv_free(orc_paged_kv* kv);

/* executor_generate.rs:357,372  LoadedModel::forward_with_kv_cache(input, kv, position)
   tokens i64[S]; logits out f32 [S][vocab] if all_logits else last position only [vocab] */
int orc_llama_forward_kv(const orc_llama* m, const int64_t* tokens, int S, orc_kv* kv, int position,
                         float* logits, int all_logits);
/* executor_generate.rs:259-262,289-292 forward_with_paged_kv_cache */
int orc_llama_forward_paged(const orc_llama* m, const int64_t* tokens, int S, orc_paged_kv* kv,
                            const int32_t* slot_mapping, const int32_t* block_table, int n_table,
                            int seq_len_k, int start_pos, float* logits, int all_logits);
/* swarm_forward.rs:205,239-263 three-piece path; hidden/prev_mlp are [S][hidden]; has_prev in/out flag */
void orc_llama_embed(const orc_llama* m, const int64_t* tokens, int S, float* hidden);
int  orc_llama_layers_range(const orc_llama* m, float* hidden, float* prev_mlp, int* has_prev, int S,
                            orc_kv* kv, int start, int end, int position);
void orc_llama_head(const orc_llama* m, const float* hidden, const float* prev_mlp, int has_prev, int S,
                    float* logits, int all_logits);

/* executor_generate.rs:341-410 contiguous-cache greedy loop (temperature 0, no penalties unless given).
   Returns number of generated tokens written to out_tokens; last_logits optional [vocab] of final sampled step */
int orc_llama_generate(const orc_llama* m, const int64_t* prompt, int n_prompt, int max_tokens,
                       float repeat_penalty, int repeat_last_n, int64_t eos_id,
                       int64_t* out_tokens, float* logits_trace /* optional [max_tokens][vocab] */);

/* ---- Mamba2 ------------------------------------------------------------- */
typedef struct {
  int hidden, n_layers, vocab;
  int d_inner, n_heads, head_dim, d_state, n_groups, conv_kernel;   /* SsmConfig (gguf.rs:219-262) */
  float rms_eps;
  int act_dtype;
} orc_mamba2_cfg;
typedef struct {
  const float* norm;            /* [hidden] */
  orc_linear in_proj;           /* [2 d_inner + 2 G d_state + n_heads, hidden] */
  const float* conv_w;          /* [conv_dim][k] */
  const float* conv_b;          /* [conv_dim] */
  const float* dt_bias; const float* A_log; const float* D;   /* [n_heads] */
  const float* gnorm;           /* [d_inner] gated RMSNorm weight */
  orc_linear out_proj;          /* [hidden, d_inner] */
} orc_mamba2_layer;
typedef struct {
  orc_mamba2_cfg cfg;
  const void* embed; int embed_dtype;
  const float* final_norm;
  orc_linear lm_head;
  orc_mamba2_layer* layers;
} orc_mamba2;
typedef struct { float* ssm; float* conv; } orc_ssm_state;   /* [L][n_heads][head_dim][d_state], [L][conv_dim][k-1] */
orc_mamba2* orc_mamba2_new(const orc_mamba2_cfg* cfg);
void orc_mamba2_free(orc_mamba2* m);
orc_ssm_state* orc_ssm_state_new(const orc_mamba2_cfg* cfg);
void orc_ssm_state_free(orc_ssm_state* s);
/* executor_generate.rs:137,148 forward_with_ssm_state */
int orc_mamba2_forward(const orc_mamba2* m, const int64_t* tokens, int S, orc_ssm_state* st, float* logits, int all_logits);
/* the mixer's two state-carrying ops on their own (zx = rounded in_proj row [z | x B C | dt]); conv_state / ssm = ONE layer's state */
void orc_mamba2_conv1d_step(const orc_mamba2_cfg* c, const orc_mamba2_layer* L, const float* zx, float* conv_state, float* xbc);
void orc_mamba2_ssm_step(const orc_mamba2_cfg* c, const orc_mamba2_layer* L, const float* zx, const float* xbc, float* ssm, float* y);
int orc_mamba2_generate(const orc_mamba2* m, const int64_t* prompt, int n_prompt, int max_tokens, int64_t eos_id, int64_t* out_tokens,
                        float* logits_trace);

/* ---- DeepSeek-V2 family (MLA + MoE), see orc_dsv2.c -------------------- */
typedef struct {
  int hidden, n_layers, n_heads, vocab, max_seq_len;
  int kv_lora_rank, q_lora_rank, nope_dim, rope_dim, v_dim;   /* AttentionConfig kv_latent_dim / q_latent_dim / d_rope (gguf.rs:188-196) */
  int inter;                                                  /* dense MLP of layers < first_dense */
  int n_experts, top_k, n_shared, moe_inter, first_dense;     /* MoeConfig (gguf.rs:271-283) */
  float routed_scale; int norm_topk;
  float rms_eps; int act_dtype;
  orc_rope_cfg rope;                                          /* head_dim / max_pos filled by orc_dsv2_new */
  float softmax_mscale;                                       /* YaRN (HF DeepseekV2Attention: softmax_scale * mscale^2, mscale from mscale_all_dim); 0 = 1 */
} orc_dsv2_cfg;
typedef struct {
  const float* attn_norm; const float* ffn_norm; const float* kv_norm; const float* q_norm;
  orc_linear q_proj;   /* [n_heads (nope+rope), hidden]; with q_lora_rank > 0 this is q_a [q_lora, hidden] and q_b expands */
  orc_linear q_b, kv_a, kv_b, o;
  int is_moe;
  orc_linear gate, up, down;                    /* dense layers */
  orc_linear router;                            /* [E, hidden] */
  orc_linear *e_gate, *e_up, *e_down;           /* [E] each */
  orc_linear s_gate, s_up, s_down;              /* shared experts as one MLP of n_shared * moe_inter */
  float* kv_b_f32;                              /* owned (orc_dsv2_prepare) */
} orc_dsv2_layer;
typedef struct {
  orc_dsv2_cfg cfg;
  const void* embed; int embed_dtype;
  const float* final_norm;
  orc_linear lm_head;
  orc_dsv2_layer* layers;
  float *cos_t, *sin_t;
} orc_dsv2;
typedef struct { int n_layers, width, capacity, seq_len; float* lat; } orc_mla_cache;   /* [L][capacity][rank + rope] */
orc_dsv2* orc_dsv2_new(const orc_dsv2_cfg* cfg);
void orc_dsv2_prepare(orc_dsv2* m);
void orc_dsv2_free(orc_dsv2* m);
orc_mla_cache* orc_mla_cache_new(const orc_dsv2_cfg* cfg, int capacity);
void orc_mla_cache_free(orc_mla_cache* k);
void orc_moe_route(const float* logits, int E, int top_k, float routed_scale, int norm_topk, int* sel, float* w);
int orc_dsv2_forward(const orc_dsv2* m, const int64_t* tokens, int S, orc_mla_cache* kc, int position, float* logits, int all_logits);
int orc_dsv2_generate(const orc_dsv2* m, const int64_t* prompt, int n_prompt, int max_tokens, int64_t eos_id, int64_t* out_tokens, float* logits_trace);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
