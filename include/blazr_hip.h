/*
 * blazr_hip.h -- C ABI of libblazr_hip.so: the MI355X (gfx950) forward path that sits where blazr's
 * `boostr` dependency sits today.
 *
 * blazr is generic over `R: boostr::Runtime` (/root/reference/src/engine/executor.rs:67-80).  A Rust
 * `HipRuntime` whose client methods forward to the functions below is the drop-in; the extern "C" block a
 * maintainer would add is shown in INTEGRATION.md.  Every entry point cites the reference interface it
 * replaces (file:line under /root/reference).
 *
 * Conventions
 *   - every function returns BZ_OK (0) or a negative BZ_E* code; bz_last_error() gives the message of the
 *     calling thread's last failure (boostr returns Result<_, E: Display>, stringified by blazr with
 *     anyhow!("...: {}", e), e.g. executor_generate.rs:138).  No C++ exception crosses the boundary.
 *   - handles are opaque; all device work is enqueued on the device handle's HIP stream; a cache / state
 *     object must not be used from two host threads at once (blazr owns one per request,
 *     executor_generate.rs:131,208,350).
 *   - plain pointers and sizes only.  "host" pointers are ordinary host memory; device memory is only
 *     reachable through bz_tensor handles.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with BZ_E_NODEVICE.
 */
#ifndef BLAZR_HIP_H
#define BLAZR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZ_ABI_VERSION 4

enum {
  BZ_OK = 0,
  BZ_E_INVALID = -1,    /* bad argument / shape / state */
  BZ_E_NODEVICE = -2,   /* no usable HIP device */
  BZ_E_HIP = -3,        /* HIP runtime error (message has hipGetErrorString) */
  BZ_E_UNSUPPORTED = -4,/* valid request this build does not implement */
  BZ_E_NOTFOUND = -5,   /* tensor name not registered */
  BZ_E_OOM = -6
};

/* boostr::DType::{F32,F16,BF16,I64,I32,U32} (executor_cache.rs:178, awq.rs:194, gptq.rs:222) + U8 for raw blocks */
enum { BZ_F32 = 0, BZ_F16 = 1, BZ_BF16 = 2, BZ_I64 = 3, BZ_I32 = 4, BZ_U32 = 5, BZ_U8 = 6 };

/* ggml type ids accepted by bz_model_add_gguf (format::Gguf tensor_info().ggml_type, loader/gguf.rs:33) */
enum { BZ_GGML_F32 = 0, BZ_GGML_F16 = 1, BZ_GGML_Q8_0 = 8, BZ_GGML_Q4_K = 12, BZ_GGML_Q6_K = 14, BZ_GGML_BF16 = 30 };

enum { BZ_ARCH_LLAMA = 0, BZ_ARCH_MAMBA2 = 1, BZ_ARCH_DEEPSEEK2 = 2 };
enum { BZ_ROPE_NONE = 0, BZ_ROPE_LINEAR = 1, BZ_ROPE_LLAMA3 = 2, BZ_ROPE_YARN = 3 };

typedef struct bz_device bz_device;
typedef struct bz_tensor bz_tensor;
typedef struct bz_model bz_model;
typedef struct bz_kv bz_kv;
typedef struct bz_paged_kv bz_paged_kv;
typedef struct bz_decode_graph bz_decode_graph;
typedef struct bz_ssm_state bz_ssm_state;

/* POD mirror of the fields of boostr::model::UniversalConfig that the forward path reads
 * (loader/safetensors/config.rs:31-95, loader/gguf.rs:101-306, config/blazr.rs:35-52). */
typedef struct {
  int32_t abi_version;     /* BZ_ABI_VERSION */
  int32_t arch;            /* BZ_ARCH_* */
  int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, inter, vocab;
  int32_t max_seq_len;
  float   rms_eps;         /* default 1e-5, gguf.rs:157-160 */
  int32_t act_dtype;       /* inference dtype: BZ_F16 for AWQ/GPTQ (awq.rs:69-71), BZ_F32 for GGUF (gguf.rs:305) */
  int32_t tie_embeddings;  /* lm_head aliases model.embed_tokens.weight */
  float   rope_theta;
  int32_t rope_interleaved;/* 0: HF half-split pairs, 1: GGML NORM pairs (2i,2i+1) */
  int32_t rope_scaling;    /* BZ_ROPE_* ; fields below per loader/safetensors/config.rs:83-95 */
  float   rope_factor, rope_low_freq_factor, rope_high_freq_factor;
  int32_t rope_original_max_pos;
  /* boostr::model::SsmConfig (loader/gguf.rs:219-262) -- BZ_ARCH_MAMBA2 only */
  int32_t ssm_d_inner, ssm_n_heads, ssm_head_dim, ssm_d_state, ssm_n_groups, ssm_conv_kernel;
  /* BZ_ARCH_DEEPSEEK2 only: AttentionConfig kv_latent_dim / q_latent_dim / d_rope (loader/gguf.rs:188-196) plus the HF head split,
   * MoeConfig expert_count / expert_used_count / shared experts (loader/gguf.rs:271-283).  For this arch `inter` is the dense MLP width of
   * layers < moe_first_dense, `n_kv_heads` / `head_dim` are ignored (the latent cache is created with n_kv = 1, head_dim = rank + rope). */
  int32_t mla_kv_lora_rank, mla_q_lora_rank, mla_nope_dim, mla_rope_dim, mla_v_dim;
  int32_t moe_n_experts, moe_top_k, moe_n_shared, moe_inter, moe_first_dense, moe_norm_topk;
  float   moe_routed_scale;
  /* BZ_ROPE_YARN (RopeScalingConfig beta_fast / beta_slow / attention_factor, loader/safetensors/config.rs:83-95 -- the reference maps them to None = the
   * defaults): 0 means default (beta_fast 32, beta_slow 1, attention_factor 0.1 ln(factor) + 1).  cos / sin are multiplied by the attention factor.
   * mla_softmax_mscale (DeepSeek-V2 YaRN, mscale_all_dim): the MLA softmax scale is multiplied by its square; 0 = 1. */
  float   rope_beta_fast, rope_beta_slow, rope_attn_factor, mla_softmax_mscale;
  int32_t reserved[4];
} bz_model_config;

/* ---- errors / device ------------------------------------------------------------------------------- */
const char* bz_last_error(void);
int bz_abi_version(void);
/* boostr::CudaDevice::new(id) + CudaClient::new(device) (cli/run.rs:70-81) */
int bz_device_open(int device_id, bz_device** out);
int bz_device_close(bz_device* dev);
int bz_device_synchronize(bz_device* dev);
/* CudaDevice::memory_info() (cli/serve.rs:61) */
int bz_device_memory_info(bz_device* dev, size_t* free_bytes, size_t* total_bytes);
int bz_device_name(bz_device* dev, char* buf, size_t n);
/* raw hipStream_t of the handle (for callers that time with HIP events on the launch stream) */
void* bz_device_stream(bz_device* dev);

/* ---- tensors (Tensor<R>::from_slice / zeros / to_vec / record_event / to_vec_pipelined) --------------- */
int bz_tensor_from_host(bz_device* dev, int dtype, const int64_t* shape, int ndim, const void* host, bz_tensor** out);
int bz_tensor_zeros(bz_device* dev, int dtype, const int64_t* shape, int ndim, bz_tensor** out);
int bz_tensor_free(bz_tensor* t);
int bz_tensor_to_host(const bz_tensor* t, void* host, size_t bytes);           /* Tensor::to_vec (sampling.rs:49) */
int bz_tensor_nbytes(const bz_tensor* t, size_t* out);
int bz_tensor_copy_from_host(bz_tensor* t, const void* host, size_t bytes);
/* record_event() -> u64 ; to_vec_pipelined(event) (executor_cache.rs:199-204): event-synchronised D2H on a copy stream */
int bz_event_record(bz_device* dev, uint64_t* event_out);
int bz_event_sync(bz_device* dev, uint64_t event);
int bz_tensor_to_host_pipelined(const bz_tensor* t, uint64_t event, void* host, size_t bytes);

/* ---- model construction (VarMap::insert / insert_decomposed_quant / from_gguf -> LoadedModel::load) --- */
int bz_model_create(bz_device* dev, const bz_model_config* cfg, bz_model** out);
int bz_model_free(bz_model* m);
/* VarMap::insert(name, tensor) (awq.rs:104, regular.rs:89-117). HF tensor names; shape [N,K] or [N]. dtype F32/F16/BF16. */
int bz_model_add_dense(bz_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* host);
/* DecomposedQuantTensor::new(qweight u32[K,N/8], scales f32[K/gs,N], qzeros f32[K/gs,N], None, Awq{gs}, [N,K]) (awq.rs:190-225) */
int bz_model_add_awq(bz_model* m, const char* name, int64_t N, int64_t K, const uint32_t* qweight, const float* scales,
                     const float* zeros, int group_size);
/* DecomposedQuantTensor::new(qweight u32[K/8,N], scales f32[G,N], qzeros u32[G,N/8], g_idx i32[K]?, Gptq{gs}, [N,K]) + bias f32[N]? (gptq.rs:198-259) */
int bz_model_add_gptq(bz_model* m, const char* name, int64_t N, int64_t K, const uint32_t* qweight, const float* scales,
                      const uint32_t* qzeros, const int32_t* g_idx, const float* bias, int group_size);
/* VarMap::from_gguf (gguf.rs:33): raw ggml blocks of one tensor, N rows of K weights */
int bz_model_add_gguf(bz_model* m, const char* name, int ggml_type, int64_t N, int64_t K, const void* raw_blocks);
/* LoadedModel::load(&config.model, &mut vb) (awq.rs:133-136): validates names/shapes, repacks weights into the
 * kernels' HBM layout, builds RoPE tables and workspaces.  Host copies passed to add_* may be freed afterwards. */
int bz_model_finalize(bz_model* m);
/* LoadedModel::{num_layers, num_kv_heads, head_dim, hidden_size, vocab_size, needs_kv_cache, needs_ssm_state} */
int bz_model_get_config(const bz_model* m, bz_model_config* out);
/* bytes of weights resident in HBM after repack, and the algorithmic bytes one decoded token streams */
int bz_model_weight_bytes(const bz_model* m, size_t* resident, size_t* per_token_stream);

/* ---- checkpoint ingestion (the loader either side of LoadedModel::load) ---------------- */
enum { BZ_FORMAT_SAFETENSORS = 0, BZ_FORMAT_GGUF = 1 };          /* loader/detect.rs:9-15 ModelFormat */
typedef struct { int32_t format; int32_t has_config; char weights_path[1024]; char config_path[1024]; } bz_model_source;   /* detect.rs:17-26 ModelSource */
/* detect_model_source(path) (loader/detect.rs:34-150): a .safetensors / .gguf file, or a directory (model.safetensors, pytorch_model.safetensors,
 * model-00001-of-*.safetensors, *.gguf in that order; SafeTensors preferred over GGUF); config.json / .yaml / .yml next to the weights. */
int bz_detect_model_source(const char* path, bz_model_source* out);
enum { BZ_LAYER_TRANSFORMER = 0, BZ_LAYER_MAMBA2 = 1, BZ_LAYER_MAMBA3 = 2, BZ_LAYER_MLA_MOE = 3, BZ_LAYER_MLA_MLP = 4 };   /* detection::LayerType */
typedef struct { int32_t format; /* 0 HuggingFace ("model." prefix), 1 Oxidizr */ int32_t num_layers, tie_word_embeddings; uint8_t layer_types[512]; } bz_detected_arch;
/* boostr::model::detection::detect_architecture_from_names as the reference's tests pin it (loader/safetensors/detect_arch.rs:200-315) */
int bz_detect_architecture_from_names(const char* const* names, int n, bz_detected_arch* out);
typedef struct { int32_t quant_method; /* 0 none, 1 awq, 2 gptq */ int32_t group_size; int32_t torch_dtype; /* BZ_* or -1 */ } bz_quant_info;
/* HuggingFaceConfig::from_json(..).to_universal() + detect_dtype_from_config (loader/safetensors/config.rs:14-70,83-95): HF config.json text ->
 * config POD; quantization_config (detect_arch.rs:79-90,118-131,146-196) -> quant info.  AWQ / GPTQ force f16 (awq.rs:69-71). */
int bz_config_from_hf_json(const char* json_text, bz_model_config* cfg, bz_quant_info* quant);
typedef struct { char architecture[64]; int32_t n_tensors, version, dominant_ggml_type, is_mla, is_moe, is_ssm; uint64_t file_size_bytes; } bz_gguf_info;
/* config_from_gguf_metadata + get_gguf_info (loader/gguf.rs:101-306,309-346): GGUF metadata -> config POD (inference dtype f32, gguf.rs:305) */
int bz_config_from_gguf(const char* path, bz_model_config* cfg, bz_gguf_info* info);
/* SafeTensorsLoader::{tensor_names, tensor_info, is_sharded, num_shards, total_size} (regular.rs:38-61) as one JSON document */
int bz_safetensors_describe(const char* path, char* json_out, size_t cap, size_t* needed);
/* loaders.rs load_model -> regular.rs:20-86 / awq.rs:40-137 / gptq.rs:40-137 / gguf.rs:20-44: detect, configure, add every tensor, finalize */
int bz_load_model(bz_device* dev, const char* path, bz_model** out, bz_model_config* cfg_out);

/* ---- inference state ---------------------------------------------------------------------------------- */
/* LayeredKvCache::new_positional(layers,batch,kv_heads,initial_capacity,max_seq_len,head_dim,dtype,device) (executor_generate.rs:350-353) */
int bz_kv_create(bz_device* dev, int layers, int batch, int n_kv_heads, int initial_capacity, int max_seq_len, int head_dim,
                 int dtype, bz_kv** out);
int bz_kv_free(bz_kv* kv);
int bz_kv_reset(bz_kv* kv);
int bz_kv_seq_len(const bz_kv* kv);                    /* LayeredKvCache::seq_len() (executor_generate.rs:371) */
/* debug/test: copy K or V rows [0,len) of (layer, kv_head) to host as f32 [len][head_dim] */
int bz_kv_read(const bz_kv* kv, int layer, int kv_head, int which /*0=K,1=V*/, int len, float* host);
/* LayeredPagedKvCache::new(layers,num_blocks,block_size,kv_heads,head_dim,dtype,device) (executor_generate.rs:208-210) */
int bz_paged_kv_create(bz_device* dev, int layers, int num_blocks, int block_size, int n_kv_heads, int head_dim, int dtype,
                       bz_paged_kv** out);
int bz_paged_kv_free(bz_paged_kv* kv);
int bz_paged_kv_set_seq_len(bz_paged_kv* kv, int seq_len);  /* set_seq_len (executor_generate.rs:242,286) */
int bz_paged_kv_seq_len(const bz_paged_kv* kv);

/* LayeredSsmState::new(layers, batch, mamba_config, dtype, device) (executor_generate.rs:131-133): recurrent state
 * [layers][n_heads][head_dim][d_state] in `dtype` + conv window [layers][conv_dim][k-1] (docs/architecture.md:52-54) */
int bz_ssm_state_create(bz_model* m, int batch, int dtype, bz_ssm_state** out);
int bz_ssm_state_free(bz_ssm_state* s);
int bz_ssm_state_reset(bz_ssm_state* s);

/* ---- forward ------------------------------------------------------------------------------------------ */
/* process_decode_batch (engine/batch_decode.rs:35-150): N sequences, one new token each, one shared paged cache; slot_mapping I32[N],
 * block_table I32[N, max_blocks] (rows padded with 0), seq_lens host i32[N] (length of each sequence including the new token).
 * logits_out F32 [N, vocab].  int4 (no act-order) and dense 16-bit models share the weights across the batch (multi-row dot4 kernel up to 8
 * sequences, matrix-core GEMMs beyond); other formats run the sequences one after another. */
int bz_forward_paged_batch(bz_model* m, const bz_tensor* tokens, int N, bz_paged_kv* kv, const bz_tensor* slot_mapping, const bz_tensor* block_table,
                           int max_blocks, const int32_t* seq_lens, bz_tensor* logits_out);
/* LoadedModel::forward_with_ssm_state(&input, &mut ssm) (executor_generate.rs:137,148): Mamba2; tokens I64 [1,S].  S >= 8 on a dense 16-bit
 * model takes the batched prefill (matrix-core GEMMs + in-kernel scan over the tokens); the state it leaves continues like the per-token one. */
int bz_forward_ssm(bz_model* m, const bz_tensor* tokens, int S, bz_ssm_state* state, bz_tensor* logits_out, uint32_t flags);
#define BZ_FWD_ALL_LOGITS 1u  /* logits for all S positions ([S,V]); default: last position only ([1,V]) */
/* LoadedModel::forward_with_kv_cache(&input,&mut kv,position) (executor_generate.rs:357,372).
 * tokens: I64 [1,S] device tensor; logits_out: F32 [S or 1, vocab] device tensor (values rounded to act dtype). */
int bz_forward_kv(bz_model* m, const bz_tensor* tokens, int S, bz_kv* kv, int position, bz_tensor* logits_out, uint32_t flags);
/* LoadedModel::forward_with_paged_kv_cache(input,cache,slot_mapping,block_table,seq_len_k,start_pos) (executor_generate.rs:259-262,289-292).
 * slot_mapping I32 [S], block_table I32 [1,n_table] device tensors. */
int bz_forward_paged(bz_model* m, const bz_tensor* tokens, int S, bz_paged_kv* kv, const bz_tensor* slot_mapping,
                     const bz_tensor* block_table, int n_table, int seq_len_k, int start_pos, bz_tensor* logits_out, uint32_t flags);
/* forward_embed / forward_layers_range(hidden, prev_mlp, kv, start, end, position) / forward_head(hidden, prev_mlp)
 * (cli/swarm_forward.rs:205,239-263; executor_multimodal.rs:263-268).  hidden/prev_mlp: F32 [S,hidden] device tensors;
 * has_prev: in/out flag (prev_mlp = Option<Tensor>).  head(layers(embed(x))) == forward_kv(x) bit-for-bit. */
int bz_forward_embed(bz_model* m, const bz_tensor* tokens, int S, bz_tensor* hidden_out);
int bz_forward_layers_range(bz_model* m, bz_tensor* hidden, bz_tensor* prev_mlp, int* has_prev, int S, bz_kv* kv, int start,
                            int end, int position);
int bz_forward_head(bz_model* m, const bz_tensor* hidden, const bz_tensor* prev_mlp, int has_prev, int S, bz_tensor* logits_out,
                    uint32_t flags);

/* ---- sampling ----------------------------------------------------------------------------------------- */
/* SamplingOps::logits_to_token(logits, ids, cnts, n, repeat, freq, presence, temperature, top_k, top_p, min_p, seed)
 * -> I64[1] on device (engine/sampling.rs:445-460).  logits F32 [rows,vocab]: the LAST row is used (narrow).
 * ids I64[n], cnts I32[n] device tensors (may be NULL when n == 0). */
int bz_logits_to_token(bz_device* dev, const bz_tensor* logits, int64_t rows, int64_t vocab, const bz_tensor* ids,
                       const bz_tensor* cnts, int n, float repeat_penalty, float freq_penalty, float presence_penalty,
                       float temperature, int top_k, float top_p, float min_p, uint64_t seed, bz_tensor* token_out);
/* decode_graph::argmax_to_buf(client, logits, next_token_buf) (cuda_graphs.rs:107,128); argmax_on_gpu (executor_cache.rs:189-196) */
int bz_argmax_to_buf(bz_device* dev, const bz_tensor* logits, int64_t rows, int64_t vocab, bz_tensor* token_out);

/* ---- whole-step graph (Runtime::capture_graph + DecodeGraph, cuda_graphs.rs:97-189) --------------------- */
/* Captures one greedy decode step {embed(token_buf) -> layers -> head -> argmax -> token_buf, ++position} as a hipGraph
 * over stable buffers with a device-resident position.  The cache must already hold the prefill. */
int bz_decode_graph_capture(bz_model* m, bz_kv* kv, bz_decode_graph** out);
int bz_decode_graph_capture_paged(bz_model* m, bz_paged_kv* kv, int max_blocks, bz_decode_graph** out);
int bz_decode_graph_capture_ssm(bz_model* m, bz_ssm_state* state, bz_decode_graph** out);
/* DecodeGraph::seed_next_token (cuda_graphs.rs:149-163): first input token + its position */
int bz_decode_graph_seed(bz_decode_graph* g, int64_t token, int position);
/* paged only: block table for the sequence (host i32[n]) -- slot_mapping is derived on device from position */
int bz_decode_graph_set_block_table(bz_decode_graph* g, const int32_t* block_table, int n);
/* DecodeGraph::pre_replay_and_launch (cuda_graphs.rs:166-170): one graph launch = one token */
int bz_decode_graph_replay(bz_decode_graph* g);
/* event-pipelined read of the token produced by replay number `step` (0-based since seed) */
int bz_decode_graph_read_token(bz_decode_graph* g, int64_t step, int64_t* token_out);
/* last logits of the most recent replay (F32 [vocab]) copied to host */
int bz_decode_graph_read_logits(bz_decode_graph* g, float* host, size_t n);
int bz_decode_graph_free(bz_decode_graph* g);

/* ---- batched decode graph (Executor::capture_batched_graph / replay_batched_graph + BatchedGraphState, cuda_graphs_batched.rs:43-257) --------
 * ONE hipGraph decodes one token for N sequences over a shared paged cache: the weight-sharing multi-row step of bz_forward_paged_batch between
 * two bookkeeping kernels.  Stable-address device buffers as in BatchedGraphState (token_buf [N], slot_mapping [N], block_table [N, max_blocks],
 * next_token_buf [N]); beyond the reference (one shared seq_len_k) every sequence keeps its own device-resident position, the argmax
 * (batch_argmax_to_buf) is fed back on the device and the slot is derived from the block table, so consecutive replays need no host work.
 * Llama-family models that take the multi-row step (int4 without act-order, or dense 16-bit; 16-bit lm_head); 2 <= N <= 512. */
typedef struct bz_batch_graph bz_batch_graph;
int bz_decode_batch_graph_capture(bz_model* m, bz_paged_kv* kv, int N, int max_blocks, bz_batch_graph** out);
/* state before the first replay: tokens[i] = the token sequence i feeds next, seq_lens[i] = its length INCLUDING that token (batch_decode.rs:79-88),
 * block_table = host I32 [N, max_blocks] */
int bz_decode_batch_graph_seed(bz_batch_graph* g, const int64_t* tokens, const int32_t* seq_lens, const int32_t* block_table);
/* new block-table rows (a sequence is about to cross into a block the device table does not hold yet) */
int bz_decode_batch_graph_set_block_table(bz_batch_graph* g, const int32_t* block_table);
int bz_decode_batch_graph_replay(bz_batch_graph* g);
/* the N greedy tokens produced by replay `step` (0-based since the seed); waits for the device */
int bz_decode_batch_graph_read_tokens(bz_batch_graph* g, int64_t step, int64_t* tokens_out);
/* logits F32 [N, vocab] of the last replay: a device tensor owned by the graph (do not free) */
int bz_decode_batch_graph_logits(bz_batch_graph* g, bz_tensor** logits_out);
int bz_decode_batch_graph_free(bz_batch_graph* g);

/* ---- host decode loop (Executor::generate contiguous branch, executor_generate.rs:341-410) ---------------- */
typedef struct {
  int32_t max_tokens;
  float   temperature;       /* 0 => greedy (generation.rs:262-264) */
  float   repeat_penalty;    /* reference default 1.1 (generation.rs:164-166); 1.0 disables */
  int32_t repeat_last_n;     /* 64 (commands.rs:40) */
  float   frequency_penalty, presence_penalty;
  int32_t top_k; float top_p, min_p; uint64_t seed;
  int64_t eos_id;            /* -1: none */
  int32_t use_graph;         /* --graphs (cli/run.rs:144-157): greedy only, penalties ignored as in the reference */
  int32_t paged;             /* --paged-attention */
  int32_t block_size;        /* 16 (inference.rs:189-191) */
  /* host-side sampler options (config/generation.rs:190-222; applied as sampling.rs:393-437 does: DRY, typical, logit bias, dynatemp, mirostat) */
  float   dry_multiplier;    /* 0 disables */
  int32_t dry_base;          /* default 2 */
  int32_t dry_allowed_length;
  float   typical_p;         /* 0 disables */
  float   dynatemp_range;    /* 0 disables */
  float   dynatemp_exponent; /* default 1.0 */
  int32_t mirostat_mode;     /* >= 2: Mirostat v2 replaces logits_to_token (sampling.rs:96-110) */
  float   mirostat_tau, mirostat_eta;
  int32_t n_logit_bias; const uint32_t* logit_bias_ids; const float* logit_bias_vals;
  int32_t reserved[4];
} bz_gen_config;
/* prefill_ms / decode_ms: host wall time of the prompt phase and of everything after it.  The other timing fields are the reference bench's
 * (/root/reference/src/cli/bench.rs:142-160,285-306), measured where its stream consumer measures them -- at the moment a token id has reached the host:
 *   ttft_ms = start -> first token; itl_*: time between consecutive tokens (p50 / p99 / max over the n_generated - 1 gaps, nearest-rank percentiles);
 *   total_ms = start -> last token; decode_tok_per_s = (n_generated - 1) / (total - ttft). */
typedef struct { double prefill_ms, decode_ms; int32_t n_generated; int32_t finish_reason; /* 0 length, 1 eos */
                 double ttft_ms, total_ms, itl_p50_ms, itl_p99_ms, itl_max_ms, decode_tok_per_s; } bz_gen_stats;
/* ---- host-side sampler pieces, each a line-for-line restatement of the reference's Rust (they run on the CPU there too) ---------------- */
float bz_compute_dynamic_temperature(const float* logits, int64_t vocab, float base, float range, float exponent);      /* sampling.rs:41-86 */
int bz_apply_dry_penalty(float* logits, int64_t vocab, const uint32_t* recent, int64_t n_recent, float multiplier, int base, int allowed_length);   /* :270-320 */
int bz_apply_typical_filter(float* logits, int64_t vocab, float typical_p);                                              /* :322-369 */
int bz_apply_logit_bias(float* logits, int64_t vocab, const uint32_t* ids, const float* bias, int n);                   /* :464-480 */
int bz_compute_logprobs(const float* logits, int64_t vocab, uint32_t chosen, int top_n, float* chosen_logprob, uint32_t* top_ids, float* top_logprobs,
                        int* n_top);                                                                                     /* :197-256 */
typedef struct bz_mirostat bz_mirostat;                                                                                  /* mirostat.rs:12-17 MirostatState */
int bz_mirostat_create(float tau, float eta, uint64_t seed, bz_mirostat** out);                                          /* mirostat.rs:19-34 */
int bz_mirostat_sample(bz_mirostat* s, const float* logits, int64_t vocab, float temperature, uint32_t* token, float* logprob);   /* :41-110 */
float bz_mirostat_mu(const bz_mirostat* s);
int bz_mirostat_free(bz_mirostat* s);
/* prompt: host i64[n_prompt]; out_tokens: host i64[max_tokens] */
int bz_generate(bz_model* m, const int64_t* prompt, int n_prompt, const bz_gen_config* gc, int64_t* out_tokens, bz_gen_stats* stats);

/* ---- measurement (methodology of /root/reference/src/cli/bench.rs:24-33,299-306) ----------------------- */
typedef struct { char name[48]; int32_t launches; double total_ms; double algo_bytes; } bz_kernel_time;
/* Runs `iters` eager decode steps (token at position, position+1, ...) with every kernel launched through
 * hipExtLaunchKernelGGL start/stop events on the compute stream and returns, per kernel, launches / summed dispatch time /
 * summed algorithmic bytes.  Durations are pure kernel times (comparable with rocprofv3 --kernel-trace). */
int bz_profile_step(bz_model* m, bz_kv* kv, int64_t token, int position, int iters, bz_kernel_time* out, int max_out, int* n_out);
/* the same for a Mamba2 model (advances `state` by `iters` tokens) */
int bz_profile_step_ssm(bz_model* m, bz_ssm_state* state, int64_t token, int iters, bz_kernel_time* out, int max_out, int* n_out);

/* Kernel tuning aid: mean dispatch time of the int4 GEMV kernel alone on synthetic [N,K] gs-128 weights rotated over `nbuf`
 * HBM buffers.  mode 0 plain x / 1 fused residual+RMSNorm prologue / 2 SiLU*up prologue; flags are debugging knobs (0). */
int bz_tune_gemv(bz_device* dev, int N, int K, int groups_per_wg, int mode, int nbuf, int iters, int flags, double* avg_us);

/* the same for the fused MLP kernel (norm + gate/up + SiLU*up + down) on synthetic int4 weights; stamps_out (optional, 2 x 16 x 16 values) receives the
 * diagnostic build's per-wave phase stamps in 10 ns units */
int bz_tune_mlp(bz_device* dev, int H, int I, int nbuf, int iters, int flags, double* avg_us, long long* stamps_out);

/* the same for the dense row GEMV (16-bit weights [N,K], wdt BZ_F16 / BZ_BF16); mode 0 plain, 1 residual + RMSNorm prologue, 2 SiLU*up prologue;
 * sk = split-K count (0: the loader's choice) */
int bz_tune_rows(bz_device* dev, int N, int K, int wdt, int mode, int sk, int nbuf, int iters, double* avg_us);

/* Measured HBM read ceiling of this device, GB/s: a streaming read of `bytes` (rotating buffers beyond the Infinity Cache) with the decode
 * kernels' load pattern and no arithmetic.  bench.py reports roofline fractions against the 8 TB/s spec peak AND against this number
 * (SURVEY.md 8d "record the measured peak on the box and report against both"). */
int bz_probe_hbm_read(bz_device* dev, size_t bytes, int iters, double* gbs);

/* ---- op-level entry points (parity tests; each is the kernel the forward path uses) ------------------------ */
/* QuantMatmulOps / dense matmul on a registered weight `name` ("….weight"): y[S,N] = x[S,K] W^T (+bias); x,y F32 device tensors */
int bz_quant_matmul(bz_model* m, const char* name, const bz_tensor* x, int S, bz_tensor* y);
/* Prefill GEMM on the matrix cores: y[S,N] = round16(x)[S,K] . W[N,K]^T, f32 accumulate, result unrounded.  Dense f16 / bf16 weights
 * (K % 64 == 0, x rounded to the weight dtype), or int4 group-quantised weights without act-order (AWQ / GPTQ: x rounded to f16, S >= 9).  The same kernel runs inside bz_forward_kv / bz_forward_paged when a dense 16-bit Llama-family model is given S >= 8 tokens
 * (regular.rs:89-117 bf16 SafeTensors path; boostr's matmul behind LoadedModel::forward_with_kv_cache at executor_generate.rs:357). */
int bz_prefill_matmul(bz_model* m, const char* name, const bz_tensor* x, int S, bz_tensor* y);
/* DequantOps: whole weight -> F32 [N,K] on host (from the REPACKED device layout: validates the repack) */
int bz_dequant(bz_model* m, const char* name, float* host_out);
/* NormalizationOps::rms_norm with optional fused residual: h' = round(h + prev) ; y = w * round(h' * rsqrt(mean(h'^2)+eps)) */
int bz_rms_norm(bz_device* dev, const bz_tensor* x, const bz_tensor* prev /*nullable*/, const bz_tensor* weight, int rows, int n,
                float eps, int act_dtype, bz_tensor* y, bz_tensor* h_out /*nullable*/);
/* RoPE on [S, n_heads, head_dim] F32 in place, positions position..position+S-1, using the model's cos/sin caches (rope_caches()) */
int bz_rope(bz_model* m, bz_tensor* x, int S, int n_heads, int position);
/* ActivationOps/BinaryOps: y = round(round(silu(gate)) * up) */
int bz_silu_mul(bz_device* dev, const bz_tensor* gate, const bz_tensor* up, int64_t n, int act_dtype, bz_tensor* y);
/* single-token attention of q F32 [n_heads, head_dim] over layer `layer` of the cache, first `len` positions */
int bz_attn_decode(bz_model* m, const bz_tensor* q, bz_kv* kv, int layer, int len, bz_tensor* out);
int bz_paged_attn_decode(bz_model* m, const bz_tensor* q, bz_paged_kv* kv, int layer, const bz_tensor* block_table, int len,
                         bz_tensor* out);
/* kv_insert kernel (cuda_graphs.rs:5): write k,v F32 [n_kv_heads, head_dim] at `position` (contiguous) */
int bz_kv_insert(bz_model* m, bz_kv* kv, int layer, int position, const bz_tensor* k, const bz_tensor* v);
/* ConvOps (engine/executor.rs:71) -- Mamba2 layer `layer`: depthwise causal conv1d STEP (+ bias, SiLU) over the raw in_proj row zx = [z | x B C | dt]
 * (F32 device, 2 d_inner + 2 n_groups d_state + n_heads values): xbc_out [d_inner + 2 n_groups d_state] F32; the layer's conv window inside `state`
 * (LayeredSsmState, docs/architecture.md:52-54: conv [conv_dim, k-1]) moves on by this token; the SSM state is not touched. */
int bz_conv1d_step(bz_model* m, int layer, bz_ssm_state* state, const bz_tensor* zx, bz_tensor* xbc_out);
/* The Mamba2 mixer's step as the decode path runs it (one launch: conv1d step + SiLU, h = exp(dt A) h + dt B x, y = C h + D x, gate y * silu(z)):
 * y_out [d_inner] F32 (before the gated RMSNorm); `state` (conv window + SSM state [n_heads, head_dim, d_state]) advances by one token
 * (forward_with_ssm_state, engine/executor_generate.rs:137,148). */
int bz_ssm_step(bz_model* m, int layer, bz_ssm_state* state, const bz_tensor* zx, bz_tensor* y_out);
/* One layer of a LayeredSsmState copied to the host as F32 (which = 0: SSM state [n_heads * head_dim * d_state], 1: conv window [conv_dim * (k-1)]);
 * n = the expected element count.  Test / debugging aid for the two entry points above (the state layouts are docs/architecture.md:52-54). */
int bz_ssm_state_read(const bz_ssm_state* state, int layer, int which, float* host, size_t n);
/* MoE router of DeepSeek-V2 layer `layer` (docs/architecture.md:108-119: softmax -> greedy top-k -> weights): hidden = the residual stream row [H] F32,
 * the layer's post-attention RMSNorm runs inside (as in the decode step).  sel_out I32 / w_out F32 [top_k + n_shared]: the routed experts in selection
 * order (ties -> lowest index), then the shared experts' slots (index n_experts + j, weight 1); xn_out (optional) F32 [H] the normalised row. */
int bz_moe_route(bz_model* m, int layer, const bz_tensor* hidden, bz_tensor* sel_out, bz_tensor* w_out, bz_tensor* xn_out);
/* Grouped expert GEMV over the stacked expert weights (engine/executor_cache.rs:218-219,344-348): slot s uses expert sel[s] (I32 device, n_slots <= 128).
 * which = 0: gate|up projections, x F32 [H] shared by all slots -> y F32 [n_slots][2 moe_inter];
 * which = 1: down projection with the SiLU(gate) * up prologue, x F32 [n_slots][2 moe_inter] -> y F32 [n_slots][H]. */
int bz_moe_grouped_gemv(bz_model* m, int layer, int which, const bz_tensor* sel, int n_slots, const bz_tensor* x, bz_tensor* y);
/* The exponential every kernel of this library uses (SiLU, softmax weights, sampling): ONE specified sequence of IEEE operations (range reduction by ln2
 * in two fma steps, degree-5 polynomial, exact 2^n scaling; < 1 ulp), evaluated here on the HOST for n values -- the same function body the device code
 * compiles, so a CPU-only test can pin it bit for bit against an independent restatement.  boostr's own exp (ActivationOps / softmax behind
 * executor.rs:67-80) is not visible; any faithful expf is "the" exp, and a specified one makes both sides of a parity test compute the same bits. */
int bz_expf_spec(const float* x, int n, float* y);
/* rope_caches() -> (cos, sin) F32 [max_pos, head_dim/2] copied to host */
int bz_rope_caches(bz_model* m, float* cos_host, float* sin_host);

#ifdef __cplusplus
}
#endif
#endif
