// bz_host.hip -- host side of libblazr_hip.so: device/tensor handles, model construction + repack,
// the Llama-family decode step built from the fused kernels, hipGraph capture, and the decode loop.
// C-ABI entry points are declared in include/blazr_hip.h (each cites the reference interface it replaces).
#include "bz_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <chrono>
#include <numeric>
#include <unordered_map>

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static bool bz_trace_on() { static int t = -1; if (t < 0) t = getenv("BZ_TRACE") ? 1 : 0; return t == 1; }
#define BZ_TRACE(...) do { if (bz_trace_on()) { fprintf(stderr, "[bz] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
void bz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* bz_last_error(void) { return g_err; }
extern "C" int bz_abi_version(void) { return BZ_ABI_VERSION; }

size_t bz_dtype_size(int dt) {
  switch (dt) {
    case BZ_F32: case BZ_I32: case BZ_U32: return 4;
    case BZ_F16: case BZ_BF16: return 2;
    case BZ_I64: return 8;
    case BZ_U8: return 1;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------
extern "C" int bz_device_open(int id, bz_device** out) {
  BZ_API_BEGIN
  if (!out) BZ_FAIL(BZ_E_INVALID, "bz_device_open: out is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) BZ_FAIL(BZ_E_NODEVICE, "no HIP device available (%s); libblazr_hip has no CPU fallback", hipGetErrorString(e));
  if (id < 0 || id >= n) BZ_FAIL(BZ_E_INVALID, "device id %d out of range [0,%d)", id, n);
  BZ_HIP(hipSetDevice(id));
  bz_device* d = new bz_device();
  d->id = id;
  BZ_HIP(hipGetDeviceProperties(&d->prop, id));
  if (strncmp(d->prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = d->prop.gcnArchName;
    delete d;
    BZ_FAIL(BZ_E_NODEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", id, arch.c_str());
  }
  BZ_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  BZ_HIP(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
  d->pinned_bytes = 1 << 20;
  BZ_HIP(hipHostMalloc(&d->pinned, d->pinned_bytes, hipHostMallocDefault));
  BZ_HIP(hipMalloc((void**)&d->scratch, 4096));
  BZ_HIP(hipMalloc((void**)&d->persist_bar, bzk_persist_bar_words() * 4));
  BZ_HIP(hipMemset(d->persist_bar, 0, bzk_persist_bar_words() * 4));
  { void* pe = nullptr; BZ_HIP(hipHostMalloc(&pe, 64, hipHostMallocMapped)); d->persist_err = (volatile unsigned*)pe; *d->persist_err = 0u; }
  *out = d;
  return BZ_OK;
  BZ_API_END
}
// The device handle is reference counted: children (tensors, models, caches, graphs) keep it alive, so that a host
// language whose destructors run in arbitrary order (GC) cannot free a child against a destroyed stream.
void bz_dev_retain(bz_device* d) { __atomic_add_fetch(&d->refs, 1, __ATOMIC_RELAXED); }
void bz_dev_release(bz_device* d) {
  if (__atomic_sub_fetch(&d->refs, 1, __ATOMIC_ACQ_REL) > 0) return;
  hipSetDevice(d->id);
  hipStreamSynchronize(d->stream);
  for (auto ev : d->events) hipEventDestroy(ev);
  if (d->pinned) hipHostFree(d->pinned);
  if (d->scratch) hipFree(d->scratch);
  if (d->persist_bar) hipFree(d->persist_bar);
  if (d->persist_err) hipHostFree((void*)d->persist_err);
  bzk_sample_free(d->samp_ws);
  hipStreamDestroy(d->stream);
  hipStreamDestroy(d->copy_stream);
  delete d;
}
extern "C" int bz_device_close(bz_device* d) {
  BZ_API_BEGIN
  if (!d) return BZ_OK;
  hipSetDevice(d->id);
  hipStreamSynchronize(d->stream);
  bz_dev_release(d);
  return BZ_OK;
  BZ_API_END
}
// The persistent decode launch (bz_persist.hip) ends with this word set when one of its grid-barrier waits ran into its wall-clock limit (a lost arrival):
// the results of that step are garbage.  Checked wherever the host has just synchronised with the device; the word is sticky until the device is reopened.
static int persist_check(bz_device* d) {
  if (d && d->persist_err && *d->persist_err != 0u)
    BZ_FAIL(BZ_E_HIP, "persistent decode launch: a grid-barrier wait exceeded its limit (results invalid); set BZ_NO_PERSIST=1 to use the launch-per-phase path");
  return BZ_OK;
}
extern "C" int bz_device_synchronize(bz_device* d) {
  BZ_API_BEGIN
  if (!d) BZ_FAIL(BZ_E_INVALID, "null device");
  BZ_HIP(hipStreamSynchronize(d->stream));
  BZ_HIP(hipStreamSynchronize(d->copy_stream));
  return persist_check(d);
  BZ_API_END
}
extern "C" int bz_device_memory_info(bz_device* d, size_t* f, size_t* t) {
  BZ_API_BEGIN
  if (!d) BZ_FAIL(BZ_E_INVALID, "null device");
  BZ_HIP(hipSetDevice(d->id));
  BZ_HIP(hipMemGetInfo(f, t));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_device_name(bz_device* d, char* buf, size_t n) {
  BZ_API_BEGIN
  if (!d || !buf) BZ_FAIL(BZ_E_INVALID, "null argument");
  snprintf(buf, n, "%s (%s, %d CUs)", d->prop.name, d->prop.gcnArchName, d->prop.multiProcessorCount);
  return BZ_OK;
  BZ_API_END
}
extern "C" void* bz_device_stream(bz_device* d) { return d ? (void*)d->stream : nullptr; }

// pinned staging: returns a host pointer valid until ~pinned_bytes more have been requested
static void* pinned_slot(bz_device* d, size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  if (d->pinned_off + bytes > d->pinned_bytes) d->pinned_off = 0;
  void* p = (char*)d->pinned + d->pinned_off;
  d->pinned_off += bytes;
  return p;
}

// ---------------------------------------------------------------------------------------------------------
// tensors
// ---------------------------------------------------------------------------------------------------------
static int tensor_alloc(bz_device* dev, int dtype, const int64_t* shape, int ndim, bz_tensor** out) {
  if (!dev || !out || ndim < 0 || ndim > 8) BZ_FAIL(BZ_E_INVALID, "tensor: bad arguments");
  size_t es = bz_dtype_size(dtype);
  if (!es) BZ_FAIL(BZ_E_INVALID, "tensor: bad dtype %d", dtype);
  size_t n = 1;
  for (int i = 0; i < ndim; i++) { if (shape[i] < 0) BZ_FAIL(BZ_E_INVALID, "tensor: negative dim"); n *= (size_t)shape[i]; }
  bz_tensor* t = new bz_tensor();
  t->dev = dev; t->dtype = dtype; t->shape.assign(shape, shape + ndim); t->nbytes = n * es;
  BZ_HIP(hipSetDevice(dev->id));
  hipError_t e = hipMalloc(&t->ptr, std::max<size_t>(t->nbytes, 16));
  if (e != hipSuccess) { delete t; BZ_FAIL(BZ_E_OOM, "hipMalloc(%zu) failed: %s", n * es, hipGetErrorString(e)); }
  bz_dev_retain(dev);
  *out = t;
  return BZ_OK;
}
extern "C" int bz_tensor_from_host(bz_device* dev, int dtype, const int64_t* shape, int ndim, const void* host, bz_tensor** out) {
  BZ_API_BEGIN
  BZ_TRY(tensor_alloc(dev, dtype, shape, ndim, out));
  if ((*out)->nbytes && host) {
    BZ_HIP(hipMemcpyAsync((*out)->ptr, host, (*out)->nbytes, hipMemcpyHostToDevice, dev->stream));
    BZ_HIP(hipStreamSynchronize(dev->stream));
  }
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_zeros(bz_device* dev, int dtype, const int64_t* shape, int ndim, bz_tensor** out) {
  BZ_API_BEGIN
  BZ_TRY(tensor_alloc(dev, dtype, shape, ndim, out));
  if ((*out)->nbytes) BZ_HIP(hipMemsetAsync((*out)->ptr, 0, (*out)->nbytes, dev->stream));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_free(bz_tensor* t) {
  BZ_API_BEGIN
  if (!t) return BZ_OK;
  if (t->owned && t->ptr) { hipStreamSynchronize(t->dev->stream); hipFree(t->ptr); }
  bz_dev_release(t->dev);
  delete t;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_nbytes(const bz_tensor* t, size_t* out) {
  BZ_API_BEGIN
  if (!t || !out) BZ_FAIL(BZ_E_INVALID, "null argument");
  *out = t->nbytes;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_to_host(const bz_tensor* t, void* host, size_t bytes) {
  BZ_API_BEGIN
  if (!t || !host) BZ_FAIL(BZ_E_INVALID, "null argument");
  if (bytes > t->nbytes) BZ_FAIL(BZ_E_INVALID, "to_host: %zu bytes requested, tensor holds %zu", bytes, t->nbytes);
  BZ_HIP(hipMemcpyAsync(host, t->ptr, bytes, hipMemcpyDeviceToHost, t->dev->stream));
  BZ_HIP(hipStreamSynchronize(t->dev->stream));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_copy_from_host(bz_tensor* t, const void* host, size_t bytes) {
  BZ_API_BEGIN
  if (!t || !host) BZ_FAIL(BZ_E_INVALID, "null argument");
  if (bytes > t->nbytes) BZ_FAIL(BZ_E_INVALID, "copy_from_host: %zu bytes given, tensor holds %zu", bytes, t->nbytes);
  BZ_HIP(hipMemcpyAsync(t->ptr, host, bytes, hipMemcpyHostToDevice, t->dev->stream));
  BZ_HIP(hipStreamSynchronize(t->dev->stream));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_event_record(bz_device* d, uint64_t* out) {
  BZ_API_BEGIN
  if (!d || !out) BZ_FAIL(BZ_E_INVALID, "null argument");
  std::lock_guard<std::mutex> dlock__(d->mu);
  // small ring of reusable events (a recycled id still orders after the work it was first recorded behind: waiting on it is conservative)
  const size_t RING = 64;
  if (d->events.size() < RING) {
    hipEvent_t ev;
    BZ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    d->events.push_back(ev);
  }
  const uint64_t id = d->event_counter++ % d->events.size();
  BZ_HIP(hipEventRecord(d->events[id], d->stream));
  *out = id;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_event_sync(bz_device* d, uint64_t ev) {
  BZ_API_BEGIN
  if (!d) BZ_FAIL(BZ_E_INVALID, "bad event");
  hipEvent_t e;
  { std::lock_guard<std::mutex> dlock__(d->mu); if (ev >= d->events.size()) BZ_FAIL(BZ_E_INVALID, "bad event"); e = d->events[ev]; }
  BZ_HIP(hipEventSynchronize(e));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_tensor_to_host_pipelined(const bz_tensor* t, uint64_t ev, void* host, size_t bytes) {
  BZ_API_BEGIN
  if (!t || !host) BZ_FAIL(BZ_E_INVALID, "bad argument");
  if (bytes > t->nbytes) BZ_FAIL(BZ_E_INVALID, "to_host_pipelined: size");
  bz_device* d = t->dev;
  std::lock_guard<std::mutex> dlock__(d->mu);       // one copy stream per device: concurrent readers take turns
  if (ev >= d->events.size()) BZ_FAIL(BZ_E_INVALID, "bad event");
  // the copy waits only for `ev`, on the copy stream: the compute stream keeps running forward(t+1)
  BZ_HIP(hipStreamWaitEvent(d->copy_stream, d->events[ev], 0));
  BZ_HIP(hipMemcpyAsync(host, t->ptr, bytes, hipMemcpyDeviceToHost, d->copy_stream));
  BZ_HIP(hipStreamSynchronize(d->copy_stream));
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------
struct RawTensor {
  int kind = 0;  // 0 dense, 1 awq, 2 gptq, 3 gguf
  int dtype = BZ_F32;
  std::vector<int64_t> shape;
  int64_t N = 0, K = 0; int gs = 0; int ggml_type = 0;
  void* d0 = nullptr; void* d1 = nullptr; void* d2 = nullptr;  // dense data | qweight,scales,zeros/qzeros | blocks
  float* d_bias = nullptr;
  std::vector<int32_t> g_idx;
  size_t bytes = 0;
  bool consumed = false;
};

struct FusedLinear {
  std::vector<LinearDev> parts;
  std::vector<int> n_off;
  int N = 0, K = 0;
  bool fix_out = true;  // parts accumulate into fixed point (else direct f32 store)
};

struct LayerDev {
  float* attn_norm = nullptr; float* ffn_norm = nullptr;
  FusedLinear qkv, o, gateup, down;
  void* down_slabs = nullptr;   // dense 16-bit down_proj, slab-major copy [I / 32][H][32] for the fused MLP (k_mlp_dense); nullptr: not built
};

struct MambaLayerDev {
  float* norm = nullptr; float* conv_w = nullptr; float* conv_b = nullptr; float* dt_bias = nullptr; float* A_log = nullptr; float* D = nullptr;
  float* gnorm = nullptr;
  FusedLinear in_proj, out_proj;
};

struct DsLayerDev {
  float* attn_norm = nullptr; float* ffn_norm = nullptr; float* kv_norm = nullptr;
  FusedLinear qkva, o;                       // [q_proj ; kv_a_proj_with_mqa] (q_lora_rank > 0: [q_a_proj ; kv_a_proj_with_mqa]), o_proj
  float* q_norm = nullptr; FusedLinear q_b;  // q_lora_rank > 0: q_a_layernorm, q_b_proj [n_heads (nope + rope)][q_lora_rank]
  void* kv_b = nullptr; int kv_b_dt = BZ_BF16;   // kv_b_proj [n_heads (nope+v)][rank], as stored
  bool is_moe = false;
  FusedLinear gateup, down;                  // dense layers
  void* router = nullptr; int router_dt = BZ_BF16;   // [E][hidden]
  void* e_gu = nullptr; void* e_dn = nullptr; int e_dt = BZ_BF16;   // stacked [E + n_shared][2 moe_inter][hidden] / [E + n_shared][hidden][moe_inter]
};

struct bz_model {
  // One step's kernels share the model's workspace (residual stream, accumulator ring, logits), so the enqueue of a step is atomic:
  // concurrent generate() calls on one model (engine/scheduler.rs:67, startup.rs:234-236) interleave whole steps, never kernels.
  std::recursive_mutex mu;
  std::vector<DsLayerDev> dlayers;
  float* mla_ws = nullptr; int mla_nsplit = 1;   // MLA decode over context slices: partials [n_heads][nsplit][rank + 2]
  float* mla_scw = nullptr; float* mla_mxw = nullptr;        // exact decode MLA, three-launch form: scores [n_heads][nsplit][ceil(max_seq_len / nsplit) + 1], slice maxima [n_heads][nsplit]
  double* mla_wsd = nullptr; unsigned* mla_sync = nullptr;   // exact decode MLA (k_mla_attn_x): double partials [n_heads][nsplit][rank + 1]; per-head {maximum, arrivals} words
  long long* moe_gu_acc = nullptr;   // fixed-point gate / up of the MoE slots (k_gemv_rows2's MoE form); zeroed by the combine launch
  float* moe_xn = nullptr; float* moe_gu = nullptr; float* moe_out = nullptr; long long* moe_acc = nullptr; int* moe_sel = nullptr; float* moe_w = nullptr; float* moe_lg = nullptr; unsigned* moe_cnt = nullptr;
  // DeepSeek-V2 batched-prefill rows (allocated on first use for dpf_rows prompt rows)
  int dpf_rows = 0; float* dpf_att = nullptr; void* dpf_xg16 = nullptr; float* dpf_gu = nullptr; void* dpf_a16 = nullptr; float* dpf_ye = nullptr; float* dpf_ysh = nullptr;
  int* dpf_sel = nullptr; float* dpf_w = nullptr; int* dpf_cnt = nullptr; int* dpf_off = nullptr; int* dpf_rowof = nullptr; int* dpf_tokof = nullptr;
  std::vector<MambaLayerDev> mlayers;
  float* ybuf = nullptr; float* vss = nullptr;   // Mamba2 workspace
  int mpf_rows = 0; float* mpf_h = nullptr; float* mpf_t = nullptr; float* mpf_zx = nullptr; float* mpf_xbc = nullptr; float* mpf_y = nullptr; float* mpf_vss = nullptr;
  void* mpf_x16 = nullptr;   // Mamba2 batched-prefill rows (allocated on first use)
  bz_device* dev = nullptr;
  bz_model_config cfg;
  bool finalized = false;
  std::unordered_map<std::string, RawTensor> raw;
  std::vector<LayerDev> layers;
  void* embed = nullptr; int embed_dt = BZ_F16;
  float* final_norm = nullptr;
  FusedLinear lm_head;
  std::unordered_map<std::string, LinearDev> named;  // views for the op-level API
  std::vector<void*> owned;                           // device allocations to free
  float* cos_t = nullptr; float* sin_t = nullptr;
  float* rope_cur = nullptr;   // [cos | sin] row of the current position (staged by the embed kernel)
  float* att_ws = nullptr;     // split-KV attention partials (bzk_attn_split_ws_bytes)
  void* persist_tab = nullptr; // device table of per-layer weight pointers for the persistent decode launch (nullptr: the model does not qualify)
  // workspace
  float* hbuf[2] = {nullptr, nullptr};
  // batched-prefill workspace (bz_prefill.hip), allocated on first use for `pf_rows` prompt rows
  int pf_rows = 0; float* pf_h = nullptr; float* pf_t = nullptr; float* pf_qkv = nullptr; float* pf_gu = nullptr; void* pf_x16 = nullptr;
  int* row_pos = nullptr; int row_pos_n = 0;   // device copy of a decode batch's per-row positions
  long long* pf_acc = nullptr;   // int4 multi-row GEMM scratch: 8 rows x widest N, fixed point, kept zero between launches
  // block-format (GGUF) weights on the batched prompt path: per fused linear the power-of-two scale (device float) and, while the budget lasts, the matrix
  // as three f16 pieces per weight [N][3 K] (6 bytes per weight: sized for 288 GB of HBM -- a 7B Q4_K_M model keeps 43 GB of them next to its 4.3 GB of blocks)
  struct GqPf { float* wscale = nullptr; void* w3 = nullptr; };
  std::unordered_map<const FusedLinear*, GqPf> gq_pf;
  size_t gq_cache_bytes = 0;
  unsigned* gq_amax = nullptr; void* gq_w3_scratch = nullptr; size_t gq_w3_scratch_bytes = 0;
  void* pf_x3 = nullptr; float* pf_rscale = nullptr;   // split activation rows [rows][3 xw] f16 + their row scales
  void* pf_xq = nullptr; size_t pf_xq_bytes = 0;       // int8 digit planes + row parameters of the exact integer-MFMA GEMM (bzk_pf_quant_i8)
  float* pf_ws = nullptr; size_t pf_ws_bytes = 0;   // W4A16 MFMA GEMM: split-K partials for short prompts / decode batches
  long long* ring[3] = {nullptr, nullptr, nullptr};
  float* dring[3] = {nullptr, nullptr, nullptr};   // direct-output twins of the ring (ROWS kernels)
  int ring_n = 0;
  float* attn_out = nullptr;
  float* logits = nullptr;
  float* pval = nullptr; int* pidx = nullptr; int nparts = 0;
  float* scratch = nullptr;
  long long* tok_tmp = nullptr;
  int* pos_tmp = nullptr;
  size_t resident = 0, per_token = 0;
};

static int dev_alloc(bz_model* m, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
  if (e != hipSuccess) BZ_FAIL(BZ_E_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  m->owned.push_back(*p);
  return BZ_OK;
}
static int upload(bz_device* d, void** out, const void* host, size_t bytes) {
  hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
  if (e != hipSuccess) BZ_FAIL(BZ_E_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  if (bytes) BZ_HIP(hipMemcpy(*out, host, bytes, hipMemcpyHostToDevice));
  (void)d;
  return BZ_OK;
}

extern "C" int bz_model_create(bz_device* dev, const bz_model_config* cfg, bz_model** out) {
  BZ_API_BEGIN
  if (!dev || !cfg || !out) BZ_FAIL(BZ_E_INVALID, "bz_model_create: null argument");
  if (cfg->abi_version != BZ_ABI_VERSION) BZ_FAIL(BZ_E_INVALID, "config abi_version %d != %d", cfg->abi_version, BZ_ABI_VERSION);
  if (cfg->arch != BZ_ARCH_LLAMA && cfg->arch != BZ_ARCH_MAMBA2 && cfg->arch != BZ_ARCH_DEEPSEEK2) BZ_FAIL(BZ_E_UNSUPPORTED, "arch %d is not implemented", cfg->arch);
  if (cfg->arch == BZ_ARCH_DEEPSEEK2) {
    if (cfg->hidden <= 0 || cfg->n_layers <= 0 || cfg->n_heads <= 0 || cfg->vocab <= 0 || cfg->max_seq_len <= 0 || cfg->mla_kv_lora_rank <= 0 ||
        cfg->mla_nope_dim <= 0 || cfg->mla_rope_dim <= 0 || cfg->mla_v_dim <= 0 || cfg->moe_n_experts < 0 || cfg->moe_first_dense < 0)
      BZ_FAIL(BZ_E_INVALID, "config: non-positive deepseek2 dimension");
    if (cfg->mla_q_lora_rank < 0 || cfg->mla_q_lora_rank % 8) BZ_FAIL(BZ_E_INVALID, "config: q_lora_rank %d must be a non-negative multiple of 8", cfg->mla_q_lora_rank);
    if (cfg->hidden % 8 || cfg->mla_kv_lora_rank % 8 || cfg->mla_kv_lora_rank > 1024 || cfg->mla_rope_dim > 64 || (cfg->mla_rope_dim & 1) || cfg->mla_nope_dim % 4 ||
        cfg->mla_v_dim % 4)
      BZ_FAIL(BZ_E_UNSUPPORTED, "config: MLA dims unsupported (rank %% 8, rank <= 1024, rope <= 64 even, nope/v %% 4)");
    if (cfg->moe_n_experts > 0 && (cfg->moe_top_k <= 0 || cfg->moe_top_k > cfg->moe_n_experts || cfg->moe_top_k > 16 || cfg->moe_n_shared < 0 || cfg->moe_n_shared > 8 ||
                                   cfg->moe_inter <= 0 || cfg->moe_inter % 8 || cfg->moe_n_experts > 1024))
      BZ_FAIL(BZ_E_INVALID, "config: bad MoE configuration");
    if ((cfg->moe_first_dense > 0 || cfg->moe_n_experts == 0) && (cfg->inter <= 0 || cfg->inter % 8)) BZ_FAIL(BZ_E_INVALID, "config: dense layers need inter %% 8 == 0");
    if ((size_t)(cfg->mla_kv_lora_rank * 6 + cfg->mla_rope_dim * 2 + cfg->mla_nope_dim + 8 + cfg->max_seq_len) * 4 + 64 > 160 * 1024)
      BZ_FAIL(BZ_E_UNSUPPORTED, "config: max_seq_len %d too long for the single-pass MLA attention kernel", cfg->max_seq_len);
  } else if (cfg->arch == BZ_ARCH_MAMBA2) {
    if (cfg->hidden <= 0 || cfg->n_layers <= 0 || cfg->vocab <= 0 || cfg->ssm_d_inner <= 0 || cfg->ssm_n_heads <= 0 || cfg->ssm_head_dim <= 0 ||
        cfg->ssm_d_state <= 0 || cfg->ssm_n_groups <= 0 || cfg->ssm_conv_kernel < 2)
      BZ_FAIL(BZ_E_INVALID, "config: non-positive mamba2 dimension");
    if (cfg->ssm_n_heads * cfg->ssm_head_dim != cfg->ssm_d_inner || cfg->ssm_n_heads % cfg->ssm_n_groups || cfg->hidden % 8 || cfg->ssm_d_inner % 8)
      BZ_FAIL(BZ_E_INVALID, "config: inconsistent mamba2 dimensions");
  } else {
  if (cfg->hidden <= 0 || cfg->n_layers <= 0 || cfg->n_heads <= 0 || cfg->n_kv_heads <= 0 || cfg->head_dim <= 0 || cfg->inter <= 0 ||
      cfg->vocab <= 0 || cfg->max_seq_len <= 0)
    BZ_FAIL(BZ_E_INVALID, "config: non-positive dimension");
  if (cfg->n_heads % cfg->n_kv_heads) BZ_FAIL(BZ_E_INVALID, "config: n_heads %% n_kv_heads != 0");
  if (cfg->hidden % 8 || cfg->head_dim % 8 || cfg->inter % 8) BZ_FAIL(BZ_E_UNSUPPORTED, "config: hidden/head_dim/inter must be multiples of 8");
  }
  if (cfg->act_dtype != BZ_F32 && cfg->act_dtype != BZ_F16 && cfg->act_dtype != BZ_BF16) BZ_FAIL(BZ_E_INVALID, "config: bad act_dtype");
  bz_model* m = new bz_model();
  m->dev = dev; m->cfg = *cfg;
  if (cfg->arch == BZ_ARCH_DEEPSEEK2) { m->cfg.n_kv_heads = 1; m->cfg.head_dim = cfg->mla_kv_lora_rank + cfg->mla_rope_dim; }   // shape of the latent cache
  bz_dev_retain(dev);
  *out = m;
  return BZ_OK;
  BZ_API_END
}

static void raw_free(RawTensor& r) {
  if (r.d0) hipFree(r.d0);
  if (r.d1) hipFree(r.d1);
  if (r.d2) hipFree(r.d2);
  if (r.d_bias) hipFree(r.d_bias);
  r.d0 = r.d1 = r.d2 = nullptr; r.d_bias = nullptr;
}

extern "C" int bz_model_free(bz_model* m) {
  BZ_API_BEGIN
  if (!m) return BZ_OK;
  hipSetDevice(m->dev->id);
  hipStreamSynchronize(m->dev->stream);
  for (auto& kv : m->raw) raw_free(kv.second);
  for (void* p : m->owned) hipFree(p);
  bz_dev_release(m->dev);
  delete m;
  return BZ_OK;
  BZ_API_END
}

static int check_add(bz_model* m, const char* name) {
  if (!m || !name) BZ_FAIL(BZ_E_INVALID, "model add: null argument");
  if (m->finalized) BZ_FAIL(BZ_E_INVALID, "model add: model already finalized");
  if (m->raw.count(name)) BZ_FAIL(BZ_E_INVALID, "model add: duplicate tensor '%s'", name);
  BZ_HIP(hipSetDevice(m->dev->id));
  return BZ_OK;
}

extern "C" int bz_model_add_dense(bz_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* host) {
  BZ_API_BEGIN
  BZ_TRY(check_add(m, name));
  if (!host || ndim < 1 || ndim > 3) BZ_FAIL(BZ_E_INVALID, "add_dense '%s': need 1-D .. 3-D host data", name);
  if (dtype != BZ_F32 && dtype != BZ_F16 && dtype != BZ_BF16) BZ_FAIL(BZ_E_INVALID, "add_dense '%s': dtype %d", name, dtype);
  RawTensor r; r.kind = 0; r.dtype = dtype; r.shape.assign(shape, shape + ndim);
  r.N = shape[0]; r.K = 1;
  for (int i = 1; i < ndim; i++) r.K *= shape[i];   // conv1d.weight [C,1,k] folds to [C,k]
  for (int i = 0; i < ndim; i++) if (shape[i] <= 0) BZ_FAIL(BZ_E_INVALID, "add_dense '%s': non-positive dimension", name);
  r.bytes = (size_t)r.N * r.K * bz_dtype_size(dtype);
  BZ_TRY(upload(m->dev, &r.d0, host, r.bytes));
  m->raw[name] = r;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_model_add_awq(bz_model* m, const char* name, int64_t N, int64_t K, const uint32_t* qweight, const float* scales,
                                const float* zeros, int gs) {
  BZ_API_BEGIN
  BZ_TRY(check_add(m, name));
  if (!qweight || !scales || !zeros) BZ_FAIL(BZ_E_INVALID, "add_awq '%s': null data", name);
  if (gs != 128) BZ_FAIL(BZ_E_UNSUPPORTED, "add_awq '%s': group_size %d (only 128 is implemented)", name, gs);
  if (N % 64 || K % 128) BZ_FAIL(BZ_E_UNSUPPORTED, "add_awq '%s': N=%lld must be a multiple of 64 and K=%lld of 128", name, (long long)N, (long long)K);
  RawTensor r; r.kind = 1; r.N = N; r.K = K; r.gs = gs;
  const size_t G = (size_t)K / gs;
  // the reference hands f32 scales/zeros that originate from f16 / 4-bit values (awq.rs:202-213): verify so that
  // storing them as f16 / u8 in HBM is exact
  for (size_t i = 0; i < G * (size_t)N; i++) {
    float s = scales[i];
    if (__half2float(__float2half(s)) != s) BZ_FAIL(BZ_E_UNSUPPORTED, "add_awq '%s': scale[%zu]=%g is not f16-representable", name, i, s);
    float z = zeros[i];
    if (!(z >= 0.f && z <= 15.f && z == floorf(z))) BZ_FAIL(BZ_E_INVALID, "add_awq '%s': zero[%zu]=%g is not an integer in [0,15]", name, i, z);
  }
  BZ_TRY(upload(m->dev, &r.d0, qweight, (size_t)K * (N / 8) * 4));
  BZ_TRY(upload(m->dev, &r.d1, scales, G * N * 4));
  BZ_TRY(upload(m->dev, &r.d2, zeros, G * N * 4));
  m->raw[name] = r;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_model_add_gptq(bz_model* m, const char* name, int64_t N, int64_t K, const uint32_t* qweight, const float* scales,
                                 const uint32_t* qzeros, const int32_t* g_idx, const float* bias, int gs) {
  BZ_API_BEGIN
  BZ_TRY(check_add(m, name));
  if (!qweight || !scales || !qzeros) BZ_FAIL(BZ_E_INVALID, "add_gptq '%s': null data", name);
  if (gs != 128) BZ_FAIL(BZ_E_UNSUPPORTED, "add_gptq '%s': group_size %d (only 128 is implemented)", name, gs);
  if (N % 64 || K % 128) BZ_FAIL(BZ_E_UNSUPPORTED, "add_gptq '%s': N must be a multiple of 64 and K of 128", name);
  RawTensor r; r.kind = 2; r.N = N; r.K = K; r.gs = gs;
  const size_t G = (size_t)K / gs;
  for (size_t i = 0; i < G * (size_t)N; i++)
    if (__half2float(__float2half(scales[i])) != scales[i]) BZ_FAIL(BZ_E_UNSUPPORTED, "add_gptq '%s': scale[%zu] is not f16-representable", name, i);
  if (g_idx) {
    r.g_idx.assign(g_idx, g_idx + K);
    std::vector<int> cnt(G, 0);
    for (int64_t k = 0; k < K; k++) {
      if (g_idx[k] < 0 || (size_t)g_idx[k] >= G) BZ_FAIL(BZ_E_INVALID, "add_gptq '%s': g_idx[%lld]=%d out of range", name, (long long)k, g_idx[k]);
      cnt[g_idx[k]]++;
    }
    for (size_t g = 0; g < G; g++) if (cnt[g] != gs) BZ_FAIL(BZ_E_UNSUPPORTED, "add_gptq '%s': group %zu has %d members (expected %d)", name, g, cnt[g], gs);
  }
  BZ_TRY(upload(m->dev, &r.d0, qweight, (size_t)(K / 8) * N * 4));
  BZ_TRY(upload(m->dev, &r.d1, scales, G * N * 4));
  BZ_TRY(upload(m->dev, &r.d2, qzeros, G * (N / 8) * 4));
  if (bias) { void* b; BZ_TRY(upload(m->dev, &b, bias, (size_t)N * 4)); r.d_bias = (float*)b; }
  m->raw[name] = r;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_model_add_gguf(bz_model* m, const char* name, int ggml_type, int64_t N, int64_t K, const void* blocks) {
  BZ_API_BEGIN
  BZ_TRY(check_add(m, name));
  if (!blocks) BZ_FAIL(BZ_E_INVALID, "add_gguf '%s': null data", name);
  if (ggml_type == BZ_GGML_F32 || ggml_type == BZ_GGML_F16 || ggml_type == BZ_GGML_BF16) {
    int64_t shape[2] = {N, K};
    int dt = ggml_type == BZ_GGML_F32 ? BZ_F32 : (ggml_type == BZ_GGML_F16 ? BZ_F16 : BZ_BF16);
    return bz_model_add_dense(m, name, dt, shape, K > 1 ? 2 : 1, blocks);
  }
  size_t rowb = 0;
  if (ggml_type == BZ_GGML_Q8_0) rowb = (size_t)K / 32 * 34;
  else if (ggml_type == BZ_GGML_Q4_K) rowb = (size_t)K / 256 * 144;
  else if (ggml_type == BZ_GGML_Q6_K) rowb = (size_t)K / 256 * 210;
  else BZ_FAIL(BZ_E_UNSUPPORTED, "add_gguf '%s': ggml type %d is not implemented (F32, F16, BF16, Q8_0, Q4_K, Q6_K are)", name, ggml_type);
  if (N % 64 || K % 256) BZ_FAIL(BZ_E_UNSUPPORTED, "add_gguf '%s': N=%lld must be a multiple of 64 and K=%lld of 256", name, (long long)N, (long long)K);
  RawTensor r; r.kind = 3; r.N = N; r.K = K; r.ggml_type = ggml_type; r.bytes = (size_t)N * rowb;
  r.shape = {N, K};
  BZ_TRY(upload(m->dev, &r.d0, blocks, r.bytes));
  m->raw[name] = r;
  return BZ_OK;
  BZ_API_END
}

// --- finalize helpers ---------------------------------------------------------------------------------------
static int choose_gw(int N, int K, int target) {
  const int G = K / 128, nst = (N + 255) / 256;
  int best = 1; double bestc = 1e30;
  for (int gw = 1; gw <= std::min(G, 16); gw++) {
    if (G % gw) continue;
    const double wgs = (double)nst * (G / gw);
    double c = fabs(log(wgs / (double)target));
    if (wgs < 256) c += 1.0;  // never fewer workgroups than CUs
    if (c < bestc) { bestc = c; best = gw; }
  }
  return best;
}

static int gemv_target_wgs() {
  const char* e = getenv("BZ_GEMV_TARGET_WGS");
  int t = e ? atoi(e) : 0;
  return t > 0 ? t : 600;
}

// Build one LinearDev (kernel layout) from several raw tensors concatenated along N (same K, same kind).
static int build_q4g(bz_model* m, const std::vector<RawTensor*>& rs, LinearDev* L) {
  const int K = (int)rs[0]->K, gs = rs[0]->gs;
  int N = 0;
  for (auto* r : rs) N += (int)r->N;
  const size_t G = (size_t)K / gs;
  const size_t wbytes = (size_t)N * K / 2, sbytes = (size_t)N * G * 2, zbytes = (size_t)N * G;
  void *w, *s, *z;
  BZ_TRY(dev_alloc(m, &w, wbytes));
  BZ_TRY(dev_alloc(m, &s, sbytes));
  BZ_TRY(dev_alloc(m, &z, zbytes));
  hipStream_t st = m->dev->stream;
  size_t n0 = 0;
  int* d_perm = nullptr; int* d_gidx = nullptr;
  for (auto* r : rs) {
    char* wo = (char*)w + n0 * K / 2;
    char* so = (char*)s + n0 * G * 2;
    char* zo = (char*)z + n0 * G;
    if (r->kind == 1) {
      BZ_TRY(bzk_repack_awq(st, (const uint32_t*)r->d0, (const float*)r->d1, (const float*)r->d2, (int)r->N, K, gs, wo, so, zo));
    } else {
      if (!r->g_idx.empty() && !d_perm) {
        // act-order: sort k by group (stable) so that every group is contiguous; x is gathered through perm in the prologue
        std::vector<int> perm(K);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return r->g_idx[a] < r->g_idx[b]; });
        bool ident = true;
        for (int k = 0; k < K; k++) if (perm[k] != k) { ident = false; break; }
        if (!ident) {
          void* p; BZ_TRY(dev_alloc(m, &p, (size_t)K * 4));
          BZ_HIP(hipMemcpy(p, perm.data(), (size_t)K * 4, hipMemcpyHostToDevice));
          d_perm = (int*)p;
          void* gi; BZ_TRY(dev_alloc(m, &gi, (size_t)K * 4));
          BZ_HIP(hipMemcpy(gi, r->g_idx.data(), (size_t)K * 4, hipMemcpyHostToDevice));
          d_gidx = (int*)gi;
        }
      }
      BZ_TRY(bzk_repack_gptq(st, (const uint32_t*)r->d0, (const float*)r->d1, (const uint32_t*)r->d2, d_perm, d_gidx, (int)r->N, K, gs, wo, so, zo));
    }
    n0 += (size_t)r->N;
  }
  // bias (GPTQ): concatenated, zero where absent
  bool any_bias = false;
  for (auto* r : rs) any_bias |= r->d_bias != nullptr;
  float* bias = nullptr;
  if (any_bias) {
    void* b; BZ_TRY(dev_alloc(m, &b, (size_t)N * 4));
    BZ_HIP(hipMemsetAsync(b, 0, (size_t)N * 4, st));
    size_t o = 0;
    for (auto* r : rs) { if (r->d_bias) BZ_HIP(hipMemcpyAsync((float*)b + o, r->d_bias, (size_t)r->N * 4, hipMemcpyDeviceToDevice, st)); o += (size_t)r->N; }
    bias = (float*)b;
  }
  BZ_HIP(hipStreamSynchronize(st));
  L->kind = LK_Q4G; L->N = N; L->K = K; L->gs = gs; L->w = w; L->scales = s; L->zeros = z; L->perm = d_perm; L->bias = bias;
  L->gw = choose_gw(N, K, gemv_target_wgs());
  L->bytes = wbytes + sbytes + zbytes;
  L->algo_bytes = wbytes + sbytes + (size_t)N * G / 2;
  return BZ_OK;
}

static int build_rows(bz_model* m, const std::vector<RawTensor*>& rs, LinearDev* L) {
  const int K = (int)rs[0]->K, dt = rs[0]->dtype;
  int N = 0;
  for (auto* r : rs) { N += (int)r->N; if (r->dtype != dt || r->K != K) BZ_FAIL(BZ_E_UNSUPPORTED, "fused dense parts differ in dtype/K"); }
  const size_t es = bz_dtype_size(dt);
  void* w;
  if (rs.size() == 1) {
    w = rs[0]->d0; rs[0]->d0 = nullptr; m->owned.push_back(w);  // take ownership, no copy
  } else {
    BZ_TRY(dev_alloc(m, &w, (size_t)N * K * es));
    size_t o = 0;
    for (auto* r : rs) { BZ_HIP(hipMemcpy((char*)w + o, r->d0, (size_t)r->N * K * es, hipMemcpyDeviceToDevice)); o += (size_t)r->N * K * es; }
  }
  L->kind = LK_ROWS; L->N = N; L->K = K; L->wdt = dt; L->w = w;
  L->bytes = (size_t)N * K * es; L->algo_bytes = L->bytes;
  return BZ_OK;
}

// dense linear whose consumer reads plain f32 (lm_head logits, Mamba2 in_proj): no split-K
static void force_direct(FusedLinear* F) {
  bool rows = true;
  for (auto& p : F->parts) rows = rows && p.kind == LK_ROWS;
  if (!rows) return;
  for (auto& p : F->parts) p.sk = 1;
  F->fix_out = false;
}

static int choose_sbw(int N, int K, int target) {
  const int SB = K / 256, nst = (N + 255) / 256;
  int best = 1; double bestc = 1e30;
  for (int w = 1; w <= std::min(SB, 8); w++) {
    if (SB % w) continue;
    const double wgs = (double)nst * (SB / w);
    double c = fabs(log(wgs / (double)target));
    if (wgs < 256) c += 1.0;
    if (c < bestc) { bestc = c; best = w; }
  }
  return best;
}

// GGUF block tensors of one ggml type, concatenated along N
static int build_gq(bz_model* m, const std::vector<RawTensor*>& rs, LinearDev* L) {
  const int K = (int)rs[0]->K, type = rs[0]->ggml_type;
  int N = 0;
  for (auto* r : rs) N += (int)r->N;
  const int kind = type == BZ_GGML_Q8_0 ? LK_Q80 : (type == BZ_GGML_Q4_K ? LK_Q4K : LK_Q6K);
  size_t wqb, whb = 0, hdb = 0, ddb;
  if (kind == LK_Q80) { wqb = (size_t)N * K; ddb = (size_t)N * (K / 32) * 2; }
  else if (kind == LK_Q4K) { wqb = (size_t)N * K / 2; hdb = (size_t)N * (K / 256) * 16; ddb = 0; }
  else { wqb = (size_t)N * K / 2; whb = (size_t)N * K / 4; hdb = (size_t)N * (K / 256) * 16; ddb = (size_t)N * (K / 256) * 2; }
  void *wq = nullptr, *wh = nullptr, *hd = nullptr, *dd = nullptr;
  BZ_TRY(dev_alloc(m, &wq, wqb));
  if (whb) BZ_TRY(dev_alloc(m, &wh, whb));
  if (hdb) BZ_TRY(dev_alloc(m, &hd, hdb));
  if (ddb) BZ_TRY(dev_alloc(m, &dd, ddb));
  size_t n0 = 0;
  for (auto* r : rs) {
    // per-tile layouts: a part starting at column n0 starts at tile n0/64 of every array
    const size_t t0 = n0 / 64;
    char* wqo = (char*)wq + t0 * (wqb / (N / 64));
    char* who = wh ? (char*)wh + t0 * (whb / (N / 64)) : nullptr;
    char* hdo = hd ? (char*)hd + t0 * (hdb / (N / 64)) : nullptr;
    char* ddo = dd ? (char*)dd + t0 * (ddb / (N / 64)) : nullptr;
    BZ_TRY(bzk_repack_gq(m->dev->stream, kind, r->d0, (int)r->N, K, wqo, who, hdo, ddo));
    n0 += (size_t)r->N;
  }
  BZ_HIP(hipStreamSynchronize(m->dev->stream));
  L->kind = kind; L->N = N; L->K = K; L->gs = kind == LK_Q80 ? 32 : 256; L->w = wq; L->zeros = wh; L->hdr = hd; L->scales = dd;
  L->gw = choose_sbw(N, K, gemv_target_wgs());
  L->bytes = wqb + whb + hdb + ddb;
  L->algo_bytes = L->bytes;     // the repack keeps the ggml bytes per weight (34/32, 144/256, 210/256)
  return BZ_OK;
}

static bool same_perm(const RawTensor* a, const RawTensor* b) { return a->g_idx == b->g_idx; }

// names: HF tensor names (without ".weight"); fuses along N when layouts allow, otherwise keeps separate parts
static int build_fused(bz_model* m, const std::vector<std::string>& names, FusedLinear* F) {
  std::vector<RawTensor*> rs;
  for (auto& nm : names) {
    auto it = m->raw.find(nm + ".weight");
    if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: tensor '%s.weight' was not added", nm.c_str());
    // optional dense bias "<name>.bias" for dense/awq layers is not part of the Llama family; GPTQ carries its own
    rs.push_back(&it->second);
  }
  F->K = (int)rs[0]->K; F->N = 0;
  for (auto* r : rs) { if (r->K != F->K) BZ_FAIL(BZ_E_INVALID, "finalize: fused parts disagree on K"); F->N += (int)r->N; }
  // consecutive parts of one storage kind share a launch (Q4_K_M gives q, k in Q4_K and v in Q6_K: [q, k] + [v])
  auto compatible = [&](const RawTensor* a, const RawTensor* b) {
    if (a->kind != b->kind) return false;
    if (a->kind == 2 && !same_perm(a, b)) return false;
    if (a->kind == 0 && a->dtype != b->dtype) return false;
    if (a->kind == 3 && a->ggml_type != b->ggml_type) return false;
    return true;
  };
  std::vector<std::vector<RawTensor*>> groups;
  for (auto* r : rs) {
    if (!groups.empty() && compatible(groups.back()[0], r)) groups.back().push_back(r);
    else groups.push_back({r});
  }
  int noff = 0; size_t ri = 0;
  for (auto& g : groups) {
    LinearDev L;
    if (g[0]->kind == 1 || g[0]->kind == 2) BZ_TRY(build_q4g(m, g, &L));
    else if (g[0]->kind == 0) { if (g[0]->shape.size() != 2) BZ_FAIL(BZ_E_INVALID, "finalize: linear weight must be 2-D"); BZ_TRY(build_rows(m, g, &L)); }
    else if (g[0]->kind == 3) BZ_TRY(build_gq(m, g, &L));
    else BZ_FAIL(BZ_E_UNSUPPORTED, "finalize: tensor kind %d", g[0]->kind);
    F->parts.push_back(L); F->n_off.push_back(noff);
    // per-name views for the op-level API
    int sub = 0;
    for (auto* r : g) {
      LinearDev V = L; V.owned = false; V.N = (int)r->N;
      if (L.kind == LK_Q4G) {
        const size_t G = (size_t)L.K / 128;
        V.w = (char*)L.w + (size_t)sub * L.K / 2; V.scales = (char*)L.scales + (size_t)sub * G * 2; V.zeros = (char*)L.zeros + (size_t)sub * G;
        V.bias = L.bias ? L.bias + sub : nullptr;
        V.gw = choose_gw(V.N, V.K, gemv_target_wgs());
      } else if (L.kind == LK_ROWS) {
        V.sk = 1;
        V.w = (char*)L.w + (size_t)sub * L.K * bz_dtype_size(L.wdt);
      } else {
        const size_t t0 = (size_t)sub / 64, K = (size_t)L.K;
        if (L.kind == LK_Q80) { V.w = (char*)L.w + t0 * 64 * K; V.scales = (char*)L.scales + t0 * (K / 32) * 64 * 2; }
        else if (L.kind == LK_Q4K) { V.w = (char*)L.w + t0 * 32 * K; V.hdr = (char*)L.hdr + t0 * (K / 256) * 64 * 16; }
        else { V.w = (char*)L.w + t0 * 32 * K; V.zeros = (char*)L.zeros + t0 * 16 * K; V.hdr = (char*)L.hdr + t0 * (K / 256) * 64 * 16;
               V.scales = (char*)L.scales + t0 * (K / 256) * 64 * 2; }
        V.gw = choose_sbw(V.N, V.K, gemv_target_wgs());
      }
      m->named[names[ri] + ".weight"] = V;
      sub += (int)r->N; ri++;
    }
    noff += L.N;
  }
  static const bool no_sk = getenv("BZ_NO_ROWS_SPLITK") != nullptr;
  for (auto& p : F->parts) if (p.kind == LK_ROWS && !no_sk) p.sk = bzk_rows_choose_sk(p.N, p.K);
  auto is_fix = [](const LinearDev& p) { return p.kind != LK_ROWS || p.sk > 1; };
  F->fix_out = is_fix(F->parts[0]);
  for (auto& p : F->parts) if (is_fix(p) != F->fix_out) BZ_FAIL(BZ_E_UNSUPPORTED, "finalize: mixed fixed-point/direct parts in one fused linear");
  for (auto* r : rs) { raw_free(*r); r->consumed = true; }
  return BZ_OK;
}

static int take_vector_f32(bz_model* m, const std::string& name, int n, float** out, bool round_to_act = true) {
  auto it = m->raw.find(name);
  if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: tensor '%s' was not added", name.c_str());
  RawTensor& r = it->second;
  if ((int64_t)r.N * r.K != n) BZ_FAIL(BZ_E_INVALID, "finalize: '%s' has %lld elements, expected %d", name.c_str(), (long long)(r.N * r.K), n);
  // norms are kept as f32 holding values rounded to the activation dtype (awq.rs:93-103 casts BF16 -> F16)
  std::vector<char> host(r.bytes);
  BZ_HIP(hipMemcpy(host.data(), r.d0, r.bytes, hipMemcpyDeviceToHost));
  std::vector<float> f(n);
  for (int i = 0; i < n; i++) {
    float v;
    if (r.dtype == BZ_F32) v = ((float*)host.data())[i];
    else if (r.dtype == BZ_F16) v = __half2float(((__half*)host.data())[i]);
    else { uint32_t u = (uint32_t)((uint16_t*)host.data())[i] << 16; memcpy(&v, &u, 4); }
    if (!round_to_act) {}
    else if (m->cfg.act_dtype == BZ_F16) v = __half2float(__float2half(v));
    else if (m->cfg.act_dtype == BZ_BF16) { uint32_t u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&v, &u, 4); }
    f[i] = v;
  }
  void* d; BZ_TRY(dev_alloc(m, &d, (size_t)n * 4));
  BZ_HIP(hipMemcpy(d, f.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  *out = (float*)d;
  raw_free(r); r.consumed = true;
  return BZ_OK;
}

// RoPE tables: same formula as oracle/orc_ops.c (restated, not shared): HF inv_freq with linear / llama3 scaling
// (/root/reference/src/loader/safetensors/config.rs:83-95), angle = (float)pos * (float)inv_freq, cos/sin via double libm.
static void rope_tables_host(const bz_model_config& c, std::vector<float>& cs, std::vector<float>& sn) {
  const int half = c.head_dim / 2;
  cs.resize((size_t)c.max_seq_len * half); sn.resize((size_t)c.max_seq_len * half);
  const double PI2 = 6.283185307179586476925286766559;
  // YaRN (HF _compute_yarn_parameters): NTK-by-parts blend of the interpolated and the original frequencies between the dims that make beta_fast and
  // beta_slow rotations over the original context; cos / sin scaled by the attention factor
  double ylow = 0.0, yhigh = 0.0; float af = 1.0f;
  if (c.rope_scaling == BZ_ROPE_YARN) {
    const double bf = c.rope_beta_fast > 0.f ? c.rope_beta_fast : 32.0, bsl = c.rope_beta_slow > 0.f ? c.rope_beta_slow : 1.0;
    const double dim = c.head_dim, base = c.rope_theta, omax = c.rope_original_max_pos;
    auto corr = [&](double rot) { return dim * log(omax / (rot * PI2)) / (2.0 * log(base)); };
    ylow = std::max(floor(corr(bf)), 0.0); yhigh = std::min(ceil(corr(bsl)), dim - 1.0);
    if (ylow == yhigh) yhigh += 0.001;
    af = c.rope_attn_factor > 0.f ? c.rope_attn_factor : (c.rope_factor <= 1.f ? 1.0f : (float)(0.1 * log((double)c.rope_factor) + 1.0));
  }
  for (int i = 0; i < half; i++) {
    double inv = 1.0 / pow((double)c.rope_theta, (double)(2 * i) / (double)c.head_dim);
    if (c.rope_scaling == BZ_ROPE_YARN) {
      const double ramp = std::min(std::max(((double)i - ylow) / (yhigh - ylow), 0.0), 1.0), ext = 1.0 - ramp;
      inv = (inv / (double)c.rope_factor) * (1.0 - ext) + inv * ext;
    }
    if (c.rope_scaling == BZ_ROPE_LINEAR) inv /= (double)c.rope_factor;
    else if (c.rope_scaling == BZ_ROPE_LLAMA3) {
      double low_wl = (double)c.rope_original_max_pos / (double)c.rope_low_freq_factor;
      double high_wl = (double)c.rope_original_max_pos / (double)c.rope_high_freq_factor;
      double wl = PI2 / inv;
      if (wl > low_wl) inv = inv / (double)c.rope_factor;
      else if (wl >= high_wl) {
        double smooth = ((double)c.rope_original_max_pos / wl - (double)c.rope_low_freq_factor) /
                        ((double)c.rope_high_freq_factor - (double)c.rope_low_freq_factor);
        inv = (1.0 - smooth) * inv / (double)c.rope_factor + smooth * inv;
      }
    }
    const float invf = (float)inv;
    for (int p = 0; p < c.max_seq_len; p++) {
      float ang = (float)p * invf;
      cs[(size_t)p * half + i] = (float)cos((double)ang) * af;
      sn[(size_t)p * half + i] = (float)sin((double)ang) * af;
    }
  }
}
static float mla_softmax_scale(const bz_model_config& c) {
  const float ms = c.mla_softmax_mscale > 0.f ? c.mla_softmax_mscale : 1.0f;
  return 1.0f / sqrtf((float)(c.mla_nope_dim + c.mla_rope_dim)) * ms * ms;     // HF DeepseekV2Attention: softmax_scale * mscale * mscale
}

static int finalize_mamba2(bz_model* m);
static int finalize_dsv2(bz_model* m);

extern "C" int bz_model_finalize(bz_model* m) {
  BZ_API_BEGIN
  if (!m) BZ_FAIL(BZ_E_INVALID, "null model");
  if (m->finalized) BZ_FAIL(BZ_E_INVALID, "model already finalized");
  BZ_HIP(hipSetDevice(m->dev->id));
  if (m->cfg.arch == BZ_ARCH_MAMBA2) return finalize_mamba2(m);
  if (m->cfg.arch == BZ_ARCH_DEEPSEEK2) return finalize_dsv2(m);
  const bz_model_config& c = m->cfg;
  const int H = c.hidden, nq = c.n_heads, nkv = c.n_kv_heads, hd = c.head_dim, I = c.inter, V = c.vocab;
  char nm[256];
  m->layers.resize(c.n_layers);
  for (int l = 0; l < c.n_layers; l++) {
    LayerDev& Ld = m->layers[l];
    snprintf(nm, sizeof nm, "model.layers.%d.", l);
    std::string p = nm;
    BZ_TRY(take_vector_f32(m, p + "input_layernorm.weight", H, &Ld.attn_norm));
    BZ_TRY(take_vector_f32(m, p + "post_attention_layernorm.weight", H, &Ld.ffn_norm));
    BZ_TRY(build_fused(m, {p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"}, &Ld.qkv));
    BZ_TRY(build_fused(m, {p + "self_attn.o_proj"}, &Ld.o));
    BZ_TRY(build_fused(m, {p + "mlp.gate_proj", p + "mlp.up_proj"}, &Ld.gateup));
    BZ_TRY(build_fused(m, {p + "mlp.down_proj"}, &Ld.down));
    if (Ld.qkv.N != (nq + 2 * nkv) * hd || Ld.qkv.K != H) BZ_FAIL(BZ_E_INVALID, "layer %d: q/k/v shapes do not match the config", l);
    if (Ld.o.N != H || Ld.o.K != nq * hd) BZ_FAIL(BZ_E_INVALID, "layer %d: o_proj shape does not match the config", l);
    if (Ld.gateup.N != 2 * I || Ld.gateup.K != H) BZ_FAIL(BZ_E_INVALID, "layer %d: gate/up shapes do not match the config", l);
    if (Ld.down.N != H || Ld.down.K != I) BZ_FAIL(BZ_E_INVALID, "layer %d: down_proj shape does not match the config", l);
    // dense 16-bit MLP of the Llama-3.2-1B class: a slab-major copy of down_proj for the fused MLP launch (the row-major one stays for prefill / fallbacks)
    if (Ld.gateup.parts.size() == 1 && Ld.down.parts.size() == 1 && Ld.down.parts[0].kind == LK_ROWS) {
      char dummy = 0;
      if (bzk_mlp_dense_fusable(Ld.gateup.parts[0], Ld.down.parts[0], &dummy, H, I, c.act_dtype)) {
        BZ_TRY(dev_alloc(m, &Ld.down_slabs, (size_t)H * I * 2));
        BZ_TRY(bzk_repack_down_slabs(m->dev->stream, Ld.down.parts[0].w, H, I, Ld.down_slabs));
        m->resident += (size_t)H * I * 2;
      }
    }
  }
  BZ_HIP(hipStreamSynchronize(m->dev->stream));
  BZ_TRY(take_vector_f32(m, "model.norm.weight", H, &m->final_norm));
  // embeddings stay in their storage dtype (rows are gathered); tied lm_head reads the same buffer
  {
    auto it = m->raw.find("model.embed_tokens.weight");
    if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: 'model.embed_tokens.weight' was not added");
    RawTensor& r = it->second;
    if (r.kind != 0 || r.N != V || r.K != H) BZ_FAIL(BZ_E_INVALID, "finalize: embed_tokens must be dense [vocab, hidden]");
    m->embed = r.d0; m->embed_dt = r.dtype; r.d0 = nullptr; m->owned.push_back(m->embed); r.consumed = true;
    m->resident += r.bytes;
    if (c.tie_embeddings || !m->raw.count("lm_head.weight")) {
      LinearDev L; L.kind = LK_ROWS; L.N = V; L.K = H; L.wdt = m->embed_dt; L.w = m->embed; L.owned = false;
      L.bytes = 0; L.algo_bytes = r.bytes;
      m->lm_head.parts.push_back(L); m->lm_head.n_off.push_back(0); m->lm_head.N = V; m->lm_head.K = H; m->lm_head.fix_out = false;
      m->named["lm_head.weight"] = L;
    } else {
      BZ_TRY(build_fused(m, {"lm_head"}, &m->lm_head));
      force_direct(&m->lm_head);
      if (m->lm_head.N != V || m->lm_head.K != H) BZ_FAIL(BZ_E_INVALID, "finalize: lm_head shape does not match the config");
    }
  }
  if (m->lm_head.parts.size() != 1) BZ_FAIL(BZ_E_UNSUPPORTED, "finalize: lm_head must be a single tensor");
  for (auto& kv : m->raw) if (!kv.second.consumed) BZ_FAIL(BZ_E_INVALID, "finalize: tensor '%s' is not used by this architecture", kv.first.c_str());
  m->raw.clear();

  // rope caches
  std::vector<float> cs, sn;
  rope_tables_host(c, cs, sn);
  void* p;
  BZ_TRY(dev_alloc(m, &p, cs.size() * 4)); m->cos_t = (float*)p; BZ_HIP(hipMemcpy(p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  BZ_TRY(dev_alloc(m, &p, sn.size() * 4)); m->sin_t = (float*)p; BZ_HIP(hipMemcpy(p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
  BZ_TRY(dev_alloc(m, &p, 256 * 4)); m->rope_cur = (float*)p; BZ_HIP(hipMemset(p, 0, 256 * 4));
  BZ_TRY(dev_alloc(m, &p, bzk_attn_split_ws_bytes(c.n_heads))); m->att_ws = (float*)p;

  // workspace
  m->ring_n = std::max(std::max((nq + 2 * nkv) * hd, 2 * I), std::max(H, nq * hd));
  if (m->lm_head.fix_out) m->ring_n = std::max(m->ring_n, V);
  for (int i = 0; i < 3; i++) {
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 8)); m->ring[i] = (long long*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 8));
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 4)); m->dring[i] = (float*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 4));
  }
  for (int i = 0; i < 2; i++) { BZ_TRY(dev_alloc(m, &p, (size_t)H * 4)); m->hbuf[i] = (float*)p; }
  BZ_TRY(dev_alloc(m, &p, (size_t)nq * hd * 4)); m->attn_out = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)V * 4)); m->logits = (float*)p;
  m->nparts = m->lm_head.fix_out ? 64 : bzk_gemv_rows_blocks(m->lm_head.parts[0]);
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pval = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pidx = (int*)p;
  BZ_TRY(dev_alloc(m, &p, 4096)); m->scratch = (float*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->tok_tmp = (long long*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->pos_tmp = (int*)p;

  // the persistent decode launch (bz_persist.hip): every layer int4 without act-order / bias, the Llama-3-8B head geometry, f16 activations
  if (bzk_persist_shape_ok(H, I, nq, nkv, hd, c.act_dtype, BZ_F16) && !c.rope_interleaved) {
    bool ok = true;
    for (auto& Ld : m->layers)
      for (const FusedLinear* F : {&Ld.qkv, &Ld.o, &Ld.gateup, &Ld.down})
        ok = ok && F->parts.size() == 1 && F->parts[0].kind == LK_Q4G && !F->parts[0].perm && !F->parts[0].bias && F->fix_out;
    if (ok) {
      const size_t eb = bzk_persist_layer_bytes();
      std::vector<char> tab(eb * c.n_layers);
      for (int l = 0; l < c.n_layers && ok; l++) {
        const LayerDev& Ld = m->layers[l];
        ok = bzk_persist_fill_layer(tab.data() + eb * l, Ld.qkv.parts[0], Ld.o.parts[0], Ld.gateup.parts[0], Ld.down.parts[0], Ld.attn_norm, Ld.ffn_norm) == BZ_OK;
      }
      if (ok) {
        BZ_TRY(dev_alloc(m, &p, tab.size()));
        BZ_HIP(hipMemcpy(p, tab.data(), tab.size(), hipMemcpyHostToDevice));
        m->persist_tab = p;
      }
    }
  }

  // accounting
  size_t act_b = bz_dtype_size(c.act_dtype);
  m->per_token = (size_t)(2 * c.n_layers + 1) * H * act_b + (size_t)H * bz_dtype_size(m->embed_dt);
  for (auto& Ld : m->layers)
    for (FusedLinear* F : {&Ld.qkv, &Ld.o, &Ld.gateup, &Ld.down})
      for (auto& L : F->parts) { m->resident += L.bytes; m->per_token += L.algo_bytes; }
  for (auto& L : m->lm_head.parts) { m->resident += L.bytes; m->per_token += L.algo_bytes; }
  BZ_HIP(hipDeviceSynchronize());   // null-stream memsets above vs. the non-blocking compute stream
  m->finalized = true;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_model_get_config(const bz_model* m, bz_model_config* out) {
  BZ_API_BEGIN
  if (!m || !out) BZ_FAIL(BZ_E_INVALID, "null argument");
  *out = m->cfg;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_model_weight_bytes(const bz_model* m, size_t* resident, size_t* per_token) {
  BZ_API_BEGIN
  if (!m || !m->finalized) BZ_FAIL(BZ_E_INVALID, "model not finalized");
  if (resident) *resident = m->resident;
  if (per_token) *per_token = m->per_token;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_rope_caches(bz_model* m, float* c, float* s) {
  BZ_API_BEGIN
  if (!m || !m->finalized) BZ_FAIL(BZ_E_INVALID, "model not finalized");
  size_t n = (size_t)m->cfg.max_seq_len * (m->cfg.head_dim / 2) * 4;
  if (c) BZ_HIP(hipMemcpy(c, m->cos_t, n, hipMemcpyDeviceToHost));
  if (s) BZ_HIP(hipMemcpy(s, m->sin_t, n, hipMemcpyDeviceToHost));
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// DeepSeek-V2 family (BZ_ARCH_DEEPSEEK2): HF tensor names
// ---------------------------------------------------------------------------------------------------------
// take a dense 2-D tensor as stored (device pointer ownership moves to the model)
static int take_dense(bz_model* m, const std::string& name, int64_t N, int64_t K, void** out, int* dt, size_t* bytes) {
  auto it = m->raw.find(name);
  if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: tensor '%s' was not added", name.c_str());
  RawTensor& r = it->second;
  if (r.kind != 0 || r.N != N || r.K != K) BZ_FAIL(BZ_E_INVALID, "finalize: '%s' must be dense [%lld, %lld]", name.c_str(), (long long)N, (long long)K);
  *out = r.d0; *dt = r.dtype; if (bytes) *bytes = r.bytes;
  r.d0 = nullptr; m->owned.push_back(*out); r.consumed = true;
  m->resident += r.bytes;
  return BZ_OK;
}
// copy a dense 2-D tensor (or a column block of it) into a stacked buffer
static int stack_dense(bz_model* m, const std::string& name, int64_t N, int64_t K, int dt, void* dst, int64_t col0, int64_t ncols) {
  auto it = m->raw.find(name);
  if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: tensor '%s' was not added", name.c_str());
  RawTensor& r = it->second;
  if (r.kind != 0 || r.N != N || r.K != K || r.dtype != dt) BZ_FAIL(BZ_E_INVALID, "finalize: '%s' must be dense [%lld, %lld] of the experts' dtype", name.c_str(), (long long)N, (long long)K);
  const size_t es = bz_dtype_size(dt);
  BZ_HIP(hipMemcpy2D(dst, (size_t)ncols * es, (const char*)r.d0 + (size_t)col0 * es, (size_t)K * es, (size_t)ncols * es, (size_t)N, hipMemcpyDeviceToDevice));
  return BZ_OK;
}

static int finalize_dsv2(bz_model* m) {
  const bz_model_config& c = m->cfg;
  const int H = c.hidden, NH = c.n_heads, R = c.mla_kv_lora_rank, DN = c.mla_nope_dim, DR = c.mla_rope_dim, DV = c.mla_v_dim, V = c.vocab;
  const int E = c.moe_n_experts, TK = c.moe_top_k, NS = c.moe_n_shared, MI = c.moe_inter;
  const size_t act_b = bz_dtype_size(c.act_dtype);
  char nm[256];
  m->dlayers.resize(c.n_layers);
  size_t per_token = 0;
  int ring_n = std::max(H, NH * DV);
  for (int l = 0; l < c.n_layers; l++) {
    DsLayerDev& L = m->dlayers[l];
    snprintf(nm, sizeof nm, "model.layers.%d.", l);
    std::string p = nm;
    BZ_TRY(take_vector_f32(m, p + "input_layernorm.weight", H, &L.attn_norm));
    BZ_TRY(take_vector_f32(m, p + "post_attention_layernorm.weight", H, &L.ffn_norm));
    BZ_TRY(take_vector_f32(m, p + "self_attn.kv_a_layernorm.weight", R, &L.kv_norm));
    const int QL = c.mla_q_lora_rank;
    if (QL > 0) {   // gguf.rs:188-196 (DeepSeek-V2 full): q = q_b_proj(q_a_layernorm(q_a_proj(x)))
      BZ_TRY(build_fused(m, {p + "self_attn.q_a_proj", p + "self_attn.kv_a_proj_with_mqa"}, &L.qkva));
      BZ_TRY(take_vector_f32(m, p + "self_attn.q_a_layernorm.weight", QL, &L.q_norm));
      BZ_TRY(build_fused(m, {p + "self_attn.q_b_proj"}, &L.q_b));
      if (L.q_b.parts.size() != 1 || L.q_b.parts[0].kind != LK_ROWS || L.q_b.N != NH * (DN + DR) || L.q_b.K != QL) BZ_FAIL(BZ_E_INVALID, "layer %d: q_b_proj must be dense [n_heads (nope + rope), q_lora_rank]", l);
      for (auto& P : L.q_b.parts) { m->resident += P.bytes; per_token += P.algo_bytes; }
      per_token += (size_t)QL * act_b;
    } else {
      BZ_TRY(build_fused(m, {p + "self_attn.q_proj", p + "self_attn.kv_a_proj_with_mqa"}, &L.qkva));
    }
    if (L.qkva.parts.size() != 1 || L.qkva.parts[0].kind != LK_ROWS || L.qkva.N != (QL > 0 ? QL : NH * (DN + DR)) + R + DR || L.qkva.K != H)
      BZ_FAIL(BZ_E_INVALID, "layer %d: q_proj (q_a_proj) / kv_a_proj_with_mqa must be dense tensors matching the config", l);
    if (QL > 0) force_direct(&L.qkva);   // q_b_proj's norm prologue and the attention kernel's `kva` read q_a / the latent as plain f32
    size_t kvb_bytes = 0;
    BZ_TRY(take_dense(m, p + "self_attn.kv_b_proj.weight", (int64_t)NH * (DN + DV), R, &L.kv_b, &L.kv_b_dt, &kvb_bytes));
    BZ_TRY(build_fused(m, {p + "self_attn.o_proj"}, &L.o));
    if (L.o.N != H || L.o.K != NH * DV || L.o.parts[0].kind != LK_ROWS) BZ_FAIL(BZ_E_INVALID, "layer %d: o_proj must be dense [hidden, n_heads v_dim]", l);
    per_token += kvb_bytes + (size_t)(2 * H + R) * act_b;
    for (FusedLinear* F : {&L.qkva, &L.o}) for (auto& P : F->parts) { m->resident += P.bytes; per_token += P.algo_bytes; }
    L.is_moe = E > 0 && l >= c.moe_first_dense;
    if (!L.is_moe) {
      BZ_TRY(build_fused(m, {p + "mlp.gate_proj", p + "mlp.up_proj"}, &L.gateup));
      BZ_TRY(build_fused(m, {p + "mlp.down_proj"}, &L.down));
      if (L.gateup.N != 2 * c.inter || L.gateup.K != H || L.down.N != H || L.down.K != c.inter) BZ_FAIL(BZ_E_INVALID, "layer %d: dense MLP shapes do not match the config", l);
      for (FusedLinear* F : {&L.gateup, &L.down}) for (auto& P : F->parts) { m->resident += P.bytes; per_token += P.algo_bytes; }
      ring_n = std::max(ring_n, 2 * c.inter);
    } else {
      size_t rb = 0;
      BZ_TRY(take_dense(m, p + "mlp.gate.weight", E, H, &L.router, &L.router_dt, &rb));
      ring_n = std::max(ring_n, E);   // the router logits pass through the ring (fixed point)
      auto it = m->raw.find(p + "mlp.experts.0.gate_proj.weight");
      if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: layer %d has no expert tensors", l);
      L.e_dt = it->second.dtype;
      const size_t es = bz_dtype_size(L.e_dt);
      const size_t gu_sz = (size_t)2 * MI * H * es, dn_sz = (size_t)H * MI * es;
      BZ_TRY(dev_alloc(m, &L.e_gu, (size_t)(E + NS) * gu_sz));
      BZ_TRY(dev_alloc(m, &L.e_dn, (size_t)(E + NS) * dn_sz));
      for (int e = 0; e < E; e++) {
        snprintf(nm, sizeof nm, "%smlp.experts.%d.", p.c_str(), e);
        std::string q = nm;
        BZ_TRY(stack_dense(m, q + "gate_proj.weight", MI, H, L.e_dt, (char*)L.e_gu + (size_t)e * gu_sz, 0, H));
        BZ_TRY(stack_dense(m, q + "up_proj.weight", MI, H, L.e_dt, (char*)L.e_gu + (size_t)e * gu_sz + gu_sz / 2, 0, H));
        BZ_TRY(stack_dense(m, q + "down_proj.weight", H, MI, L.e_dt, (char*)L.e_dn + (size_t)e * dn_sz, 0, MI));
        for (const char* t : {"gate_proj.weight", "up_proj.weight", "down_proj.weight"}) { RawTensor& r = m->raw[q + t]; raw_free(r); r.consumed = true; }
      }
      if (NS > 0) {
        // the shared experts are one MLP of width NS * moe_inter: split into NS expert-shaped slots (exact: the down projection is a sum over its input)
        std::string q = p + "mlp.shared_experts.";
        auto itg = m->raw.find(q + "gate_proj.weight"), itu = m->raw.find(q + "up_proj.weight");
        if (itg == m->raw.end() || itu == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: layer %d shared experts missing", l);
        if (itg->second.kind != 0 || itg->second.N != (int64_t)NS * MI || itg->second.K != H || itg->second.dtype != L.e_dt || itu->second.kind != 0 ||
            itu->second.N != (int64_t)NS * MI || itu->second.K != H || itu->second.dtype != L.e_dt)
          BZ_FAIL(BZ_E_INVALID, "finalize: layer %d shared expert shapes do not match the config", l);
        for (int j = 0; j < NS; j++) {
          const size_t half_rows = (size_t)MI * H * es;
          BZ_HIP(hipMemcpy((char*)L.e_gu + (size_t)(E + j) * gu_sz, (const char*)itg->second.d0 + (size_t)j * half_rows, half_rows, hipMemcpyDeviceToDevice));
          BZ_HIP(hipMemcpy((char*)L.e_gu + (size_t)(E + j) * gu_sz + gu_sz / 2, (const char*)itu->second.d0 + (size_t)j * half_rows, half_rows, hipMemcpyDeviceToDevice));
          BZ_TRY(stack_dense(m, q + "down_proj.weight", H, (int64_t)NS * MI, L.e_dt, (char*)L.e_dn + (size_t)(E + j) * dn_sz, (int64_t)j * MI, MI));
        }
        for (const char* t : {"gate_proj.weight", "up_proj.weight", "down_proj.weight"}) { RawTensor& r = m->raw[q + t]; raw_free(r); r.consumed = true; }
      }
      m->resident += (size_t)(E + NS) * (gu_sz + dn_sz);
      per_token += rb + (size_t)(TK + NS) * (gu_sz + dn_sz);
    }
  }
  BZ_TRY(take_vector_f32(m, "model.norm.weight", H, &m->final_norm));
  {
    auto it = m->raw.find("model.embed_tokens.weight");
    if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: 'model.embed_tokens.weight' was not added");
    RawTensor& r = it->second;
    if (r.kind != 0 || r.N != V || r.K != H) BZ_FAIL(BZ_E_INVALID, "finalize: embed_tokens must be dense [vocab, hidden]");
    m->embed = r.d0; m->embed_dt = r.dtype; r.d0 = nullptr; m->owned.push_back(m->embed); r.consumed = true;
    m->resident += r.bytes;
    if (c.tie_embeddings || !m->raw.count("lm_head.weight")) {
      LinearDev L; L.kind = LK_ROWS; L.N = V; L.K = H; L.wdt = m->embed_dt; L.w = m->embed; L.owned = false; L.bytes = 0; L.algo_bytes = r.bytes;
      m->lm_head.parts.push_back(L); m->lm_head.n_off.push_back(0); m->lm_head.N = V; m->lm_head.K = H; m->lm_head.fix_out = false;
      m->named["lm_head.weight"] = L;
    } else {
      BZ_TRY(build_fused(m, {"lm_head"}, &m->lm_head));
      force_direct(&m->lm_head);
      if (m->lm_head.N != V || m->lm_head.K != H || m->lm_head.fix_out) BZ_FAIL(BZ_E_INVALID, "finalize: lm_head must be dense [vocab, hidden]");
    }
  }
  for (auto& kv : m->raw) if (!kv.second.consumed) BZ_FAIL(BZ_E_INVALID, "finalize: tensor '%s' is not used by this architecture", kv.first.c_str());
  m->raw.clear();
  // decoupled RoPE tables over the rope dims only
  bz_model_config rc = c; rc.head_dim = DR;
  std::vector<float> cs, sn;
  rope_tables_host(rc, cs, sn);
  void* p;
  BZ_TRY(dev_alloc(m, &p, cs.size() * 4)); m->cos_t = (float*)p; BZ_HIP(hipMemcpy(p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  BZ_TRY(dev_alloc(m, &p, sn.size() * 4)); m->sin_t = (float*)p; BZ_HIP(hipMemcpy(p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
  m->ring_n = std::max(ring_n, NH * (DN + DR) + R + DR + c.mla_q_lora_rank);
  for (int i = 0; i < 3; i++) {
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 8)); m->ring[i] = (long long*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 8));
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 4)); m->dring[i] = (float*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 4));
  }
  for (int i = 0; i < 2; i++) { BZ_TRY(dev_alloc(m, &p, (size_t)H * 4)); m->hbuf[i] = (float*)p; }
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * DV * 4)); m->attn_out = (float*)p;
  m->mla_nsplit = bzk_mla_nsplit(NH);
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * m->mla_nsplit * (R + 2) * 4)); m->mla_ws = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * m->mla_nsplit * (R + 1) * 8)); m->mla_wsd = (double*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)(2 * NH + 2) * 4)); m->mla_sync = (unsigned*)p; BZ_HIP(hipMemset(p, 0, (size_t)(2 * NH + 2) * 4));
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * m->mla_nsplit * ((c.max_seq_len + m->mla_nsplit - 1) / m->mla_nsplit + 1) * 4)); m->mla_scw = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * m->mla_nsplit * 4)); m->mla_mxw = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)V * 4)); m->logits = (float*)p;
  if (E > 0) {
    BZ_TRY(dev_alloc(m, &p, (size_t)H * 4)); m->moe_xn = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)(TK + NS) * 2 * MI * 4)); m->moe_gu = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)(TK + NS) * 2 * MI * 8)); m->moe_gu_acc = (long long*)p; BZ_HIP(hipMemset(p, 0, (size_t)(TK + NS) * 2 * MI * 8));
    BZ_TRY(dev_alloc(m, &p, (size_t)H * 4)); m->moe_out = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)(TK + 1) * H * 8)); m->moe_acc = (long long*)p; BZ_HIP(hipMemset(p, 0, (size_t)(TK + 1) * H * 8));
    BZ_TRY(dev_alloc(m, &p, (size_t)(TK + NS + 4) * 4)); m->moe_sel = (int*)p; BZ_HIP(hipMemset(p, 0, (size_t)(TK + NS + 4) * 4));
    BZ_TRY(dev_alloc(m, &p, (size_t)(TK + NS + 4) * 4)); m->moe_w = (float*)p; BZ_HIP(hipMemset(p, 0, (size_t)(TK + NS + 4) * 4));
    BZ_TRY(dev_alloc(m, &p, (size_t)(E + 4) * 4)); m->moe_lg = (float*)p;
    BZ_TRY(dev_alloc(m, &p, 64)); m->moe_cnt = (unsigned*)p; BZ_HIP(hipMemset(p, 0, 64));
  }
  m->nparts = bzk_gemv_rows_blocks(m->lm_head.parts[0]);
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pval = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pidx = (int*)p;
  BZ_TRY(dev_alloc(m, &p, 4096)); m->scratch = (float*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->tok_tmp = (long long*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->pos_tmp = (int*)p;
  for (auto& L : m->lm_head.parts) { m->resident += L.bytes; per_token += L.algo_bytes; }
  m->per_token = per_token + (size_t)H * bz_dtype_size(m->embed_dt) + (size_t)H * act_b;
  BZ_HIP(hipDeviceSynchronize());
  m->finalized = true;
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Mamba2 (BZ_ARCH_MAMBA2): HF Mamba2 tensor names
// ---------------------------------------------------------------------------------------------------------
static int finalize_mamba2(bz_model* m) {
  const bz_model_config& c = m->cfg;
  const int D = c.hidden, DI = c.ssm_d_inner, NH = c.ssm_n_heads, NS = c.ssm_d_state, G = c.ssm_n_groups, KC = c.ssm_conv_kernel, V = c.vocab;
  const int conv_dim = DI + 2 * G * NS, d_in = 2 * DI + 2 * G * NS + NH;
  char nm[256];
  m->mlayers.resize(c.n_layers);
  for (int l = 0; l < c.n_layers; l++) {
    MambaLayerDev& L = m->mlayers[l];
    snprintf(nm, sizeof nm, "backbone.layers.%d.", l);
    std::string p = nm;
    BZ_TRY(take_vector_f32(m, p + "norm.weight", D, &L.norm));
    BZ_TRY(take_vector_f32(m, p + "mixer.conv1d.weight", conv_dim * KC, &L.conv_w));
    BZ_TRY(take_vector_f32(m, p + "mixer.conv1d.bias", conv_dim, &L.conv_b));
    BZ_TRY(take_vector_f32(m, p + "mixer.dt_bias", NH, &L.dt_bias, false));   // kept as stored (f32 in HF checkpoints)
    if (m->raw.count(p + "mixer.A_log")) BZ_TRY(take_vector_f32(m, p + "mixer.A_log", NH, &L.A_log, false));   // kept as stored (f32 in HF checkpoints)
    else {   // GGUF (llama.cpp convention): ssm_a = A = -exp(A_log); the kernels take A_log
      BZ_TRY(take_vector_f32(m, p + "mixer.A", NH, &L.A_log, false));
      std::vector<float> a(NH);
      BZ_HIP(hipMemcpy(a.data(), L.A_log, (size_t)NH * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < NH; i++) {
        if (!(a[i] < 0.f)) BZ_FAIL(BZ_E_INVALID, "layer %d: mixer.A[%d] = %g is not negative (A = -exp(A_log))", l, i, (double)a[i]);
        a[i] = logf(-a[i]);
      }
      BZ_HIP(hipMemcpy(L.A_log, a.data(), (size_t)NH * 4, hipMemcpyHostToDevice));
    }
    BZ_TRY(take_vector_f32(m, p + "mixer.D", NH, &L.D, false));   // kept as stored (f32 in HF checkpoints)
    BZ_TRY(take_vector_f32(m, p + "mixer.norm.weight", DI, &L.gnorm));
    BZ_TRY(build_fused(m, {p + "mixer.in_proj"}, &L.in_proj));
    BZ_TRY(build_fused(m, {p + "mixer.out_proj"}, &L.out_proj));
    if (L.in_proj.N != d_in || L.in_proj.K != D || L.out_proj.N != D || L.out_proj.K != DI) BZ_FAIL(BZ_E_INVALID, "layer %d: in_proj/out_proj shapes do not match the config", l);
    if (L.in_proj.parts[0].kind != LK_ROWS || L.out_proj.parts[0].kind != LK_ROWS)
      BZ_FAIL(BZ_E_UNSUPPORTED, "mamba2: quantised projections are not implemented (dense f16/bf16/f32)");
  }
  BZ_TRY(take_vector_f32(m, "backbone.norm_f.weight", D, &m->final_norm));
  {
    auto it = m->raw.find("backbone.embeddings.weight");
    if (it == m->raw.end()) BZ_FAIL(BZ_E_NOTFOUND, "finalize: 'backbone.embeddings.weight' was not added");
    RawTensor& r = it->second;
    if (r.kind != 0 || r.N != V || r.K != D) BZ_FAIL(BZ_E_INVALID, "finalize: embeddings must be dense [vocab, hidden]");
    m->embed = r.d0; m->embed_dt = r.dtype; r.d0 = nullptr; m->owned.push_back(m->embed); r.consumed = true;
    m->resident += r.bytes;
    if (c.tie_embeddings || !m->raw.count("lm_head.weight")) {
      LinearDev L; L.kind = LK_ROWS; L.N = V; L.K = D; L.wdt = m->embed_dt; L.w = m->embed; L.owned = false; L.bytes = 0; L.algo_bytes = r.bytes;
      m->lm_head.parts.push_back(L); m->lm_head.n_off.push_back(0); m->lm_head.N = V; m->lm_head.K = D; m->lm_head.fix_out = false;
      m->named["lm_head.weight"] = L;
    } else {
      BZ_TRY(build_fused(m, {"lm_head"}, &m->lm_head));
      force_direct(&m->lm_head);
      if (m->lm_head.fix_out) BZ_FAIL(BZ_E_UNSUPPORTED, "mamba2: quantised lm_head is not implemented");
    }
  }
  for (auto& kv : m->raw) if (!kv.second.consumed) BZ_FAIL(BZ_E_INVALID, "finalize: tensor '%s' is not used by this architecture", kv.first.c_str());
  m->raw.clear();
  void* p;
  m->ring_n = std::max(std::max(d_in, DI), D);
  for (int i = 0; i < 3; i++) {
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 8)); m->ring[i] = (long long*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 8));
    BZ_TRY(dev_alloc(m, &p, (size_t)m->ring_n * 4)); m->dring[i] = (float*)p; BZ_HIP(hipMemset(p, 0, (size_t)m->ring_n * 4));
  }
  for (int i = 0; i < 2; i++) { BZ_TRY(dev_alloc(m, &p, (size_t)D * 4)); m->hbuf[i] = (float*)p; }
  BZ_TRY(dev_alloc(m, &p, (size_t)DI * 4)); m->ybuf = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)NH * 8)); m->vss = (float*)p;      // per head: the exact sum of squares as hi + lo floats
  BZ_TRY(dev_alloc(m, &p, (size_t)V * 4)); m->logits = (float*)p;
  m->nparts = bzk_gemv_rows_blocks(m->lm_head.parts[0]);
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pval = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)m->nparts * 4)); m->pidx = (int*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->tok_tmp = (long long*)p;
  BZ_TRY(dev_alloc(m, &p, 64)); m->pos_tmp = (int*)p;
  const size_t act_b = bz_dtype_size(c.act_dtype);
  m->per_token = (size_t)D * bz_dtype_size(m->embed_dt) + (size_t)D * act_b;
  for (auto& L : m->mlayers) {
    for (FusedLinear* F : {&L.in_proj, &L.out_proj}) for (auto& P : F->parts) { m->resident += P.bytes; m->per_token += P.algo_bytes; }
    m->per_token += (size_t)(conv_dim * (KC + 1) + 3 * NH + DI + D) * act_b;
  }
  for (auto& L : m->lm_head.parts) { m->resident += L.bytes; m->per_token += L.algo_bytes; }
  BZ_HIP(hipDeviceSynchronize());
  m->finalized = true;
  return BZ_OK;
}

extern "C" int bz_ssm_state_create(bz_model* m, int batch, int dtype, bz_ssm_state** out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || m->cfg.arch != BZ_ARCH_MAMBA2 || !out) BZ_FAIL(BZ_E_INVALID, "ssm_state_create: needs a finalized mamba2 model");
  if (batch != 1) BZ_FAIL(BZ_E_UNSUPPORTED, "ssm state: batch %d (single-stream decode only)", batch);
  if (dtype != BZ_F32 && dtype != BZ_F16 && dtype != BZ_BF16) BZ_FAIL(BZ_E_INVALID, "ssm state: dtype %d", dtype);
  if (dtype != m->cfg.act_dtype) BZ_FAIL(BZ_E_UNSUPPORTED, "ssm state dtype must equal the model's activation dtype");
  BZ_HIP(hipSetDevice(m->dev->id));
  const bz_model_config& c = m->cfg;
  bz_ssm_state* s = new bz_ssm_state();
  s->dev = m->dev; s->layers = c.n_layers; s->n_heads = c.ssm_n_heads; s->head_dim = c.ssm_head_dim; s->d_state = c.ssm_d_state;
  s->conv_dim = c.ssm_d_inner + 2 * c.ssm_n_groups * c.ssm_d_state; s->kc = c.ssm_conv_kernel; s->dtype = dtype;
  const size_t sb = (size_t)s->layers * s->n_heads * s->head_dim * s->d_state * bz_dtype_size(dtype);
  const size_t cb = (size_t)s->layers * s->conv_dim * (s->kc - 1) * 4;
  if (hipMalloc(&s->ssm, sb) != hipSuccess || hipMalloc((void**)&s->conv, cb) != hipSuccess) { delete s; BZ_FAIL(BZ_E_OOM, "ssm state: hipMalloc failed"); }
  BZ_HIP(hipMemsetAsync(s->ssm, 0, sb, m->dev->stream));
  BZ_HIP(hipMemsetAsync(s->conv, 0, cb, m->dev->stream));
  bz_dev_retain(m->dev);
  *out = s;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_ssm_state_free(bz_ssm_state* s) {
  BZ_API_BEGIN
  if (!s) return BZ_OK;
  hipStreamSynchronize(s->dev->stream);
  hipFree(s->ssm); hipFree(s->conv);
  bz_dev_release(s->dev);
  delete s;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_ssm_state_reset(bz_ssm_state* s) {
  BZ_API_BEGIN
  if (!s) BZ_FAIL(BZ_E_INVALID, "null state");
  const size_t sb = (size_t)s->layers * s->n_heads * s->head_dim * s->d_state * bz_dtype_size(s->dtype);
  const size_t cb = (size_t)s->layers * s->conv_dim * (s->kc - 1) * 4;
  BZ_HIP(hipMemsetAsync(s->ssm, 0, sb, s->dev->stream));
  BZ_HIP(hipMemsetAsync(s->conv, 0, cb, s->dev->stream));
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// KV caches
// ---------------------------------------------------------------------------------------------------------
extern "C" int bz_kv_create(bz_device* dev, int layers, int batch, int n_kv, int init_cap, int max_len, int hd, int dtype, bz_kv** out) {
  BZ_API_BEGIN
  if (!dev || !out) BZ_FAIL(BZ_E_INVALID, "null argument");
  if (batch != 1) BZ_FAIL(BZ_E_UNSUPPORTED, "kv cache: batch %d (single-stream decode only)", batch);
  if (layers <= 0 || n_kv <= 0 || init_cap <= 0 || max_len < init_cap || hd <= 0) BZ_FAIL(BZ_E_INVALID, "kv cache: bad dimensions");
  if (dtype != BZ_F16 && dtype != BZ_BF16 && dtype != BZ_F32) BZ_FAIL(BZ_E_INVALID, "kv cache: dtype %d", dtype);
  BZ_HIP(hipSetDevice(dev->id));
  bz_kv* kv = new bz_kv();
  kv->dev = dev; kv->layers = layers; kv->n_kv = n_kv; kv->cap = init_cap; kv->max_len = max_len; kv->hd = hd; kv->dtype = dtype;
  size_t bytes = (size_t)layers * n_kv * init_cap * hd * bz_dtype_size(dtype);
  if (hipMalloc(&kv->k, bytes) != hipSuccess || hipMalloc(&kv->v, bytes) != hipSuccess) { delete kv; BZ_FAIL(BZ_E_OOM, "kv cache: hipMalloc(%zu) failed", bytes); }
  BZ_HIP(hipMemsetAsync(kv->k, 0, bytes, dev->stream));
  BZ_HIP(hipMemsetAsync(kv->v, 0, bytes, dev->stream));
  bz_dev_retain(dev);
  *out = kv;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_kv_free(bz_kv* kv) {
  BZ_API_BEGIN
  if (!kv) return BZ_OK;
  hipStreamSynchronize(kv->dev->stream);
  hipFree(kv->k); hipFree(kv->v);
  bz_dev_release(kv->dev);
  delete kv;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_kv_reset(bz_kv* kv) { if (!kv) BZ_FAIL(BZ_E_INVALID, "null kv"); kv->seq_len = 0; return BZ_OK; }
extern "C" int bz_kv_seq_len(const bz_kv* kv) { return kv ? kv->seq_len : -1; }

static int kv_grow(bz_kv* kv, int need) {
  if (need <= kv->cap) return BZ_OK;
  if (need > kv->max_len) BZ_FAIL(BZ_E_INVALID, "kv cache: %d positions needed, max_seq_len is %d", need, kv->max_len);
  int ncap = std::min(kv->max_len, std::max(need, kv->cap * 2));
  const size_t es = bz_dtype_size(kv->dtype);
  const size_t rows = (size_t)kv->layers * kv->n_kv;
  void *nk, *nv;
  size_t bytes = rows * ncap * kv->hd * es;
  if (hipMalloc(&nk, bytes) != hipSuccess || hipMalloc(&nv, bytes) != hipSuccess) BZ_FAIL(BZ_E_OOM, "kv cache grow: hipMalloc(%zu) failed", bytes);
  hipStream_t st = kv->dev->stream;
  BZ_HIP(hipMemsetAsync(nk, 0, bytes, st));
  BZ_HIP(hipMemsetAsync(nv, 0, bytes, st));
  const size_t oldp = (size_t)kv->cap * kv->hd * es, newp = (size_t)ncap * kv->hd * es;
  BZ_HIP(hipMemcpy2DAsync(nk, newp, kv->k, oldp, oldp, rows, hipMemcpyDeviceToDevice, st));
  BZ_HIP(hipMemcpy2DAsync(nv, newp, kv->v, oldp, oldp, rows, hipMemcpyDeviceToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  hipFree(kv->k); hipFree(kv->v);
  kv->k = nk; kv->v = nv; kv->cap = ncap;
  return BZ_OK;
}

static KvView view_of(const bz_kv* kv) {
  KvView v{};
  v.k = kv->k; v.v = kv->v; v.dtype = kv->dtype; v.hd = kv->hd; v.n_kv = kv->n_kv; v.paged = 0;
  v.layer_stride = (long long)kv->n_kv * kv->cap * kv->hd; v.cap = kv->cap; v.bs = 1; v.block_table = nullptr; v.slot = nullptr;
  return v;
}
static KvView view_of(const bz_paged_kv* kv, const int* block_table, const int* slot) {
  KvView v{};
  v.k = kv->k; v.v = kv->v; v.dtype = kv->dtype; v.hd = kv->hd; v.n_kv = kv->n_kv; v.paged = 1;
  v.layer_stride = (long long)kv->num_blocks * kv->n_kv * kv->block_size * kv->hd; v.cap = 0; v.bs = kv->block_size;
  v.block_table = block_table; v.slot = slot;
  return v;
}

extern "C" int bz_kv_read(const bz_kv* kv, int layer, int kvh, int which, int len, float* host) {
  BZ_API_BEGIN
  if (!kv || !host || layer < 0 || layer >= kv->layers || kvh < 0 || kvh >= kv->n_kv || len < 0 || len > kv->cap) BZ_FAIL(BZ_E_INVALID, "kv_read: bad argument");
  float* d;
  BZ_HIP(hipMalloc(&d, std::max<size_t>((size_t)len * kv->hd * 4, 16)));
  int rc = bzk_kv_read(kv->dev->stream, view_of(kv), layer, kvh, which, len, d);
  if (rc == BZ_OK) { hipMemcpyAsync(host, d, (size_t)len * kv->hd * 4, hipMemcpyDeviceToHost, kv->dev->stream); hipStreamSynchronize(kv->dev->stream); }
  hipFree(d);
  return rc;
  BZ_API_END
}

extern "C" int bz_paged_kv_create(bz_device* dev, int layers, int num_blocks, int block_size, int n_kv, int hd, int dtype, bz_paged_kv** out) {
  BZ_API_BEGIN
  if (!dev || !out) BZ_FAIL(BZ_E_INVALID, "null argument");
  if (layers <= 0 || num_blocks <= 0 || block_size <= 0 || n_kv <= 0 || hd <= 0) BZ_FAIL(BZ_E_INVALID, "paged kv: bad dimensions");
  if (dtype != BZ_F16 && dtype != BZ_BF16 && dtype != BZ_F32) BZ_FAIL(BZ_E_INVALID, "paged kv: dtype %d", dtype);
  BZ_HIP(hipSetDevice(dev->id));
  bz_paged_kv* kv = new bz_paged_kv();
  kv->dev = dev; kv->layers = layers; kv->num_blocks = num_blocks; kv->block_size = block_size; kv->n_kv = n_kv; kv->hd = hd; kv->dtype = dtype;
  size_t bytes = (size_t)layers * num_blocks * n_kv * block_size * hd * bz_dtype_size(dtype);
  if (hipMalloc(&kv->k, bytes) != hipSuccess || hipMalloc(&kv->v, bytes) != hipSuccess) { delete kv; BZ_FAIL(BZ_E_OOM, "paged kv: hipMalloc(%zu) failed", bytes); }
  BZ_HIP(hipMemsetAsync(kv->k, 0, bytes, dev->stream));
  BZ_HIP(hipMemsetAsync(kv->v, 0, bytes, dev->stream));
  bz_dev_retain(dev);
  *out = kv;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_paged_kv_free(bz_paged_kv* kv) {
  BZ_API_BEGIN
  if (!kv) return BZ_OK;
  hipStreamSynchronize(kv->dev->stream);
  hipFree(kv->k); hipFree(kv->v);
  bz_dev_release(kv->dev);
  delete kv;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_paged_kv_set_seq_len(bz_paged_kv* kv, int n) { if (!kv || n < 0) BZ_FAIL(BZ_E_INVALID, "bad argument"); kv->seq_len = n; return BZ_OK; }
extern "C" int bz_paged_kv_seq_len(const bz_paged_kv* kv) { return kv ? kv->seq_len : -1; }

// ---------------------------------------------------------------------------------------------------------
// the decode step (one token through all layers), built from the fused kernels
// ---------------------------------------------------------------------------------------------------------
__global__ void k_set_int(int* p, int v) { p[0] = v; }

// Graph capture records a step on a private stream, so that other host threads can keep using the device stream meanwhile (a capturing
// stream refuses their work).  The step functions ask step_stream() instead of reading the device stream directly.
static thread_local hipStream_t tl_capture_stream = nullptr;
static hipStream_t step_stream(const bz_model* m) { return tl_capture_stream ? tl_capture_stream : m->dev->stream; }

// contexts beyond this many positions take the split-KV attention path (two launches; the fused single launch wins below it)
static int att_split_min() { const char* e = getenv("BZ_SPLIT_MIN"); return e ? atoi(e) : 512; }   // read per call: tests move it
static int att_positions_for(int len) { return len > att_split_min() ? len : 0; }

struct StepIO {
  KvView kv;
  const long long* d_tok;   // token id (device)
  const int* d_pos;         // position (device)
  // pieces API: when set, the step starts from a given hidden row (+ optional prev) instead of the embedding
  const float* hidden_in = nullptr; const float* prev_in = nullptr;
  int layer_start = 0, layer_end = -1;
  bool do_embed = true, do_head = true;
  float* hidden_out = nullptr; float* prev_out = nullptr;   // pieces API outputs (f32 rows)
  FinalArgs* final_args = nullptr;                          // graph mode: fused argmax + bookkeeping
  bz_ssm_state* ssm = nullptr;                              // Mamba2: recurrent state instead of a KV cache
  int att_positions = 0;    // > 0: long-context step -- split-KV attention sized for this many positions (eager: position + 1; graph: capacity)
};

// Fixed-point accumulator ring.  Launch j accumulates into ring[j % 3] (which must be zero), reads the output of
// launch j-1 and zeroes ring[(j+1) % 3] -- last written by launch j-2 and last read by launch j-1, both complete by
// stream order.  `dirty[i]` = number of non-zero entries ring[i] may hold.
struct RingState { int ri = 0; int dirty[3] = {0, 0, 0}; };

// Launch every part of a fused linear.  Returns the VSrc describing its output.
static int run_fused(bz_model* m, const FusedLinear& F, Pro pro, RingState& rs, VSrc* out, const ConvShift* shift = nullptr) {
  hipStream_t st = step_stream(m);
  const int act = m->cfg.act_dtype;
  const int ri = rs.ri, rz = (rs.ri + 1) % 3;
  long long* acc = m->ring[ri];
  float* direct = m->dring[ri];
  if (F.parts.size() == 1 && F.fix_out && bzk_gemv_cols_ok(F.parts[0], pro, act)) {
    // the whole K inside one workgroup per 64-column tile: finished values, stored directly (no accumulator to fill or to zero for THIS launch;
    // the ring protocol's zeroing duty for the buffer after next stays)
    BZ_TRY(bzk_gemv_cols(st, F.parts[0], pro, direct, rs.dirty[rz] > 0 ? m->ring[rz] : nullptr, rs.dirty[rz]));
    rs.dirty[rz] = 0; rs.dirty[ri] = 0; rs.ri = rz;
    out->fix = 0; out->p = (const void*)direct;
    return BZ_OK;
  }
  static const bool no_mix = getenv("BZ_NO_GQ_MIX") != nullptr;
  if (!no_mix && F.parts.size() == 2 && F.fix_out && F.n_off[1] == F.parts[0].N && bzk_gq_mix_ok(F.parts[0], F.parts[1], pro)) {
    // GGUF Q4_K_M q/k/v: the Q4_K part (q, k) and the Q6_K part (v) in one launch
    GemvOut o{};
    o.acc = acc; o.zero_buf = rs.dirty[rz] > 0 ? m->ring[rz] : nullptr; o.zero_n = rs.dirty[rz];
    BZ_TRY(bzk_gemv_gq_mix(st, F.parts[0], F.parts[1], pro, o));
    rs.dirty[rz] = 0; rs.dirty[ri] = F.N; rs.ri = rz;
    out->fix = 1; out->p = (const void*)acc;
    return BZ_OK;
  }
  for (size_t i = 0; i < F.parts.size(); i++) {
    const LinearDev& L = F.parts[i];
    Pro p = pro; p.perm = L.perm;
    GemvOut o{};
    o.acc = acc + F.n_off[i]; o.direct = direct + F.n_off[i];
    o.zero_buf = (i == 0 && rs.dirty[rz] > 0) ? m->ring[rz] : nullptr; o.zero_n = rs.dirty[rz];
    if (i == 0 && shift) o.shift = *shift;
    BZ_TRY(bzk_gemv(st, L, p, o, act));
  }
  rs.dirty[rz] = 0;
  rs.dirty[ri] = F.fix_out ? F.N : 0;
  rs.ri = rz;
  out->fix = F.fix_out ? 1 : 0;
  out->p = F.fix_out ? (const void*)acc : (const void*)direct;
  return BZ_OK;
}

static int llama_step(bz_model* m, const StepIO& io) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = step_stream(m);
  const int H = c.hidden, I = c.inter, act = c.act_dtype;
  const int lend = io.layer_end < 0 ? c.n_layers : io.layer_end;
  int cur = 0;
  RingState rs;
  VSrc prev{nullptr, 0};
  if (io.do_embed) {
    BZ_TRY(bzk_embed(st, m->embed, m->embed_dt, io.d_tok, H, act, m->hbuf[cur], io.d_pos, m->cos_t, m->sin_t, c.head_dim / 2, m->rope_cur));
  } else {
    if (io.d_pos && lend > io.layer_start) BZ_TRY(bzk_rope_row(st, io.d_pos, m->cos_t, m->sin_t, c.head_dim / 2, m->rope_cur));
    BZ_HIP(hipMemcpyAsync(m->hbuf[cur], io.hidden_in, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    if (io.prev_in) { prev.p = io.prev_in; prev.fix = 0; }
  }
  // The layers as ONE persistent launch (bz_persist.hip) -- OPT-IN (BZ_PERSIST=1): measured in round 3 it is slower than the three launches per layer
  // (profiles/r03_persist_stamps.txt, DESIGN 8: a grid barrier on a CU that has weight requests in flight waits behind them, 5-10 us each).  Applies when the
  // model qualifies, the context fits the single-launch attention and the step starts without a deferred residual.  Bit-identical to the launch-per-phase
  // path below (same arithmetic, integer accumulators): tests/test_gpu_persist.py.
  static const bool no_persist = getenv("BZ_PERSIST") == nullptr || getenv("BZ_NO_PERSIST") != nullptr;
  int l_first = io.layer_start;
  if (!no_persist && m->persist_tab && prev.p == nullptr && lend > io.layer_start && io.att_positions == 0 && io.kv.dtype == BZ_F16 && io.kv.hd == c.head_dim && io.d_pos) {
    BzPersistLaunch pl{};
    pl.layers = (const char*)m->persist_tab + bzk_persist_layer_bytes() * io.layer_start; pl.n_layers = lend - io.layer_start;
    pl.h_in = m->hbuf[cur]; pl.h_out = m->hbuf[cur ^ 1];
    pl.ring_m = m->ring[0]; pl.ring_q = m->ring[1]; pl.ring_o = m->ring[2];
    pl.rope_cur = m->rope_cur; pl.pos = io.d_pos; pl.kv = io.kv;
    // the K / V bases of the table's first layer: the kernel indexes layers from 0
    pl.kv.k = (char*)io.kv.k + (size_t)io.layer_start * io.kv.layer_stride * bz_dtype_size(io.kv.dtype);
    pl.kv.v = (char*)io.kv.v + (size_t)io.layer_start * io.kv.layer_stride * bz_dtype_size(io.kv.dtype);
    pl.bar = m->dev->persist_bar; pl.err_host = (unsigned*)m->dev->persist_err; pl.eps = c.rms_eps; pl.I = I;
    double bytes = 0.0;
    for (int l = io.layer_start; l < lend; l++)
      for (const FusedLinear* F : {&m->layers[l].qkv, &m->layers[l].o, &m->layers[l].gateup, &m->layers[l].down}) bytes += (double)F->parts[0].algo_bytes;
    pl.algo_bytes = bytes;
    // diagnostic: per-wave phase stamps of layer 1 (workgroups 0 and 131), printed after the launch (eager steps only)
    static const bool pstamps_on = getenv("BZ_PERSIST_STAMPS") != nullptr;
    static long long* pstamps = nullptr; static int pstamp_prints = 0;
    if (pstamps_on && !tl_capture_stream && pstamp_prints < 3) { if (!pstamps) hipMalloc(&pstamps, 2 * 8 * 32 * 8); hipMemsetAsync(pstamps, 0, 2 * 8 * 32 * 8, st); pl.stamps = pstamps; }
    BZ_TRY(bzk_llama_persist(st, pl));
    if (pl.stamps) {
      std::vector<long long> hs(2 * 8 * 32);
      hipStreamSynchronize(st); hipMemcpy(hs.data(), pstamps, hs.size() * 8, hipMemcpyDeviceToHost); pstamp_prints++;
      const long long t0 = hs[0];
      for (int g = 0; g < 2; g++)
        for (int w = 0; w < 8; w++) {
          fprintf(stderr, "[bz] persist stamps wg %3d wave %d (us since wg 0 wave 0 entered phase Q of layer 1):", g ? 131 : 0, w);
          for (int i = 0; i < 18; i++) fprintf(stderr, " %d:%.2f", i, hs[(g * 8 + w) * 32 + i] ? (hs[(g * 8 + w) * 32 + i] - t0) / 100.0 : -1.0);
          fprintf(stderr, "\n");
        }
      { double d; float f1, f2; memcpy(&d, &hs[24], 8); int i1 = (int)hs[25], i2 = (int)hs[26]; memcpy(&f1, &i1, 4); memcpy(&f2, &i2, 4);
        fprintf(stderr, "[bz] persist row update (last one of the launch, wg 0): ssd %.17g ss %.9g rs %.9g; wave sums:", d, f1, f2);
        for (int w8 = 0; w8 < 8; w8++) { memcpy(&d, &hs[32 + w8], 8); fprintf(stderr, " %.17g", d); }
        fprintf(stderr, "\n"); }
    }
    cur ^= 1;
    // ring state after the launch: ring[0] = the last MLP's output (the deferred residual), ring[1] zeroed in its last phase, ring[2] read but not zeroed
    prev = VSrc{m->ring[0], 1};
    rs.ri = 1; rs.dirty[0] = H; rs.dirty[1] = 0; rs.dirty[2] = H;
    l_first = lend;
  }
  for (int l = l_first; l < lend; l++) {
    const LayerDev& Ld = m->layers[l];
    Pro pn{}; pn.mode = PRO_NORM; pn.src = prev; pn.h_in = m->hbuf[cur]; pn.h_out = m->hbuf[cur ^ 1]; pn.norm_w = Ld.attn_norm;
    pn.eps = c.rms_eps; pn.H = H; pn.act = act;
    static long long* qkv_stamps = nullptr;
    static const bool qkv_stamps_on = getenv("BZ_QKV_STAMPS") != nullptr;
    if (qkv_stamps_on) {
      if (!qkv_stamps) { hipMalloc(&qkv_stamps, 256); hipMemset(qkv_stamps, 0, 256); }
      if (l == 2) {
        long long hst[16]; hipStreamSynchronize(st); hipMemcpy(hst, qkv_stamps, 128, hipMemcpyDeviceToHost);
        fprintf(stderr, "[bz] qkv stamps us since block 0 entry: first block");
        for (int q = 1; q <= 5; q++) fprintf(stderr, " %d:%.2f", q, (hst[q] - hst[0]) / 100.0);
        fprintf(stderr, " | last block entry %.2f", (hst[8] - hst[0]) / 100.0);
        for (int q = 1; q <= 5; q++) fprintf(stderr, " %d:%.2f", q, (hst[8 + q] - hst[0]) / 100.0);
        fprintf(stderr, "\n");
      }
      pn.stamps = l == 1 ? qkv_stamps : nullptr;
    }
    VSrc qkv;
    BZ_TRY(run_fused(m, Ld.qkv, pn, rs, &qkv));
    cur ^= 1;
    {   // diagnostic (scripts/qkv_dump.py): the raw q/k/v accumulator of one layer, written to a file (eager steps only)
      static const char* dump = getenv("BZ_DUMP_QKV");
      static const int dump_layer = getenv("BZ_DUMP_LAYER") ? atoi(getenv("BZ_DUMP_LAYER")) : 0;
      if (dump && !tl_capture_stream && l == dump_layer && qkv.fix) {
        std::vector<long long> hq(Ld.qkv.N);
        hipStreamSynchronize(st); hipMemcpy(hq.data(), qkv.p, hq.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(dump, "wb"); if (f) { fwrite(hq.data(), 8, hq.size(), f); fclose(f); }
      }
    }

    AttnArgs aa{};
    aa.qkv = qkv; aa.cos_t = m->cos_t; aa.sin_t = m->sin_t; aa.interleaved = c.rope_interleaved; aa.pos = io.d_pos; aa.rope_cur = m->rope_cur;
    aa.nq = c.n_heads; aa.nkv = c.n_kv_heads; aa.hd = c.head_dim; aa.act = act; aa.kv = io.kv; aa.layer = l; aa.out = m->attn_out;
    aa.zero_buf = nullptr; aa.zero_n = 0; aa.q_only = 0;
    static long long* attn_stamps = nullptr;
    static const bool attn_stamps_on = getenv("BZ_ATTN_STAMPS") != nullptr, attn_stamps_print = getenv("BZ_ATTN_STAMPS_PRINT") != nullptr;
    if (attn_stamps_on) {
      if (!attn_stamps) { hipMalloc(&attn_stamps, 512); hipMemset(attn_stamps, 0, 512); }
      if (l == 1) {
        if (attn_stamps_print) {
          long long hst[64]; hipStreamSynchronize(st); hipMemcpy(hst, attn_stamps, 512, hipMemcpyDeviceToHost);
          fprintf(stderr, "[bz] attn stamps (us since entry):");
          for (int q = 1; q <= 8; q++) fprintf(stderr, " %d:%.2f", q, (hst[q] - hst[0]) / 100.0);
          fprintf(stderr, "\n");
        }
      }
      aa.stamps = l == 0 ? attn_stamps : nullptr;
    }
    VSrc ov;
    static const bool no_fuse = getenv("BZ_NO_ATTN_FUSION") != nullptr;
    // (the f32-cache form has no split-KV partner: beyond the single-launch contexts it stays unfused)
    const bool fuse_cap = !no_fuse && Ld.o.parts.size() == 1 && Ld.o.fix_out && bzk_attn_oproj_slices(aa, Ld.o.parts[0]) > 0;
    static const bool no_split = getenv("BZ_NO_ATTN_SPLIT") != nullptr;
    const bool split_wanted = !no_split && io.att_positions > 0 && m->att_ws && bzk_attn_split_ok(aa);
    const bool merge_fused = fuse_cap && Ld.o.parts[0].kind == LK_Q4G && bzk_attn_merge_oproj_ok(aa, Ld.o.parts[0]);
    // long contexts: split-KV attention merged with the o_proj where that form exists (int4); otherwise attention (split or not) and a separate o_proj launch --
    // the fused single-launch kernel reads the context once per column slice, which only pays while the context is short
    const bool fuse_o = fuse_cap && (io.att_positions == 0 || (split_wanted && merge_fused));
    const bool split = split_wanted && (!fuse_o || merge_fused);
    int SPL = 0, nsplit = 0;
    if (split) {
      // long context: split-KV partials (all query heads of a group share the K/V rows), then merge (+ o_proj)
      bzk_attn_split_plan(io.att_positions, &SPL, &nsplit);
      BZ_TRY(bzk_attn_split(st, aa, SPL, nsplit, m->att_ws));
    }
    if (fuse_o) {
      // attention + o_proj in one launch: same ring protocol as a GEMV launch
      const int rz = (rs.ri + 1) % 3;
      aa.zero_buf = rs.dirty[rz] > 0 ? m->ring[rz] : nullptr; aa.zero_n = rs.dirty[rz];
      if (split) BZ_TRY(bzk_attn_merge_oproj(st, aa, m->att_ws, SPL, nsplit, Ld.o.parts[0], m->ring[rs.ri]));
      else BZ_TRY(bzk_attn_oproj(st, aa, Ld.o.parts[0], m->ring[rs.ri]));
      ov = VSrc{m->ring[rs.ri], 1};
      rs.dirty[rz] = 0; rs.dirty[rs.ri] = Ld.o.N; rs.ri = rz;
    } else {
      if (split) BZ_TRY(bzk_attn_merge(st, aa, m->att_ws, SPL, nsplit));
      else BZ_TRY(bzk_attn_decode(st, aa));
      Pro pp{}; pp.mode = PRO_PLAIN; pp.src = VSrc{m->attn_out, 0}; pp.act = act; pp.H = 0;
      BZ_TRY(run_fused(m, Ld.o, pp, rs, &ov));
    }

    Pro pf{}; pf.mode = PRO_NORM; pf.src = ov; pf.h_in = m->hbuf[cur]; pf.h_out = m->hbuf[cur ^ 1]; pf.norm_w = Ld.ffn_norm;
    pf.eps = c.rms_eps; pf.H = H; pf.act = act;
    VSrc dn;
    static const bool no_mlp_fuse = getenv("BZ_NO_MLP_FUSION") != nullptr;
    if (!no_mlp_fuse && act == BZ_F16 && Ld.gateup.parts.size() == 1 && Ld.down.parts.size() == 1 && bzk_mlp_fusable(Ld.gateup.parts[0], Ld.down.parts[0], H, I)) {
      // norm + gate/up + SiLU*up + down in one launch (same ring protocol as one GEMV launch)
      const int rz = (rs.ri + 1) % 3;
      BZ_TRY(bzk_mlp_q4g(st, Ld.gateup.parts[0], Ld.down.parts[0], H, I, pf, m->ring[rs.ri], rs.dirty[rz] > 0 ? m->ring[rz] : nullptr, rs.dirty[rz]));
      dn = VSrc{m->ring[rs.ri], 1};
      rs.dirty[rz] = 0; rs.dirty[rs.ri] = H; rs.ri = rz;
      cur ^= 1;
    } else if (!no_mlp_fuse && Ld.gateup.parts.size() == 1 && Ld.down.parts.size() == 1 && bzk_mlp_dense_fusable(Ld.gateup.parts[0], Ld.down.parts[0], Ld.down_slabs, H, I, act)) {
      // the dense 16-bit form of the same fusion (slab-major down_proj copy)
      const int rz = (rs.ri + 1) % 3;
      BZ_TRY(bzk_mlp_dense(st, Ld.gateup.parts[0], Ld.down.parts[0], Ld.down_slabs, H, I, pf, m->ring[rs.ri], rs.dirty[rz] > 0 ? m->ring[rz] : nullptr, rs.dirty[rz]));
      dn = VSrc{m->ring[rs.ri], 1};
      rs.dirty[rz] = 0; rs.dirty[rs.ri] = H; rs.ri = rz;
      cur ^= 1;
    } else if (!no_mlp_fuse && Ld.gateup.parts.size() == 1 && Ld.down.parts.size() == 1 && bzk_mlp_gq_fusable(Ld.gateup.parts[0], Ld.down.parts[0], H, I, act)) {
      // the GGUF form of the same fusion (Q4_K gate / up, Q4_K or Q6_K down, f32 activations)
      const int rz = (rs.ri + 1) % 3;
      BZ_TRY(bzk_mlp_gq(st, Ld.gateup.parts[0], Ld.down.parts[0], H, I, pf, m->ring[rs.ri], rs.dirty[rz] > 0 ? m->ring[rz] : nullptr, rs.dirty[rz]));
      dn = VSrc{m->ring[rs.ri], 1};
      rs.dirty[rz] = 0; rs.dirty[rs.ri] = H; rs.ri = rz;
      cur ^= 1;
    } else {
      VSrc gu;
      BZ_TRY(run_fused(m, Ld.gateup, pf, rs, &gu));
      cur ^= 1;
      Pro ps{}; ps.mode = PRO_SILU; ps.src = gu; ps.H = I; ps.act = act;
      BZ_TRY(run_fused(m, Ld.down, ps, rs, &dn));
    }
    prev = dn;
  }
  if (io.hidden_out) {
    BZ_HIP(hipMemcpyAsync(io.hidden_out, m->hbuf[cur], (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    if (io.prev_out && prev.p) {
      if (prev.fix) BZ_TRY(bzk_fix_to_f32(st, (const long long*)prev.p, H, act, io.prev_out));
      else BZ_HIP(hipMemcpyAsync(io.prev_out, prev.p, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  // ring indices at this point: rs.ri is clean; (rs.ri+2)%3 holds `prev` (still needed by the head), (rs.ri+1)%3 is stale
  const int ra = (rs.ri + 2) % 3, rb = (rs.ri + 1) % 3;
  if (io.do_head) {
    Pro ph{}; ph.mode = PRO_NORM; ph.src = prev; ph.h_in = m->hbuf[cur]; ph.h_out = nullptr; ph.norm_w = m->final_norm;
    ph.eps = c.rms_eps; ph.H = H; ph.act = act;
    int rfin = ra;   // ring buffer the final kernel zeroes
    if (m->lm_head.fix_out) {
      // quantised lm_head (GGUF output.weight): split-K GEMV into the ring, then convert + partial argmax
      VSrc lv;
      BZ_TRY(run_fused(m, m->lm_head, ph, rs, &lv));
      BZ_TRY(bzk_fix_to_f32(st, (const long long*)lv.p, c.vocab, act, m->logits));
      BZ_TRY(bzk_argmax_partials(st, m->logits, c.vocab, m->pval, m->pidx, m->nparts));
      rfin = -1;
    } else {
      GemvOut o{};
      o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
      o.zero_buf = rs.dirty[rb] > 0 ? m->ring[rb] : nullptr; o.zero_n = rs.dirty[rb];
      BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
      rs.dirty[rb] = 0;
    }
    if (io.final_args) {
      FinalArgs fa = *io.final_args;
      fa.pval = m->pval; fa.pidx = m->pidx; fa.nparts = m->nparts;
      if (rfin >= 0) { fa.zero_buf = rs.dirty[rfin] > 0 ? m->ring[rfin] : nullptr; fa.zero_n = rs.dirty[rfin]; }
      BZ_TRY(bzk_argmax_final(st, fa));
      if (rfin >= 0) rs.dirty[rfin] = 0;
    }
  }
  // every ring buffer must be zero again when the step ends
  for (int i = 0; i < 3; i++) if (rs.dirty[i] > 0) BZ_TRY(bzk_zero64(st, m->ring[i], rs.dirty[i]));
  return BZ_OK;
}


// Mamba2 decode step (forward_with_ssm_state, /root/reference/src/engine/executor_generate.rs:137,148), 3 launches per layer:
//   in_proj GEMV (prologue: residual add + RMSNorm) -> [conv1d step + SiLU + SSM recurrence + gate] -> out_proj GEMV (prologue: grouped RMSNorm)
static int mamba_step(bz_model* m, const StepIO& io) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = step_stream(m);
  const int D = c.hidden, DI = c.ssm_d_inner, NH = c.ssm_n_heads, NS = c.ssm_d_state, G = c.ssm_n_groups, KC = c.ssm_conv_kernel, act = c.act_dtype;
  const int conv_dim = DI + 2 * G * NS;
  bz_ssm_state* S = io.ssm;
  int cur = 0;
  RingState rs;
  VSrc prev{nullptr, 0};
  BZ_TRY(bzk_embed(st, m->embed, m->embed_dt, io.d_tok, D, act, m->hbuf[cur]));
  for (int l = 0; l < c.n_layers; l++) {
    const MambaLayerDev& L = m->mlayers[l];
    Pro pn{}; pn.mode = PRO_NORM; pn.src = prev; pn.h_in = m->hbuf[cur]; pn.h_out = m->hbuf[cur ^ 1]; pn.norm_w = L.norm; pn.eps = c.rms_eps; pn.H = D; pn.act = act;
    VSrc zx;
    BZ_TRY(run_fused(m, L.in_proj, pn, rs, &zx));
    cur ^= 1;
    // conv1d step + SiLU + SSM recurrence + gate in one launch (reads the projection from the ring: f32 or fixed point)
    SsmArgs sa{};
    sa.zx = zx; sa.z_off = 0; sa.x_off = DI; sa.dt_off = DI + conv_dim; sa.dt_bias = L.dt_bias; sa.A_log = L.A_log; sa.D = L.D;
    sa.conv_w = L.conv_w; sa.conv_b = L.conv_b; sa.conv_state = S->conv + (size_t)l * conv_dim * (KC - 1); sa.conv_kernel = KC;
    sa.state = (char*)S->ssm + (size_t)l * NH * c.ssm_head_dim * NS * bz_dtype_size(S->dtype); sa.sdt = S->dtype;
    sa.n_heads = NH; sa.head_dim = c.ssm_head_dim; sa.d_state = NS; sa.n_groups = G; sa.d_inner = DI; sa.act = act; sa.y = m->ybuf; sa.gate = 1; sa.vss = m->vss;
    BZ_TRY(bzk_ssm_step(st, sa));
    Pro pg{}; pg.mode = PRO_GATED2; pg.src = VSrc{m->ybuf, 0}; pg.h_in = m->vss; pg.aux2 = NH; pg.norm_w = L.gnorm; pg.eps = c.rms_eps; pg.H = DI; pg.act = act; pg.aux = G;
    VSrc ov;
    // side duty of this launch: the B / C channels' conv state moves on by this step's projection (every head's SSM step has read the old one)
    ConvShift shf{sa.conv_state, zx, DI, DI, 2 * G * NS, KC};
    BZ_TRY(run_fused(m, L.out_proj, pg, rs, &ov, &shf));
    prev = ov;
  }
  // ring indices as in llama_step: rs.ri clean, (rs.ri+2)%3 holds `prev` when it is fixed point, (rs.ri+1)%3 stale
  const int ra = (rs.ri + 2) % 3, rb = (rs.ri + 1) % 3;
  if (io.do_head) {
    Pro ph{}; ph.mode = PRO_NORM; ph.src = prev; ph.h_in = m->hbuf[cur]; ph.h_out = nullptr; ph.norm_w = m->final_norm; ph.eps = c.rms_eps; ph.H = D; ph.act = act;
    GemvOut o{};
    o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
    o.zero_buf = rs.dirty[rb] > 0 ? m->ring[rb] : nullptr; o.zero_n = rs.dirty[rb];
    BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
    rs.dirty[rb] = 0;
    if (io.final_args) {
      FinalArgs fa = *io.final_args;
      fa.pval = m->pval; fa.pidx = m->pidx; fa.nparts = m->nparts;
      fa.zero_buf = rs.dirty[ra] > 0 ? m->ring[ra] : nullptr; fa.zero_n = rs.dirty[ra];
      BZ_TRY(bzk_argmax_final(st, fa));
      rs.dirty[ra] = 0;
    }
  }
  for (int i = 0; i < 3; i++) if (rs.dirty[i] > 0) BZ_TRY(bzk_zero64(st, m->ring[i], rs.dirty[i]));
  return BZ_OK;
}

// DeepSeek-V2 decode step: per layer  [q_proj ; kv_a] GEMV (norm prologue) -> MLA attention over the latent cache -> o_proj GEMV ->
//   dense layer: gate/up GEMV (norm prologue) -> down GEMV (SiLU prologue)
//   MoE layer:   router (norm, top-k) -> grouped gate/up GEMV over the selected + shared experts -> grouped down GEMV -> combine
static int dsv2_step(bz_model* m, const StepIO& io) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = step_stream(m);
  const int H = c.hidden, NH = c.n_heads, R = c.mla_kv_lora_rank, DN = c.mla_nope_dim, DR = c.mla_rope_dim, DV = c.mla_v_dim, act = c.act_dtype;
  const int E = c.moe_n_experts, TK = c.moe_top_k, NS = c.moe_n_shared, MI = c.moe_inter;
  int cur = 0;
  RingState rs;
  VSrc prev{nullptr, 0};
  BZ_TRY(bzk_embed(st, m->embed, m->embed_dt, io.d_tok, H, act, m->hbuf[cur]));
  for (int l = 0; l < c.n_layers; l++) {
    const DsLayerDev& L = m->dlayers[l];
    Pro pn{}; pn.mode = PRO_NORM; pn.src = prev; pn.h_in = m->hbuf[cur]; pn.h_out = m->hbuf[cur ^ 1]; pn.norm_w = L.attn_norm; pn.eps = c.rms_eps; pn.H = H; pn.act = act; pn.f32_sums = 0;
    VSrc qkva;
    BZ_TRY(run_fused(m, L.qkva, pn, rs, &qkva));
    cur ^= 1;
    MlaArgs ma{};
    if (c.mla_q_lora_rank > 0) {
      // q = q_b_proj(q_a_layernorm(q_a)): a second GEMV whose prologue is the RMSNorm of the first one's leading q_lora_rank outputs (no residual);
      // the latent | k_pe part of the first GEMV's output reaches the attention kernel through `kva`
      Pro pq{}; pq.mode = PRO_NORM; pq.src = VSrc{nullptr, 0}; pq.h_in = (const float*)qkva.p; pq.h_out = nullptr; pq.norm_w = L.q_norm; pq.eps = c.rms_eps; pq.H = c.mla_q_lora_rank; pq.act = act; pq.f32_sums = 0;
      VSrc qv;
      BZ_TRY(run_fused(m, L.q_b, pq, rs, &qv));
      ma.kva = (const float*)qkva.p + c.mla_q_lora_rank;
      qkva = qv;
    }
    ma.qkv = qkva; ma.kv_norm = L.kv_norm; ma.eps = c.rms_eps; ma.wkvb = L.kv_b; ma.wdt = L.kv_b_dt; ma.cos_t = m->cos_t; ma.sin_t = m->sin_t; ma.pos = io.d_pos;
    ma.n_heads = NH; ma.rank = R; ma.nope = DN; ma.rope = DR; ma.vdim = DV; ma.act = act; ma.kv = io.kv; ma.layer = l; ma.out = m->attn_out;
    ma.scale = mla_softmax_scale(c);
    ma.ws = m->mla_ws; ma.nsplit = m->mla_nsplit;
    // exact decode (16-bit models): every sum as the oracle defines it, one maximum over the whole context (BZ_DSV2_F32_SUMS=1: the f32 kernels)
    static const bool f32_mla = getenv("BZ_DSV2_F32_SUMS") != nullptr;
    if (!f32_mla && (act == BZ_F16 || act == BZ_BF16) && bzk_mla_x_ok(ma, c.max_seq_len)) BZ_TRY(bzk_mla_attn_x(st, ma, c.max_seq_len, m->mla_wsd, m->mla_sync, (unsigned*)m->dev->persist_err, m->mla_scw, m->mla_mxw));
    else BZ_TRY(bzk_mla_attn(st, ma, c.max_seq_len));
    Pro pp{}; pp.mode = PRO_PLAIN; pp.src = VSrc{m->attn_out, 0}; pp.act = act; pp.H = 0; pp.f32_sums = 0;
    VSrc ov;
    BZ_TRY(run_fused(m, L.o, pp, rs, &ov));
    Pro pf{}; pf.mode = PRO_NORM; pf.src = ov; pf.h_in = m->hbuf[cur]; pf.h_out = m->hbuf[cur ^ 1]; pf.norm_w = L.ffn_norm; pf.eps = c.rms_eps; pf.H = H; pf.act = act; pf.f32_sums = 0;
    if (!L.is_moe) {
      VSrc gu, dn;
      BZ_TRY(run_fused(m, L.gateup, pf, rs, &gu));
      cur ^= 1;
      Pro ps{}; ps.mode = PRO_SILU; ps.src = gu; ps.H = c.inter; ps.act = act; ps.f32_sums = 0;
      BZ_TRY(run_fused(m, L.down, ps, rs, &dn));
      prev = dn;
    } else {
      const int slots = TK + NS;
      const size_t es = bz_dtype_size(L.e_dt);
      static const bool no_fr = getenv("BZ_NO_MOE_ROUTE_FUSION") != nullptr;
      if (!no_fr && bzk_moe_rows2_ok(L.e_dt, H) && bzk_moe_rows2_ok(L.e_dt, MI) && bzk_moe_rows2_ok(L.router_dt, H) && E <= 1024 && slots <= 128) {
        // router logits = one more fixed-point GEMV of the ring (norm prologue: writes h'); the top-k runs in the prologue of the gate / up launch, which
        // repeats the norm for its own x (the same inputs: h and the o_proj accumulator, both still in place)
        LinearDev RL; RL.kind = LK_ROWS; RL.N = E; RL.K = H; RL.wdt = L.router_dt; RL.w = L.router; RL.sk = 2; RL.owned = false; RL.algo_bytes = (size_t)E * H * bz_dtype_size(L.router_dt);
        FusedLinear RF; RF.parts.push_back(RL); RF.n_off.push_back(0); RF.N = E; RF.K = H; RF.fix_out = true;
        VSrc lg;
        Pro pr = pf; pr.f32_sums = 0;      // the router's logits are exact sums on both paths (prompt rows: k_moe_route_rows), so a token routes the same way in a prompt and in a decode step
        BZ_TRY(run_fused(m, RF, pr, rs, &lg));
        cur ^= 1;
        Pro pg = pf; pg.h_out = nullptr;
        MoeGemvArgs g1{};
        g1.w = L.e_gu; g1.expert_stride = (long long)2 * MI * H; g1.sel = nullptr; g1.N = 2 * MI; g1.K = H; g1.src_stride = 0;
        g1.acc = m->moe_gu_acc; g1.acc_stride = 2 * MI; g1.acc_slots = slots;
        g1.route = RouteArgs{(const long long*)lg.p, E, TK, NS, c.moe_routed_scale, c.moe_norm_topk, m->moe_sel, m->moe_w};
        static const bool moe_stamps = getenv("BZ_MOE_STAMPS") != nullptr;   // diagnostic: phase stamps of the route + gate/up launch (layer 2, printed once)
        static long long* stamp_buf = nullptr; static int stamp_prints = 0;
        if (moe_stamps && l == 2 && stamp_prints < 3 && !tl_capture_stream) { if (!stamp_buf) { hipMalloc(&stamp_buf, 64); } hipMemsetAsync(stamp_buf, 0, 64, st); pg.stamps = stamp_buf; }
        BZ_TRY(bzk_moe_gemv(st, g1, L.e_dt, slots, pg, act, true, (double)slots * 2 * MI * H * es));
        if (pg.stamps) {
          long long hs[8]; hipStreamSynchronize(st); hipMemcpy(hs, stamp_buf, 64, hipMemcpyDeviceToHost); stamp_prints++;
          fprintf(stderr, "[bz] route+gate/up stamps (us since entry): rendezvous %.2f, x published %.2f, top-k done %.2f | tile wave: range 0 done %.2f, ids seen %.2f, end %.2f\n",
                  (hs[1] - hs[0]) / 100.0, (hs[2] - hs[0]) / 100.0, (hs[3] - hs[0]) / 100.0, (hs[4] - hs[0]) / 100.0, (hs[5] - hs[0]) / 100.0, (hs[6] - hs[0]) / 100.0);
        }
        MoeGemvArgs g2{};
        g2.w = L.e_dn; g2.expert_stride = (long long)H * MI; g2.sel = m->moe_sel; g2.N = H; g2.K = MI; g2.src_stride = 2 * MI;
        g2.acc = m->moe_acc; g2.acc_stride = H; g2.acc_slots = TK + 1;
        Pro p2{}; p2.mode = PRO_SILU; p2.src = VSrc{m->moe_gu_acc, 1}; p2.H = MI; p2.act = act;
        BZ_TRY(bzk_moe_gemv(st, g2, L.e_dt, slots, p2, act, true, (double)slots * H * MI * es));
        BZ_TRY(bzk_moe_combine(st, m->moe_acc, m->moe_w, TK, NS > 0, H, act, m->moe_out, m->moe_gu_acc, slots * 2 * MI));
        prev = VSrc{m->moe_out, 0};
        continue;
      }
      BZ_TRY(bzk_moe_router(st, pf, L.router, L.router_dt, E, TK, NS, c.moe_routed_scale, c.moe_norm_topk, m->moe_xn, m->moe_sel, m->moe_w, m->moe_lg, m->moe_cnt));
      cur ^= 1;
      // gate / up of the selected + shared experts in ONE launch, down in one more.  Balanced role kernel (fixed-point accumulators both times) when the
      // expert weights are 16-bit; else the 16-row workgroup form with a direct f32 gate / up
      const bool r2 = bzk_moe_rows2_ok(L.e_dt, H) && bzk_moe_rows2_ok(L.e_dt, MI);
      MoeGemvArgs g1{};
      g1.w = L.e_gu; g1.expert_stride = (long long)2 * MI * H; g1.sel = m->moe_sel; g1.N = 2 * MI; g1.K = H; g1.src_stride = 0;
      g1.out = m->moe_gu; g1.out_stride = 2 * MI;
      if (r2) { g1.acc = m->moe_gu_acc; g1.acc_stride = 2 * MI; g1.acc_slots = slots; }
      Pro p1{}; p1.mode = PRO_PLAIN; p1.src = VSrc{m->moe_xn, 0}; p1.act = act;
      BZ_TRY(bzk_moe_gemv(st, g1, L.e_dt, slots, p1, act, r2, (double)slots * 2 * MI * H * es));
      MoeGemvArgs g2{};
      g2.w = L.e_dn; g2.expert_stride = (long long)H * MI; g2.sel = m->moe_sel; g2.N = H; g2.K = MI; g2.src_stride = 2 * MI;
      g2.acc = m->moe_acc; g2.acc_stride = H; g2.acc_slots = TK + 1;
      Pro p2{}; p2.mode = PRO_SILU; p2.src = r2 ? VSrc{m->moe_gu_acc, 1} : VSrc{m->moe_gu, 0}; p2.H = MI; p2.act = act;
      BZ_TRY(bzk_moe_gemv(st, g2, L.e_dt, slots, p2, act, true, (double)slots * H * MI * es));
      BZ_TRY(bzk_moe_combine(st, m->moe_acc, m->moe_w, TK, NS > 0, H, act, m->moe_out, r2 ? m->moe_gu_acc : nullptr, slots * 2 * MI));
      prev = VSrc{m->moe_out, 0};
    }
  }
  const int ra = (rs.ri + 2) % 3, rb = (rs.ri + 1) % 3;
  if (io.do_head) {
    Pro ph{}; ph.mode = PRO_NORM; ph.src = prev; ph.h_in = m->hbuf[cur]; ph.h_out = nullptr; ph.norm_w = m->final_norm; ph.eps = c.rms_eps; ph.H = H; ph.act = act;
    GemvOut o{};
    o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
    o.zero_buf = rs.dirty[rb] > 0 ? m->ring[rb] : nullptr; o.zero_n = rs.dirty[rb];
    BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
    rs.dirty[rb] = 0;
    if (io.final_args) {
      FinalArgs fa = *io.final_args;
      fa.pval = m->pval; fa.pidx = m->pidx; fa.nparts = m->nparts;
      fa.zero_buf = rs.dirty[ra] > 0 ? m->ring[ra] : nullptr; fa.zero_n = rs.dirty[ra];
      BZ_TRY(bzk_argmax_final(st, fa));
      rs.dirty[ra] = 0;
    }
  }
  for (int i = 0; i < 3; i++) if (rs.dirty[i] > 0) BZ_TRY(bzk_zero64(st, m->ring[i], rs.dirty[i]));
  return BZ_OK;
}

static int model_step(bz_model* m, const StepIO& io) {
  if (m->cfg.arch == BZ_ARCH_MAMBA2) return mamba_step(m, io);
  if (m->cfg.arch == BZ_ARCH_DEEPSEEK2) return dsv2_step(m, io);
  return llama_step(m, io);
}

static int check_fwd(bz_model* m, const bz_tensor* tokens, int S) {
  if (!m || !m->finalized) BZ_FAIL(BZ_E_INVALID, "forward: model not finalized");
  if (!tokens || tokens->dtype != BZ_I64 || tokens->nbytes < (size_t)S * 8 || S <= 0) BZ_FAIL(BZ_E_INVALID, "forward: tokens must be an I64 tensor with >= S elements");
  BZ_HIP(hipSetDevice(m->dev->id));
  return BZ_OK;
}

static int emit_logits(bz_model* m, bz_tensor* logits_out, int row) {
  const size_t vb = (size_t)m->cfg.vocab * 4;
  if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(row + 1) * vb) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
  BZ_HIP(hipMemcpyAsync((char*)logits_out->ptr + (size_t)row * vb, m->logits, vb, hipMemcpyDeviceToDevice, m->dev->stream));
  return BZ_OK;
}


// ---------------------------------------------------------------------------------------------------------
// batched prefill on the matrix cores (dense f16 / bf16 Llama-family models): SURVEY.md 8 row K4
// ---------------------------------------------------------------------------------------------------------
static int find_linear(bz_model* m, const char* name, LinearDev* out);
static int prefill_min_rows() {
  static const int v = getenv("BZ_NO_MFMA_PREFILL") ? (1 << 30) : (getenv("BZ_PREFILL_MIN") ? atoi(getenv("BZ_PREFILL_MIN")) : 8);
  return v;
}
static bool prefill_eligible(const bz_model* m, int S, int total_len, bool decode_batch = false) {
  const bz_model_config& c = m->cfg;
  static const bool no_gq = getenv("BZ_NO_GGUF_PREFILL") != nullptr;
  if (c.arch != BZ_ARCH_LLAMA || S < prefill_min_rows() || (c.act_dtype != BZ_F16 && c.act_dtype != BZ_BF16 && (c.act_dtype != BZ_F32 || no_gq))) return false;
  if (c.hidden % 64 || (c.n_heads * c.head_dim) % 64 || c.inter % 64 || c.head_dim % 8 || 256 % (c.head_dim / 8)) return false;
  const int rep = c.n_heads / c.n_kv_heads;
  if (rep != 1 && rep != 2 && rep != 4 && rep != 8) return false;
  if (!bzk_pf_attn_mfma_ok(c.head_dim, rep) && bzk_pf_attn_smem(c.n_heads, c.n_kv_heads, c.head_dim, total_len) > 160 * 1024) return false;   // (the scalar kernel keeps scores in LDS)
  // every projection either dense in the activation dtype (MFMA GEMM) or int4 without act-order (multi-row dot4 GEMM)
  if (c.act_dtype == BZ_F32) {
    // f32 activations (GGUF models): every projection in a block format, no bias, same K in all parts of a fused linear; the head stays the decode GEMV
    for (const LayerDev& L : m->layers)
      for (const FusedLinear* F : {&L.qkv, &L.o, &L.gateup, &L.down}) {
        if (F->parts.empty()) return false;
        unsigned long long ntot = 0;
        for (const LinearDev& P : F->parts) { if (!bzk_gq_split_ok(P) || P.bias || P.K != F->parts[0].K) return false; ntot += (unsigned long long)P.N; }
        // the split operands go through ONE 16-bit GEMM over 3 K whose element offsets are 32-bit (bzk_gemm_nt)
        if (ntot * 3ull * (unsigned long long)F->parts[0].K >= (1ull << 32) || (unsigned long long)std::min(S, 2048) * 3ull * (unsigned long long)F->parts[0].K >= (1ull << 32)) return false;
      }
    return true;
  }
  bool any_dense = false;
  for (const LayerDev& L : m->layers)
    for (const FusedLinear* F : {&L.qkv, &L.o, &L.gateup, &L.down}) {
      if (F->parts.size() != 1) return false;
      const LinearDev& P = F->parts[0];
      const bool dense_ok = P.kind == LK_ROWS && P.wdt == c.act_dtype, q4_ok = bzk_gemm_q4g_rows_ok(P);
      if (!dense_ok && !q4_ok) return false;
      any_dense = any_dense || dense_ok;
    }
  // dense 16-bit models: the decode GEMVs carry exact sums (piece_dot_d), the MFMA GEMMs f32 ones -- prompts of up to BZ_EXACT_PREFILL_MAX (16) rows stay on the
  // decode kernels, token by token, so that their rows are the oracle's bits like the int4 models' exact rows (bz_host.hip prefill_exact; BZ_EXACT_PREFILL=0: never)
  if (any_dense && !decode_batch) {
    static const int env = getenv("BZ_EXACT_PREFILL") ? atoi(getenv("BZ_EXACT_PREFILL")) : -1;
    static const int max_rows = getenv("BZ_EXACT_PREFILL_MAX") ? atoi(getenv("BZ_EXACT_PREFILL_MAX")) : 16;
    if (env != 0 && (env > 0 || S <= max_rows)) return false;
  }
  return m->lm_head.parts.size() == 1 && m->lm_head.parts[0].kind == LK_ROWS && !m->lm_head.fix_out;
}
// split-K partials of the prefill GEMMs (bzk_gemm_nt / bzk_gemm_q4g_mfma)
static int ensure_pf_ws(bz_model* m) {
  if (m->pf_ws) return BZ_OK;
  void* p;
  m->pf_ws_bytes = (size_t)48 << 20;
  BZ_TRY(dev_alloc(m, &p, m->pf_ws_bytes));
  m->pf_ws = (float*)p;
  return BZ_OK;
}
static int prefill_ws(bz_model* m, int rows) {
  if (m->pf_rows >= rows) return BZ_OK;
  const bz_model_config& c = m->cfg;
  const size_t qn = (size_t)(c.n_heads + 2 * c.n_kv_heads) * c.head_dim, xw = std::max<size_t>(std::max<size_t>(c.hidden, c.inter), (size_t)c.n_heads * c.head_dim);
  BZ_HIP(hipStreamSynchronize(m->dev->stream));
  void* p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * c.hidden * 4)); m->pf_h = (float*)p;     // (earlier, smaller buffers stay owned until the model is freed)
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * c.hidden * 4)); m->pf_t = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * qn * 4)); m->pf_qkv = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * 2 * c.inter * 4)); m->pf_gu = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * xw * (c.act_dtype == BZ_F32 ? 4 : 2))); m->pf_x16 = p;
  if (c.act_dtype == BZ_F32) {
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * 3 * xw * 2)); m->pf_x3 = p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * 4)); m->pf_rscale = (float*)p;
  }
  if (!m->pf_acc) { const size_t an = 8 * std::max<size_t>(std::max<size_t>(qn, 2 * (size_t)c.inter), c.hidden); BZ_TRY(dev_alloc(m, &p, an * 8)); m->pf_acc = (long long*)p; BZ_HIP(hipMemset(p, 0, an * 8)); }
  BZ_TRY(ensure_pf_ws(m));
  m->pf_rows = rows;
  return BZ_OK;
}

// Y[n][N] = R(X16[n][K] . W^T): dense 16-bit weights on the matrix cores, int4 weights through the multi-row dot4 GEMM
// exact: 0 = the f16 MFMA GEMM, 1 = the multi-row form of the decode kernels (8 rows per pass over the weights), 2 = the integer-MFMA GEMM (exact group sums)
static int pf_gemm(bz_model* m, const LinearDev& P, const void* x16, int n, float* y, int exact = 0) {
  hipStream_t st = step_stream(m);
  const int act = m->cfg.act_dtype;
  if (P.kind == LK_ROWS) return bzk_gemm_nt(st, act, x16, P.w, P.bias, n, P.N, P.K, act, y, m->pf_ws, m->pf_ws_bytes);
  if (exact == 2 && bzk_gemm_q4g_i8_ok(P, act)) {
    size_t po; const size_t need = bzk_pf_quant_i8_bytes(n, P.K, &po);
    if (m->pf_xq_bytes < need) { BZ_HIP(hipStreamSynchronize(st)); void* p; BZ_TRY(dev_alloc(m, &p, need)); m->pf_xq = p; m->pf_xq_bytes = need; }
    BZ_TRY(bzk_pf_quant_i8(st, x16, n, P.K, m->pf_xq));
    return bzk_gemm_q4g_i8(st, P, m->pf_xq, n, act, y);
  }
  if (exact) return bzk_gemm_q4g_rows(st, P, act, x16, n, act, m->pf_acc, y);
  // (dequantising an int4 linear to f16 once per chunk and running the f16 LDS-DMA GEMM was built and measured: 2048-token prompt 90 -> 65 ms, but
  //  R16((q - z) s) costs 1.4e-4 relative per GEMM and put awq 8B-width logits at 1.40e-3 against the 1e-3 bar -- removed)
  if (bzk_gemm_q4g_mfma_ok(P, act, n)) return bzk_gemm_q4g_mfma(st, P, x16, n, act, y, m->pf_ws, m->pf_ws_bytes);
  return bzk_gemm_q4g_rows(st, P, act, x16, n, act, m->pf_acc, y);
}

// Y[n][N] = X[n][K] . dequant(F)^T for a block-format fused linear, f32 rows in (m->pf_x16 as float), f32 rows out: both operands as three f16 pieces
// (bz_prefill.hip: k_pf_split3 / k_gq_split3) through ONE 16-bit MFMA GEMM over 3 K.  The split weights are built on first use and kept while the budget
// (BZ_GGUF_PREFILL_CACHE_GB, default 96) lasts; past it they are rebuilt into a scratch matrix per call.
static int pf_gemm_gq(bz_model* m, const FusedLinear& F, const float* xf, int n, float* y) {
  hipStream_t st = step_stream(m);
  const int K = F.parts[0].K;
  int N = 0;
  for (const LinearDev& P : F.parts) N += P.N;
  bz_model::GqPf& e = m->gq_pf[&F];
  void* p;
  if (!e.wscale) {
    if (!m->gq_amax) { BZ_TRY(dev_alloc(m, &p, 16)); m->gq_amax = (unsigned*)p; }
    BZ_TRY(dev_alloc(m, &p, 16)); e.wscale = (float*)p;
    BZ_HIP(hipMemsetAsync(m->gq_amax, 0, 4, st));
    for (const LinearDev& P : F.parts) BZ_TRY(bzk_gq_absmax(st, P, m->gq_amax));
    BZ_TRY(bzk_gq_wscale(st, m->gq_amax, e.wscale));
  }
  const size_t bytes = (size_t)N * 3 * K * 2;
  void* w3 = e.w3;
  if (!w3) {
    static const size_t budget = (size_t)(getenv("BZ_GGUF_PREFILL_CACHE_GB") ? atof(getenv("BZ_GGUF_PREFILL_CACHE_GB")) : 96.0) << 30;
    size_t free_b = 0, total_b = 0;
    const bool keep = m->gq_cache_bytes + bytes <= budget && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > bytes + ((size_t)8 << 30);
    if (keep) { BZ_TRY(dev_alloc(m, &p, bytes)); e.w3 = w3 = p; m->gq_cache_bytes += bytes; }
    else {
      if (m->gq_w3_scratch_bytes < bytes) { BZ_HIP(hipStreamSynchronize(st)); BZ_TRY(dev_alloc(m, &p, bytes)); m->gq_w3_scratch = p; m->gq_w3_scratch_bytes = bytes; }
      w3 = m->gq_w3_scratch;
    }
    int row0 = 0;
    for (const LinearDev& P : F.parts) { BZ_TRY(bzk_gq_split3(st, P, e.wscale, w3, row0)); row0 += P.N; }
  }
  BZ_TRY(bzk_pf_split3(st, xf, n, K, m->pf_x3, m->pf_rscale, e.wscale));
  return bzk_gemm_nt(st, BZ_F16, m->pf_x3, w3, nullptr, n, N, 3 * K, BZ_F32, y, m->pf_ws, m->pf_ws_bytes, m->pf_rscale);
}

// EXACT prompt rows (f16 int4 models): the multi-row form of the decode kernels' integer arithmetic for every projection, the exact scalar attention and
// the decode lm_head -- a prompt row is then the same bits as the decode step would produce for it (and as the oracle's: tests/test_gpu_exact_prefill.py).
// One pass over the weights serves 8 rows and costs ~3.9 ms at the 8B AWQ shape (13 launches per layer), the MFMA path has a floor of ~4.6 ms: up to
// BZ_EXACT_PREFILL_MAX rows (default 16: two passes) exactness is nearly free (32 rows: 15.8 vs 4.8 ms); longer prompts take the matrix cores, whose f32 accumulation order differs from the exact sums: ~1e-6 per GEMM, which the f16 roundings
// of a deep model amplify to the f16 noise floor (profiles/r03_prefill_parity_depth.txt).  BZ_EXACT_PREFILL=1 forces the exact rows for every prompt length.
// returns 0 (MFMA f16 path), 1 (exact rows) or 2 (exact integer-MFMA GEMMs: BZ_EXACT_PREFILL=2 for every prompt, or prompts of up to BZ_EXACT_I8_MAX rows)
static int prefill_exact(const bz_model* m, int n, bool decode_batch) {
  static const int env = getenv("BZ_EXACT_PREFILL") ? atoi(getenv("BZ_EXACT_PREFILL")) : -1;     // -1: by prompt length
  static const int max_rows = getenv("BZ_EXACT_PREFILL_MAX") ? atoi(getenv("BZ_EXACT_PREFILL_MAX")) : 16;
  static const int max_i8 = getenv("BZ_EXACT_I8_MAX") ? atoi(getenv("BZ_EXACT_I8_MAX")) : 0;
  if (env == 0 || decode_batch || m->cfg.act_dtype != BZ_F16) return 0;
  int mode = 0;
  if (env == 1) mode = 1; else if (env >= 2) mode = n <= 16 ? 1 : 2;
  else if (n <= max_rows) mode = 1; else if (n <= max_i8) mode = 2;
  if (!mode) return 0;
  for (const LayerDev& L : m->layers)
    for (const FusedLinear* F : {&L.qkv, &L.o, &L.gateup, &L.down}) {
      if (F->parts.size() != 1 || !bzk_gemm_q4g_rows_ok(F->parts[0])) return 0;
      if (mode == 2 && !bzk_gemm_q4g_i8_ok(F->parts[0], BZ_F16)) mode = 1;
    }
  return mode;
}

// tokens [S] at positions pos0 .. pos0+S-1; `slots` (paged only): device i32 [S].  Logits of the last row (or all rows) -> logits_out.
// per-row context of a decode batch: row r is its own sequence (position row_pos[r], block-table row r); nullptr row_pos = one prompt
struct RowsCtx { const int* row_pos = nullptr; int table_stride = 0; int max_len = 0; };

static int prefill_dense(bz_model* m, const long long* d_tok, int S, const KvView& view, int pos0, const int* slots, bool all, bz_tensor* logits_out,
                         const RowsCtx& rc = RowsCtx()) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = step_stream(m);     // (a batched decode graph records this function on its capture stream)
  const int H = c.hidden, I = c.inter, nq = c.n_heads, nkv = c.n_kv_heads, hd = c.head_dim, act = c.act_dtype, dt = c.act_dtype;
  const int CH = rc.row_pos ? 512 : 2048;      // prompts: 2048-row chunks (larger GEMMs: 8B AWQ 2048-token prompt 105 -> 90 ms)
  BZ_TRY(prefill_ws(m, std::min(S, CH)));
  const int exact = prefill_exact(m, S, rc.row_pos != nullptr);
  const bool attn_exact = exact != 0 || act == BZ_F32;       // f32 models: the oracle's double-precision sums cost little next to the f32 cache reads
  for (int s0 = 0; s0 < S; s0 += CH) {
    const int n = std::min(CH, S - s0), p0 = pos0 + s0;
    // decode batch (row_pos set): one block-table row per sequence -- the kernels index rows by the row number INSIDE the chunk, so the chunk
    // starting at sequence s0 gets the table advanced to its first row (sequences >= 512 read another sequence's blocks)
    KvView vw = view;
    if (rc.row_pos && vw.block_table) vw.block_table = view.block_table + (size_t)s0 * rc.table_stride;
    BZ_TRY(bzk_pf_embed(st, m->embed, m->embed_dt, d_tok + s0, n, H, act, m->pf_h));
    const float* prev = nullptr;
    for (int l = 0; l < c.n_layers; l++) {
      const LayerDev& L = m->layers[l];
      const bool gq = act == BZ_F32;      // block-format projections: f32 rows, split operands
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, prev, L.attn_norm, n, H, c.rms_eps, act, m->pf_x16));
      BZ_TRY(gq ? pf_gemm_gq(m, L.qkv, (const float*)m->pf_x16, n, m->pf_qkv) : pf_gemm(m, L.qkv.parts[0], m->pf_x16, n, m->pf_qkv, exact));
      BZ_TRY(bzk_pf_rope_kv(st, m->pf_qkv, n, nq, nkv, hd, m->cos_t, m->sin_t, c.rope_interleaved, p0, act, vw, l, slots ? slots + s0 : nullptr,
                            rc.row_pos ? rc.row_pos + s0 : nullptr));
      BZ_TRY(bzk_pf_attn(st, dt, m->pf_qkv, n, nq, nkv, hd, p0, act, vw, l, m->pf_x16, rc.row_pos ? rc.row_pos + s0 : nullptr, rc.table_stride, rc.max_len, attn_exact));
      BZ_TRY(gq ? pf_gemm_gq(m, L.o, (const float*)m->pf_x16, n, m->pf_t) : pf_gemm(m, L.o.parts[0], m->pf_x16, n, m->pf_t, exact));
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, m->pf_t, L.ffn_norm, n, H, c.rms_eps, act, m->pf_x16));
      BZ_TRY(gq ? pf_gemm_gq(m, L.gateup, (const float*)m->pf_x16, n, m->pf_gu) : pf_gemm(m, L.gateup.parts[0], m->pf_x16, n, m->pf_gu, exact));
      BZ_TRY(bzk_pf_silu(st, dt, m->pf_gu, n, I, act, m->pf_x16));
      BZ_TRY(gq ? pf_gemm_gq(m, L.down, (const float*)m->pf_x16, n, m->pf_t) : pf_gemm(m, L.down.parts[0], m->pf_x16, n, m->pf_t, exact));
      prev = m->pf_t;
    }
    // head.  Several rows wanted (all_logits, decode batch) and a dense lm_head in the activation dtype: final norm rows + one MFMA GEMM that
    // streams the lm_head once; otherwise the decode lm_head GEMV per row (final norm fused as its prologue)
    const LinearDev& LH = m->lm_head.parts[0];
    if (all && n > 1 && !exact && act != BZ_F32 && LH.wdt == act && LH.K % 64 == 0 && logits_out->nbytes >= (size_t)(s0 + n) * c.vocab * 4) {
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, prev, m->final_norm, n, H, c.rms_eps, act, m->pf_x16));
      BZ_TRY(bzk_gemm_nt(st, act, m->pf_x16, LH.w, LH.bias, n, c.vocab, H, act, (float*)logits_out->ptr + (size_t)s0 * c.vocab, m->pf_ws, m->pf_ws_bytes));
      continue;
    }
    for (int r = 0; r < n; r++) {
      const int srow = s0 + r;
      if (!all && srow != S - 1) continue;
      if (m->lm_head.fix_out || m->lm_head.parts.size() != 1) {   // quantised lm_head (GGUF output.weight): the decode step's head on this row
        StepIO hio{};
        hio.do_embed = false; hio.do_head = true; hio.layer_start = 0; hio.layer_end = 0;
        hio.hidden_in = m->pf_h + (size_t)r * H; hio.prev_in = prev + (size_t)r * H;
        BZ_TRY(llama_step(m, hio));
        BZ_TRY(emit_logits(m, logits_out, all ? srow : 0));
        continue;
      }
      Pro ph{}; ph.mode = PRO_NORM; ph.src = VSrc{prev + (size_t)r * H, 0}; ph.h_in = m->pf_h + (size_t)r * H; ph.h_out = nullptr; ph.norm_w = m->final_norm;
      ph.eps = c.rms_eps; ph.H = H; ph.act = act;
      GemvOut o{};
      o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
      BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
      BZ_TRY(emit_logits(m, logits_out, all ? srow : 0));
    }
  }
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DeepSeek-V2 batched prefill (BASELINE.json configs[4]: "prefill 512 (MFMA) + decode 128"): every projection of the prompt rows is one GEMM on
// the matrix cores (k_gemm_nt), the MLA attention runs the decode kernel's arithmetic with one workgroup per (head, token) over the latent
// cache that a row-wise kernel filled first, and the routed experts become per-expert GEMMs over gathered token rows
// (docs/architecture.md:108-119: softmax -> top-k -> weighted sum + shared experts).  Same rounding points as the decode step, token for token.
// ---------------------------------------------------------------------------------------------------------
static bool dsv2_prefill_eligible(const bz_model* m, int S, int total_len, const KvView& view) {
  const bz_model_config& c = m->cfg;
  if (c.arch != BZ_ARCH_DEEPSEEK2 || S < prefill_min_rows() || (c.act_dtype != BZ_F16 && c.act_dtype != BZ_BF16) || c.mla_q_lora_rank > 0) return false;
  {   // short prompts stay on the decode kernels (exact sums), as for the other families (prefill_eligible)
    static const int env = getenv("BZ_EXACT_PREFILL") ? atoi(getenv("BZ_EXACT_PREFILL")) : -1;
    static const int max_rows = getenv("BZ_EXACT_PREFILL_MAX") ? atoi(getenv("BZ_EXACT_PREFILL_MAX")) : 16;
    if (env != 0 && (env > 0 || S <= max_rows)) return false;
  }
  if (c.hidden % 64 || (c.n_heads * c.mla_v_dim) % 64 || (c.moe_n_experts > 0 && c.moe_inter % 64) || view.dtype != c.act_dtype) return false;
  for (const DsLayerDev& L : m->dlayers) {
    for (const FusedLinear* F : {&L.qkva, &L.o}) if (F->parts.size() != 1 || F->parts[0].kind != LK_ROWS || F->parts[0].wdt != c.act_dtype) return false;
    if (L.kv_b_dt != c.act_dtype) return false;
    if (!L.is_moe) { for (const FusedLinear* F : {&L.gateup, &L.down}) if (F->parts.size() != 1 || F->parts[0].kind != LK_ROWS || F->parts[0].wdt != c.act_dtype || c.inter % 64) return false; }
    else if (L.e_dt != c.act_dtype) return false;
  }
  MlaArgs probe{}; probe.rank = c.mla_kv_lora_rank; probe.rope = c.mla_rope_dim; probe.nope = c.mla_nope_dim; probe.vdim = c.mla_v_dim;
  extern size_t bzk_mla_smem(const MlaArgs&, int);
  if (bzk_mla_smem(probe, total_len) > 160 * 1024) return false;
  return m->lm_head.parts.size() == 1 && m->lm_head.parts[0].kind == LK_ROWS && !m->lm_head.fix_out;
}
static int dsv2_prefill_ws(bz_model* m, int rows) {
  if (m->dpf_rows >= rows) return BZ_OK;
  const bz_model_config& c = m->cfg;
  const int H = c.hidden, NH = c.n_heads, TK = std::max(c.moe_top_k, 1), MI = std::max(c.moe_inter, 1);
  const size_t qn = (size_t)NH * (c.mla_nope_dim + c.mla_rope_dim) + c.mla_kv_lora_rank + c.mla_rope_dim;
  const size_t xw = std::max<size_t>(std::max<size_t>(H, (size_t)NH * c.mla_v_dim), std::max<size_t>(c.inter, MI));
  BZ_HIP(hipStreamSynchronize(m->dev->stream));
  void* p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * H * 4)); m->pf_h = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * H * 4)); m->pf_t = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * qn * 4)); m->pf_qkv = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * 2 * std::max(c.inter, 1) * 4)); m->pf_gu = (float*)p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * xw * 2)); m->pf_x16 = p;
  BZ_TRY(dev_alloc(m, &p, (size_t)rows * NH * c.mla_v_dim * 4)); m->dpf_att = (float*)p;
  if (c.moe_n_experts > 0) {
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * H * 2)); m->dpf_xg16 = p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * 2 * MI * 4)); m->dpf_gu = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * MI * 2)); m->dpf_a16 = p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * H * 4)); m->dpf_ye = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * H * 4)); m->dpf_ysh = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * 4)); m->dpf_sel = (int*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * 4)); m->dpf_w = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)c.moe_n_experts * 4)); m->dpf_cnt = (int*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)c.moe_n_experts * 4)); m->dpf_off = (int*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * 4)); m->dpf_rowof = (int*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * TK * 4)); m->dpf_tokof = (int*)p;
  }
  m->dpf_rows = rows;
  return BZ_OK;
}
__global__ void k_acc_rows(float* acc, const float* y, size_t n, int first) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[i] = first ? y[i] : acc[i] + y[i];
}
static int dsv2_prefill(bz_model* m, const long long* d_tok, int S, const KvView& view, int pos0, bool all, bz_tensor* logits_out) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = m->dev->stream;
  const int H = c.hidden, NH = c.n_heads, R = c.mla_kv_lora_rank, DN = c.mla_nope_dim, DR = c.mla_rope_dim, DV = c.mla_v_dim, act = c.act_dtype, dt = c.act_dtype;
  const int E = c.moe_n_experts, TK = c.moe_top_k, NS = c.moe_n_shared, MI = c.moe_inter;
  const int NQ = NH * (DN + DR), QN = NQ + R + DR;
  const size_t es = bz_dtype_size(dt);
  const int CH = 512;
  BZ_TRY(dsv2_prefill_ws(m, std::min(S, CH)));
  BZ_TRY(ensure_pf_ws(m));
  std::vector<int> hcnt(std::max(E, 1));
  for (int s0 = 0; s0 < S; s0 += CH) {
    const int n = std::min(CH, S - s0), p0 = pos0 + s0;
    BZ_TRY(bzk_pf_embed(st, m->embed, m->embed_dt, d_tok + s0, n, H, act, m->pf_h));
    const float* prev = nullptr;
    for (int l = 0; l < c.n_layers; l++) {
      const DsLayerDev& L = m->dlayers[l];
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, prev, L.attn_norm, n, H, c.rms_eps, act, m->pf_x16));
      const LinearDev& PQ = L.qkva.parts[0];
      BZ_TRY(bzk_gemm_nt(st, dt, m->pf_x16, PQ.w, PQ.bias, n, QN, H, act, m->pf_qkv, m->pf_ws, m->pf_ws_bytes));
      BZ_TRY(bzk_mla_append_rows(st, m->pf_qkv + NQ, QN, n, L.kv_norm, c.rms_eps, R, DR, m->cos_t, m->sin_t, p0, act, view, l));
      MlaArgs ma{};
      ma.qkv = VSrc{m->pf_qkv, 0}; ma.kv_norm = L.kv_norm; ma.eps = c.rms_eps; ma.wkvb = L.kv_b; ma.wdt = L.kv_b_dt; ma.cos_t = m->cos_t; ma.sin_t = m->sin_t; ma.pos = nullptr;
      ma.n_heads = NH; ma.rank = R; ma.nope = DN; ma.rope = DR; ma.vdim = DV; ma.act = act; ma.kv = view; ma.layer = l; ma.out = m->dpf_att;
      ma.scale = mla_softmax_scale(c);
      ma.batch = n; ma.pos0 = p0; ma.q_stride = QN; ma.out_stride = (long long)NH * DV;
      BZ_TRY(bzk_mla_attn(st, ma, p0 + n));
      BZ_TRY(bzk_pf_cvt16(st, dt, m->dpf_att, (size_t)n * NH * DV, m->pf_x16));
      const LinearDev& PO = L.o.parts[0];
      BZ_TRY(bzk_gemm_nt(st, dt, m->pf_x16, PO.w, PO.bias, n, H, NH * DV, act, m->pf_t, m->pf_ws, m->pf_ws_bytes));
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, m->pf_t, L.ffn_norm, n, H, c.rms_eps, act, m->pf_x16));
      if (!L.is_moe) {
        const LinearDev& PG = L.gateup.parts[0]; const LinearDev& PD = L.down.parts[0];
        BZ_TRY(bzk_gemm_nt(st, dt, m->pf_x16, PG.w, PG.bias, n, 2 * c.inter, H, act, m->pf_gu, m->pf_ws, m->pf_ws_bytes));
        BZ_TRY(bzk_pf_silu(st, dt, m->pf_gu, n, c.inter, act, m->pf_x16));
        BZ_TRY(bzk_gemm_nt(st, dt, m->pf_x16, PD.w, PD.bias, n, H, c.inter, act, m->pf_t, m->pf_ws, m->pf_ws_bytes));
      } else {
        // routing, per-expert row lists (one small device -> host copy per layer: the row counts size the GEMM launches)
        BZ_TRY(bzk_moe_route_rows(st, dt, m->pf_x16, n, H, L.router, L.router_dt, E, TK, c.moe_routed_scale, c.moe_norm_topk, m->dpf_sel, m->dpf_w));
        BZ_TRY(bzk_moe_plan_rows(st, m->dpf_sel, n, TK, E, m->dpf_cnt, m->dpf_off, m->dpf_rowof, m->dpf_tokof));
        BZ_TRY(bzk_moe_gather_rows(st, m->pf_x16, m->dpf_tokof, n * TK, H, m->dpf_xg16));
        const size_t gu_sz = (size_t)2 * MI * H * es, dn_sz = (size_t)H * MI * es;
        // the grouped GEMMs' grid is sized for the largest expert's row count, read back from the device: one stream synchronisation per MoE layer.  (Sizing the grid
        // for the bound "every token picks this expert" instead -- no read-back, row tiles past an expert's count exit at once -- was built and measured: 512-token
        // prompt of V2-Lite 25.2 -> 28.3 ms; eight times the workgroups, seven in eight of them empty, cost more than the 26 host round trips.)
        BZ_HIP(hipMemcpyAsync(hcnt.data(), m->dpf_cnt, (size_t)E * 4, hipMemcpyDeviceToHost, st));
        BZ_HIP(hipStreamSynchronize(st));
        int maxc = 0;
        for (int e = 0; e < E; e++) maxc = std::max(maxc, hcnt[e]);
        // all the experts of the layer in ONE grouped launch per projection (per-expert launches were weight-bandwidth-bound on 44 workgroups each:
        // 126 launches x 43 us per layer, profiles/r02_dsv2_prefill_kernel_stats_before.csv)
        BZ_TRY(bzk_gemm_nt_grouped(st, dt, m->dpf_xg16, L.e_gu, (long long)2 * MI * H, E, m->dpf_off, m->dpf_cnt, maxc, (long long)n * TK, 2 * MI, H, act, m->dpf_gu));
        BZ_TRY(bzk_pf_silu(st, dt, m->dpf_gu, n * TK, MI, act, m->dpf_a16));
        BZ_TRY(bzk_gemm_nt_grouped(st, dt, m->dpf_a16, L.e_dn, (long long)H * MI, E, m->dpf_off, m->dpf_cnt, maxc, (long long)n * TK, H, MI, act, m->dpf_ye));
        // the shared experts: one MLP of width NS * moe_inter stored as NS expert-shaped slots; the slots' down products are summed unrounded
        for (int j = 0; j < NS; j++) {
          BZ_TRY(bzk_gemm_nt(st, dt, m->pf_x16, (const char*)L.e_gu + (size_t)(E + j) * gu_sz, nullptr, n, 2 * MI, H, act, m->dpf_gu, m->pf_ws, m->pf_ws_bytes));
          BZ_TRY(bzk_pf_silu(st, dt, m->dpf_gu, n, MI, act, m->dpf_a16));
          BZ_TRY(bzk_gemm_nt(st, dt, m->dpf_a16, (const char*)L.e_dn + (size_t)(E + j) * dn_sz, nullptr, n, H, MI, BZ_F32, j == 0 ? m->dpf_ysh : m->pf_t, m->pf_ws, m->pf_ws_bytes));
          if (j > 0) hipLaunchKernelGGL(k_acc_rows, dim3(1024), dim3(256), 0, st, m->dpf_ysh, m->pf_t, (size_t)n * H, 0);
        }
        BZ_TRY(bzk_moe_combine_rows(st, m->dpf_ye, m->dpf_rowof, m->dpf_w, NS > 0 ? m->dpf_ysh : nullptr, n, TK, H, act, m->pf_t));
      }
      prev = m->pf_t;
    }
    const LinearDev& LH = m->lm_head.parts[0];
    if (all && n > 1 && LH.wdt == act && LH.K % 64 == 0 && logits_out->nbytes >= (size_t)(s0 + n) * c.vocab * 4) {
      BZ_TRY(bzk_pf_norm(st, dt, m->pf_h, prev, m->final_norm, n, H, c.rms_eps, act, m->pf_x16));
      BZ_TRY(bzk_gemm_nt(st, act, m->pf_x16, LH.w, LH.bias, n, c.vocab, H, act, (float*)logits_out->ptr + (size_t)s0 * c.vocab, m->pf_ws, m->pf_ws_bytes));
      continue;
    }
    for (int r = 0; r < n; r++) {
      const int srow = s0 + r;
      if (!all && srow != S - 1) continue;
      Pro ph{}; ph.mode = PRO_NORM; ph.src = VSrc{prev + (size_t)r * H, 0}; ph.h_in = m->pf_h + (size_t)r * H; ph.h_out = nullptr; ph.norm_w = m->final_norm;
      ph.eps = c.rms_eps; ph.H = H; ph.act = act;
      GemvOut o{};
      o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
      BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
      BZ_TRY(emit_logits(m, logits_out, all ? srow : 0));
    }
  }
  return BZ_OK;
}

// op-level: y[S,N] = x16[S,K] . W[N,K]^T on the matrix cores, x rounded to the weight dtype first, f32 accumulators returned unrounded
extern "C" int bz_prefill_matmul(bz_model* m, const char* name, const bz_tensor* x, int S, bz_tensor* y) {
  BZ_API_BEGIN
  LinearDev L;
  BZ_TRY(find_linear(m, name, &L));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  const bool q4 = L.kind == LK_Q4G;   // int4 group-quantised weights x f16 activations (the AWQ / GPTQ prefill GEMM), S >= 9
  if (q4 ? !bzk_gemm_q4g_mfma_ok(L, BZ_F16, S) : (L.kind != LK_ROWS || (L.wdt != BZ_F16 && L.wdt != BZ_BF16)))
    BZ_FAIL(BZ_E_UNSUPPORTED, "prefill_matmul: '%s' is neither a dense f16 / bf16 weight nor an int4 weight without act-order (S >= 9)", name);
  if (!x || !y || x->dtype != BZ_F32 || y->dtype != BZ_F32 || S <= 0 || x->nbytes < (size_t)S * L.K * 4 || y->nbytes < (size_t)S * L.N * 4)
    BZ_FAIL(BZ_E_INVALID, "prefill_matmul: x must be F32 [S,%d], y F32 [S,%d]", L.K, L.N);
  BZ_HIP(hipSetDevice(m->dev->id));
  hipStream_t st = m->dev->stream;
  void* x16 = nullptr; void* ws = nullptr;
  const size_t ws_bytes = (size_t)48 << 20;      // split-K partials, as the prefill paths pass them
  BZ_HIP(hipMalloc(&x16, (size_t)S * L.K * 2));
  if (hipMalloc(&ws, ws_bytes) != hipSuccess) { hipFree(x16); BZ_FAIL(BZ_E_OOM, "prefill_matmul: workspace"); }
  const int xdt = q4 ? BZ_F16 : L.wdt;
  int rc = bzk_pf_cvt16(st, xdt, (const float*)x->ptr, (size_t)S * L.K, x16);
  if (rc == BZ_OK) rc = q4 ? bzk_gemm_q4g_mfma(st, L, x16, S, BZ_F32, (float*)y->ptr, (float*)ws, ws_bytes)
                           : bzk_gemm_nt(st, L.wdt, x16, L.w, L.bias, S, L.N, L.K, BZ_F32, (float*)y->ptr, (float*)ws, ws_bytes);
  hipStreamSynchronize(st);
  hipFree(x16); hipFree(ws);
  return rc;
  BZ_API_END
}

extern "C" int bz_forward_kv(bz_model* m, const bz_tensor* tokens, int S, bz_kv* kv, int position, bz_tensor* logits_out, uint32_t flags) {
  BZ_API_BEGIN
  BZ_TRY(check_fwd(m, tokens, S));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  BZ_TRACE("forward_kv: S=%d position=%d", S, position);
  if (m->cfg.arch == BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_INVALID, "forward_kv: model has no KV cache (use bz_forward_ssm)");
  if (!kv || kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "forward_kv: cache does not match the model");
  if (position < 0 || position + S > m->cfg.max_seq_len) BZ_FAIL(BZ_E_INVALID, "forward_kv: position %d + S %d exceeds max_seq_len %d", position, S, m->cfg.max_seq_len);
  BZ_TRY(kv_grow(kv, position + S));
  const bool all = flags & BZ_FWD_ALL_LOGITS;
  if (prefill_eligible(m, S, position + S)) {
    // prompt-sized inputs of dense 16-bit models: batched prefill, GEMMs on the matrix cores
    if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(all ? S : 1) * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
    BZ_TRY(prefill_dense(m, (const long long*)tokens->ptr, S, view_of(kv), position, nullptr, all, logits_out));
    kv->seq_len = position + S;
    return BZ_OK;
  }
  if (dsv2_prefill_eligible(m, S, position + S, view_of(kv))) {
    if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(all ? S : 1) * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
    BZ_TRY(dsv2_prefill(m, (const long long*)tokens->ptr, S, view_of(kv), position, all, logits_out));
    kv->seq_len = position + S;
    return BZ_OK;
  }
  for (int s = 0; s < S; s++) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, m->dev->stream, m->pos_tmp, position + s);
    StepIO io{};
    io.kv = view_of(kv); io.d_tok = (const long long*)tokens->ptr + s; io.d_pos = m->pos_tmp;
    io.att_positions = att_positions_for(position + s + 1);
    io.do_head = all || s == S - 1;
    BZ_TRY(model_step(m, io));
    if (io.do_head) BZ_TRY(emit_logits(m, logits_out, all ? s : 0));
  }
  kv->seq_len = position + S;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_forward_paged(bz_model* m, const bz_tensor* tokens, int S, bz_paged_kv* kv, const bz_tensor* slot_mapping,
                                const bz_tensor* block_table, int n_table, int seq_len_k, int start_pos, bz_tensor* logits_out, uint32_t flags) {
  BZ_API_BEGIN
  BZ_TRY(check_fwd(m, tokens, S));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch == BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_UNSUPPORTED, "forward_paged: Mamba2 has no KV cache");   // MLA: the latent cache pages like any other (one 'head' of rank + rope values)
  if (!kv || kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "forward_paged: cache does not match the model");
  if (!slot_mapping || slot_mapping->dtype != BZ_I32 || slot_mapping->nbytes < (size_t)S * 4) BZ_FAIL(BZ_E_INVALID, "forward_paged: slot_mapping must be I32[S]");
  if (!block_table || block_table->dtype != BZ_I32 || block_table->nbytes < (size_t)n_table * 4) BZ_FAIL(BZ_E_INVALID, "forward_paged: block_table must be I32[n_table]");
  if (start_pos < 0 || start_pos + S != seq_len_k) BZ_FAIL(BZ_E_INVALID, "forward_paged: start_pos + S must equal seq_len_k");
  if ((seq_len_k + kv->block_size - 1) / kv->block_size > n_table) BZ_FAIL(BZ_E_INVALID, "forward_paged: block_table too short for seq_len_k");
  if (seq_len_k > m->cfg.max_seq_len) BZ_FAIL(BZ_E_INVALID, "forward_paged: seq_len_k exceeds max_seq_len");
  const bool all = flags & BZ_FWD_ALL_LOGITS;
  if (prefill_eligible(m, S, seq_len_k)) {
    if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(all ? S : 1) * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
    BZ_TRY(prefill_dense(m, (const long long*)tokens->ptr, S, view_of(kv, (const int*)block_table->ptr, nullptr), start_pos, (const int*)slot_mapping->ptr, all, logits_out));
    kv->seq_len = seq_len_k;
    return BZ_OK;
  }
  if (dsv2_prefill_eligible(m, S, seq_len_k, view_of(kv, (const int*)block_table->ptr, nullptr))) {
    // (the rows' slots follow from the block table: slot = block_table[p / bs] * bs + p % bs, batch_decode.rs:81-88)
    if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(all ? S : 1) * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
    BZ_TRY(dsv2_prefill(m, (const long long*)tokens->ptr, S, view_of(kv, (const int*)block_table->ptr, nullptr), start_pos, all, logits_out));
    kv->seq_len = seq_len_k;
    return BZ_OK;
  }
  for (int s = 0; s < S; s++) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, m->dev->stream, m->pos_tmp, start_pos + s);
    StepIO io{};
    io.kv = view_of(kv, (const int*)block_table->ptr, (const int*)slot_mapping->ptr + s);
    io.d_tok = (const long long*)tokens->ptr + s; io.d_pos = m->pos_tmp;
    io.att_positions = att_positions_for(start_pos + s + 1);
    io.do_head = all || s == S - 1;
    BZ_TRY(model_step(m, io));
    if (io.do_head) BZ_TRY(emit_logits(m, logits_out, all ? s : 0));
  }
  kv->seq_len = seq_len_k;
  return BZ_OK;
  BZ_API_END
}

// Batched single-token decode over one paged cache (process_decode_batch, /root/reference/src/engine/batch_decode.rs:35-150): tokens [N,1],
// slot_mapping [N], block_table [N, max_blocks] (rows padded with 0), one length per sequence.  Logits [N, vocab].
// Models the multi-row pipeline covers (int4 without act-order, dense 16-bit) share the weights across the batch: one pass per 8 sequences;
// the others run the sequences one after another through the single-stream step.
extern "C" int bz_forward_paged_batch(bz_model* m, const bz_tensor* tokens, int N, bz_paged_kv* kv, const bz_tensor* slot_mapping, const bz_tensor* block_table,
                                      int max_blocks, const int32_t* seq_lens, bz_tensor* logits_out) {
  BZ_API_BEGIN
  BZ_TRY(check_fwd(m, tokens, N));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch != BZ_ARCH_LLAMA) BZ_FAIL(BZ_E_UNSUPPORTED, "forward_paged_batch: llama family only");
  if (!kv || kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: cache does not match the model");
  if (!slot_mapping || slot_mapping->dtype != BZ_I32 || slot_mapping->nbytes < (size_t)N * 4) BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: slot_mapping must be I32[N]");
  if (max_blocks <= 0 || !block_table || block_table->dtype != BZ_I32 || block_table->nbytes < (size_t)N * max_blocks * 4)
    BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: block_table must be I32[N, max_blocks]");
  if (!seq_lens) BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: seq_lens is null");
  if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)N * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: logits_out must be F32 [N, vocab]");
  int maxlen = 0;
  for (int i = 0; i < N; i++) {
    if (seq_lens[i] <= 0 || seq_lens[i] > m->cfg.max_seq_len || (seq_lens[i] + kv->block_size - 1) / kv->block_size > max_blocks)
      BZ_FAIL(BZ_E_INVALID, "forward_paged_batch: sequence %d has length %d (max_seq_len %d, %d blocks of %d)", i, seq_lens[i], m->cfg.max_seq_len, max_blocks, kv->block_size);
    maxlen = std::max(maxlen, seq_lens[i]);
  }
  static const bool no_share = getenv("BZ_NO_BATCH_SHARING") != nullptr;
  if (!no_share && N >= 2 && prefill_eligible(m, std::max(N, prefill_min_rows()), maxlen, true)) {
    // weight-sharing path: the N rows go through the multi-row pipeline (int4: one pass over the weights per 8 sequences; dense: MFMA GEMM),
    // each row with its own position, slot and block-table row
    std::vector<int> pos(N);
    for (int i = 0; i < N; i++) pos[i] = seq_lens[i] - 1;
    if (!m->row_pos || m->row_pos_n < N) {
      BZ_HIP(hipStreamSynchronize(m->dev->stream));
      void* p; BZ_TRY(dev_alloc(m, &p, (size_t)N * 4)); m->row_pos = (int*)p; m->row_pos_n = N;
    }
    BZ_HIP(hipMemcpyAsync(m->row_pos, pos.data(), (size_t)N * 4, hipMemcpyHostToDevice, m->dev->stream));
    BZ_HIP(hipStreamSynchronize(m->dev->stream));       // `pos` is a stack-lifetime host buffer
    RowsCtx rc; rc.row_pos = m->row_pos; rc.table_stride = max_blocks; rc.max_len = maxlen;
    BZ_TRY(prefill_dense(m, (const long long*)tokens->ptr, N, view_of(kv, (const int*)block_table->ptr, nullptr), 0, (const int*)slot_mapping->ptr, true, logits_out, rc));
    if (kv->seq_len < maxlen) kv->seq_len = maxlen;
    return BZ_OK;
  }
  for (int i = 0; i < N; i++) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, m->dev->stream, m->pos_tmp, seq_lens[i] - 1);   // the new token's position (batch_decode.rs:79-88)
    StepIO io{};
    io.kv = view_of(kv, (const int*)block_table->ptr + (size_t)i * max_blocks, (const int*)slot_mapping->ptr + i);
    io.d_tok = (const long long*)tokens->ptr + i; io.d_pos = m->pos_tmp;
    io.att_positions = att_positions_for(seq_lens[i]);
    BZ_TRY(model_step(m, io));
    BZ_TRY(emit_logits(m, logits_out, i));
  }
  if (kv->seq_len < maxlen) kv->seq_len = maxlen;
  return BZ_OK;
  BZ_API_END
}

static int check_ssm(bz_model* m, bz_ssm_state* st) {
  const bz_model_config& c = m->cfg;
  if (c.arch != BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_INVALID, "forward_ssm: model is not a mamba2 model");
  if (!st || st->layers != c.n_layers || st->n_heads != c.ssm_n_heads || st->head_dim != c.ssm_head_dim || st->d_state != c.ssm_d_state ||
      st->kc != c.ssm_conv_kernel || st->conv_dim != c.ssm_d_inner + 2 * c.ssm_n_groups * c.ssm_d_state || st->dtype != c.act_dtype)
    BZ_FAIL(BZ_E_INVALID, "forward_ssm: state does not match the model");
  return BZ_OK;
}

// Mamba2 batched prefill (prompts of >= prefill_min_rows() tokens, dense 16-bit in_proj / out_proj in the activation dtype): rows through the
// MFMA GEMMs, conv as a map over (token, channel), the recurrence as an in-kernel scan per head (bz_prefill.hip).  Same rounding points as
// mamba_step, token for token.
static bool mamba_prefill_eligible(const bz_model* m, int S) {
  static const bool off = getenv("BZ_NO_MFMA_PREFILL") != nullptr;
  const bz_model_config& c = m->cfg;
  if (off || c.arch != BZ_ARCH_MAMBA2 || S < prefill_min_rows()) return false;
  if (c.act_dtype != BZ_F16 && c.act_dtype != BZ_BF16) return false;
  {   // short prompts stay on the step kernels (exact sums: the oracle's bits), as for the dense Llama models (prefill_eligible)
    static const int env = getenv("BZ_EXACT_PREFILL") ? atoi(getenv("BZ_EXACT_PREFILL")) : -1;
    static const int max_rows = getenv("BZ_EXACT_PREFILL_MAX") ? atoi(getenv("BZ_EXACT_PREFILL_MAX")) : 16;
    if (env != 0 && (env > 0 || S <= max_rows)) return false;
  }
  if (!bzk_ssm_scan_ok(c.ssm_head_dim, c.ssm_d_state, c.ssm_n_groups, c.ssm_conv_kernel) || c.ssm_d_inner % c.ssm_n_groups || c.ssm_n_heads % c.ssm_n_groups) return false;
  for (const MambaLayerDev& L : m->mlayers)
    for (const FusedLinear* F : {&L.in_proj, &L.out_proj}) {
      if (F->parts.size() != 1) return false;
      const LinearDev& P = F->parts[0];
      if (P.kind != LK_ROWS || P.wdt != c.act_dtype || P.K % 64) return false;
    }
  return m->lm_head.parts.size() == 1 && m->lm_head.parts[0].kind == LK_ROWS && !m->lm_head.fix_out;
}
static int mamba_prefill(bz_model* m, const long long* d_tok, int S, bz_ssm_state* state, bool all, bz_tensor* logits_out) {
  const bz_model_config& c = m->cfg;
  hipStream_t st = m->dev->stream;
  const int D = c.hidden, DI = c.ssm_d_inner, NH = c.ssm_n_heads, NS = c.ssm_d_state, G = c.ssm_n_groups, KC = c.ssm_conv_kernel, act = c.act_dtype, dt = c.act_dtype;
  const int conv_dim = DI + 2 * G * NS, ld = DI + conv_dim + NH;
  const int CH = 512;
  const int rows = std::min(S, CH);
  BZ_TRY(ensure_pf_ws(m));
  if (m->mpf_rows < rows) {
    BZ_HIP(hipStreamSynchronize(st));
    void* p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * D * 4)); m->mpf_h = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * D * 4)); m->mpf_t = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * ld * 4)); m->mpf_zx = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * conv_dim * 4)); m->mpf_xbc = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * DI * 4)); m->mpf_y = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * NH * bzk_ssm_scan_pieces(c.ssm_head_dim) * 4)); m->mpf_vss = (float*)p;
    BZ_TRY(dev_alloc(m, &p, (size_t)rows * std::max(D, DI) * 2)); m->mpf_x16 = p;
    m->mpf_rows = rows;
  }
  for (int s0 = 0; s0 < S; s0 += CH) {
    const int n = std::min(CH, S - s0);
    BZ_TRY(bzk_pf_embed(st, m->embed, m->embed_dt, d_tok + s0, n, D, act, m->mpf_h));
    const float* prev = nullptr;
    for (int l = 0; l < c.n_layers; l++) {
      const MambaLayerDev& L = m->mlayers[l];
      const LinearDev& Pin = L.in_proj.parts[0]; const LinearDev& Pout = L.out_proj.parts[0];
      BZ_TRY(bzk_pf_norm(st, dt, m->mpf_h, prev, L.norm, n, D, c.rms_eps, act, m->mpf_x16));
      BZ_TRY(bzk_gemm_nt(st, act, m->mpf_x16, Pin.w, Pin.bias, n, Pin.N, Pin.K, act, m->mpf_zx, m->pf_ws, m->pf_ws_bytes));
      BZ_TRY(bzk_pf_conv(st, m->mpf_zx, ld, DI, conv_dim, KC, L.conv_w, L.conv_b, state->conv + (size_t)l * conv_dim * (KC - 1), n, act, m->mpf_xbc));
      BzSsmScan sc{};
      sc.xbc = m->mpf_xbc; sc.conv_dim = conv_dim; sc.zx = m->mpf_zx; sc.ld = ld; sc.dt_off = DI + conv_dim; sc.dt_bias = L.dt_bias; sc.A_log = L.A_log; sc.D = L.D;
      sc.state = (char*)state->ssm + (size_t)l * NH * c.ssm_head_dim * NS * bz_dtype_size(state->dtype);
      sc.n_heads = NH; sc.head_dim = c.ssm_head_dim; sc.d_state = NS; sc.n_groups = G; sc.d_inner = DI; sc.act = act; sc.S = n; sc.y = m->mpf_y; sc.vss = m->mpf_vss;
      BZ_TRY(bzk_ssm_scan(st, sc, state->dtype));
      BZ_TRY(bzk_pf_gnorm(st, dt, m->mpf_y, m->mpf_vss, L.gnorm, n, DI, G, NH * bzk_ssm_scan_pieces(c.ssm_head_dim), c.rms_eps, act, m->mpf_x16));
      BZ_TRY(bzk_gemm_nt(st, act, m->mpf_x16, Pout.w, Pout.bias, n, Pout.N, Pout.K, act, m->mpf_t, m->pf_ws, m->pf_ws_bytes));
      prev = m->mpf_t;
    }
    for (int r = 0; r < n; r++) {
      const int srow = s0 + r;
      if (!all && srow != S - 1) continue;
      Pro ph{}; ph.mode = PRO_NORM; ph.src = VSrc{prev + (size_t)r * D, 0}; ph.h_in = m->mpf_h + (size_t)r * D; ph.h_out = nullptr; ph.norm_w = m->final_norm;
      ph.eps = c.rms_eps; ph.H = D; ph.act = act;
      GemvOut o{};
      o.direct = m->logits; o.amax_val = m->pval; o.amax_idx = m->pidx;
      BZ_TRY(bzk_gemv(st, m->lm_head.parts[0], ph, o, act));
      BZ_TRY(emit_logits(m, logits_out, all ? srow : 0));
    }
  }
  return BZ_OK;
}

extern "C" int bz_forward_ssm(bz_model* m, const bz_tensor* tokens, int S, bz_ssm_state* st, bz_tensor* logits_out, uint32_t flags) {
  BZ_API_BEGIN
  BZ_TRY(check_fwd(m, tokens, S));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  BZ_TRY(check_ssm(m, st));
  const bool all = flags & BZ_FWD_ALL_LOGITS;
  if (mamba_prefill_eligible(m, S)) {
    if (!logits_out || logits_out->dtype != BZ_F32 || logits_out->nbytes < (size_t)(all ? S : 1) * m->cfg.vocab * 4) BZ_FAIL(BZ_E_INVALID, "forward: logits_out too small");
    BZ_HIP(hipSetDevice(m->dev->id));
    return mamba_prefill(m, (const long long*)tokens->ptr, S, st, all, logits_out);
  }
  for (int s = 0; s < S; s++) {
    StepIO io{};
    io.ssm = st; io.d_tok = (const long long*)tokens->ptr + s;
    io.do_head = all || s == S - 1;
    BZ_TRY(mamba_step(m, io));
    if (io.do_head) BZ_TRY(emit_logits(m, logits_out, all ? s : 0));
  }
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_forward_embed(bz_model* m, const bz_tensor* tokens, int S, bz_tensor* hidden_out) {
  BZ_API_BEGIN
  BZ_TRY(check_fwd(m, tokens, S));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  const int H = m->cfg.hidden;
  if (!hidden_out || hidden_out->dtype != BZ_F32 || hidden_out->nbytes < (size_t)S * H * 4) BZ_FAIL(BZ_E_INVALID, "forward_embed: hidden_out must be F32 [S,hidden]");
  for (int s = 0; s < S; s++)
    BZ_TRY(bzk_embed(m->dev->stream, m->embed, m->embed_dt, (const long long*)tokens->ptr + s, H, m->cfg.act_dtype, (float*)hidden_out->ptr + (size_t)s * H));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_forward_layers_range(bz_model* m, bz_tensor* hidden, bz_tensor* prev_mlp, int* has_prev, int S, bz_kv* kv, int start, int end, int position) {
  BZ_API_BEGIN
  if (!m || !m->finalized) BZ_FAIL(BZ_E_INVALID, "model not finalized");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch != BZ_ARCH_LLAMA) BZ_FAIL(BZ_E_UNSUPPORTED, "layers_range: llama family only");
  const int H = m->cfg.hidden;
  if (!hidden || hidden->dtype != BZ_F32 || hidden->nbytes < (size_t)S * H * 4 || !prev_mlp || prev_mlp->dtype != BZ_F32 || prev_mlp->nbytes < (size_t)S * H * 4 || !has_prev)
    BZ_FAIL(BZ_E_INVALID, "layers_range: hidden / prev_mlp must be F32 [S,hidden]");
  if (start < 0 || end > m->cfg.n_layers || start > end) BZ_FAIL(BZ_E_INVALID, "layers_range: bad layer range [%d,%d)", start, end);
  if (!kv || position < 0 || position + S > m->cfg.max_seq_len) BZ_FAIL(BZ_E_INVALID, "layers_range: bad cache / position");
  BZ_HIP(hipSetDevice(m->dev->id));
  BZ_TRY(kv_grow(kv, position + S));
  if (start == end) return BZ_OK;
  for (int s = 0; s < S; s++) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, m->dev->stream, m->pos_tmp, position + s);
    StepIO io{};
    io.kv = view_of(kv); io.d_tok = nullptr; io.d_pos = m->pos_tmp;
    io.att_positions = att_positions_for(position + s + 1);
    io.do_embed = false; io.do_head = false; io.layer_start = start; io.layer_end = end;
    io.hidden_in = (float*)hidden->ptr + (size_t)s * H;
    io.prev_in = *has_prev ? (float*)prev_mlp->ptr + (size_t)s * H : nullptr;
    io.hidden_out = (float*)hidden->ptr + (size_t)s * H;
    io.prev_out = (float*)prev_mlp->ptr + (size_t)s * H;
    BZ_TRY(llama_step(m, io));
  }
  *has_prev = 1;
  if (end == m->cfg.n_layers) kv->seq_len = position + S;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_forward_head(bz_model* m, const bz_tensor* hidden, const bz_tensor* prev_mlp, int has_prev, int S, bz_tensor* logits_out, uint32_t flags) {
  BZ_API_BEGIN
  if (!m || !m->finalized) BZ_FAIL(BZ_E_INVALID, "model not finalized");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  const int H = m->cfg.hidden;
  if (!hidden || hidden->dtype != BZ_F32 || hidden->nbytes < (size_t)S * H * 4) BZ_FAIL(BZ_E_INVALID, "forward_head: hidden must be F32 [S,hidden]");
  if (has_prev && (!prev_mlp || prev_mlp->nbytes < (size_t)S * H * 4)) BZ_FAIL(BZ_E_INVALID, "forward_head: prev_mlp must be F32 [S,hidden]");
  BZ_HIP(hipSetDevice(m->dev->id));
  const bool all = flags & BZ_FWD_ALL_LOGITS;
  for (int s = all ? 0 : S - 1; s < S; s++) {
    StepIO io{};
    io.do_embed = false; io.do_head = true; io.layer_start = 0; io.layer_end = 0;
    io.hidden_in = (const float*)hidden->ptr + (size_t)s * H;
    io.prev_in = has_prev ? (const float*)prev_mlp->ptr + (size_t)s * H : nullptr;
    BZ_TRY(llama_step(m, io));
    BZ_TRY(emit_logits(m, logits_out, all ? s : 0));
  }
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// measurement: per-kernel dispatch times of real decode steps (rocprof-comparable kernel durations
// taken with HIP events on the launch stream)
// ---------------------------------------------------------------------------------------------------------
static int profile_collect(BzTimingSink& sink, int rc, bz_kernel_time* out, int max_out, int* n_out) {
  int n = 0;
  for (auto& r : sink.recs) {
    float ms = 0.f;
    if (rc == BZ_OK && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      int j = 0;
      for (; j < n; j++) if (strcmp(out[j].name, r.label) == 0) break;
      if (j == n && n < max_out) { memset(&out[n], 0, sizeof(out[n])); strncpy(out[n].name, r.label, sizeof(out[n].name) - 1); n++; }
      if (j < n) { out[j].launches++; out[j].total_ms += ms; out[j].algo_bytes += r.bytes; }
    }
    hipEventDestroy(r.e0); hipEventDestroy(r.e1);
  }
  *n_out = n;
  return rc;
}

extern "C" int bz_profile_step(bz_model* m, bz_kv* kv, int64_t token, int position, int iters, bz_kernel_time* out, int max_out, int* n_out) {
  BZ_API_BEGIN
  if (m && m->finalized && m->cfg.arch == BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_UNSUPPORTED, "profile_step: model has no KV cache (use bz_profile_step_ssm)");
  if (!m || !m->finalized || !kv || !out || !n_out || iters <= 0 || max_out <= 0) BZ_FAIL(BZ_E_INVALID, "profile_step: bad argument");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (token < 0 || token >= m->cfg.vocab || position < 0 || position + iters > m->cfg.max_seq_len) BZ_FAIL(BZ_E_INVALID, "profile_step: token/position out of range");
  BZ_HIP(hipSetDevice(m->dev->id));
  BZ_TRY(kv_grow(kv, position + iters));
  hipStream_t st = m->dev->stream;
  long long t = token;
  BZ_HIP(hipMemcpyAsync(m->tok_tmp, &t, 8, hipMemcpyHostToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  BzTimingSink sink;
  int rc = BZ_OK;
  for (int i = 0; i < iters && rc == BZ_OK; i++) {
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, m->pos_tmp, position + i);
    FinalArgs fa{};
    fa.tok_out = m->tok_tmp + 1;   // scratch: do not feed the sampled token back (same input every iteration)
    StepIO io{};
    io.kv = view_of(kv); io.d_tok = m->tok_tmp; io.d_pos = m->pos_tmp; io.final_args = &fa;
    io.att_positions = att_positions_for(position + i + 1);
    bzk_set_timing_sink(&sink);
    rc = model_step(m, io);
    bzk_set_timing_sink(nullptr);
  }
  hipStreamSynchronize(st);
  if (kv->seq_len < position + iters) kv->seq_len = position + iters;
  return profile_collect(sink, rc, out, max_out, n_out);
  BZ_API_END
}

extern "C" int bz_profile_step_ssm(bz_model* m, bz_ssm_state* ssm, int64_t token, int iters, bz_kernel_time* out, int max_out, int* n_out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !out || !n_out || iters <= 0 || max_out <= 0) BZ_FAIL(BZ_E_INVALID, "profile_step_ssm: bad argument");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  BZ_TRY(check_ssm(m, ssm));
  if (token < 0 || token >= m->cfg.vocab) BZ_FAIL(BZ_E_INVALID, "profile_step_ssm: token out of range");
  BZ_HIP(hipSetDevice(m->dev->id));
  hipStream_t st = m->dev->stream;
  long long t = token;
  BZ_HIP(hipMemcpyAsync(m->tok_tmp, &t, 8, hipMemcpyHostToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  BzTimingSink sink;
  int rc = BZ_OK;
  for (int i = 0; i < iters && rc == BZ_OK; i++) {
    FinalArgs fa{};
    fa.tok_out = m->tok_tmp + 1;
    StepIO io{};
    io.ssm = ssm; io.d_tok = m->tok_tmp; io.final_args = &fa;
    bzk_set_timing_sink(&sink);
    rc = mamba_step(m, io);
    bzk_set_timing_sink(nullptr);
  }
  hipStreamSynchronize(st);
  return profile_collect(sink, rc, out, max_out, n_out);
  BZ_API_END
}

__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}

// Kernel tuning aid: times the int4 GEMV kernel alone on synthetic weights ([N,K], gs 128) rotated over `nbuf` buffers
// (so that every launch streams from HBM, not from the 256 MiB Infinity Cache).  mode: 0 plain f32 x, 1 fused
// residual+RMSNorm prologue (fixed-point prev), 2 SiLU*up prologue (fixed-point gate/up).  Returns the mean dispatch time.
extern "C" int bz_tune_gemv(bz_device* dev, int N, int K, int gw, int mode, int nbuf, int iters, int flags, double* avg_us) {
  BZ_API_BEGIN
  if (!dev || !avg_us || N % 64 || K % 128 || gw <= 0 || (K / 128) % gw || gw > 16 || nbuf <= 0 || iters <= 0 || mode < 0 || mode > 2) BZ_FAIL(BZ_E_INVALID, "tune_gemv: bad argument");
  BZ_HIP(hipSetDevice(dev->id));
  hipStream_t st = dev->stream;
  const size_t G = (size_t)K / 128, wb = (size_t)N * K / 2, sb = (size_t)N * G * 2, zb = (size_t)N * G;
  std::vector<void*> bufs;
  int rc = BZ_OK;
  auto alloc = [&](size_t bytes, void** p) { if (hipMalloc(p, bytes) != hipSuccess) { rc = BZ_E_OOM; *p = nullptr; } else bufs.push_back(*p); };
  std::vector<LinearDev> Ls(nbuf);
  for (int b = 0; b < nbuf && rc == BZ_OK; b++) {
    void *w, *s2, *z;
    alloc(wb, &w); alloc(sb, &s2); alloc(zb, &z);
    if (rc != BZ_OK) break;
    hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, st, (uint32_t*)w, wb / 4, 17u * b + 1u);
    hipMemsetAsync(s2, 0x1c, sb, st);   // f16 0x1c1c ~ 0.004
    hipMemsetAsync(z, 8, zb, st);
    LinearDev& L = Ls[b];
    L.kind = LK_Q4G; L.N = N; L.K = K; L.gs = 128; L.w = w; L.scales = s2; L.zeros = z; L.gw = gw; L.npf = (flags & 8) ? 4 : 2; L.algo_bytes = wb + sb + zb / 2;
  }
  const int KX = mode == 2 ? 2 * K : K;
  void *xs, *acc, *hin, *hout, *nw, *src;
  alloc((size_t)KX * 8, &src); alloc((size_t)N * 8, &acc); alloc((size_t)K * 4, &hin); alloc((size_t)K * 4, &hout); alloc((size_t)K * 4, &nw); alloc((size_t)KX * 4, &xs);
  if (rc != BZ_OK) { for (void* p : bufs) hipFree(p); BZ_FAIL(BZ_E_OOM, "tune_gemv: out of memory"); }
  hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, st, (uint32_t*)src, (size_t)KX * 2, 5u);
  hipMemsetAsync(xs, 0x3c, (size_t)KX * 4, st); hipMemsetAsync(hin, 0x3c, (size_t)K * 4, st); hipMemsetAsync(nw, 0x3c, (size_t)K * 4, st);
  hipMemsetAsync(acc, 0, (size_t)N * 8, st);
  Pro p{};
  p.act = BZ_F16; p.dbg = flags; p.eps = 1e-5f;
  if (mode == 0) { p.mode = PRO_PLAIN; p.src = VSrc{xs, 0}; }
  else if (mode == 1) { p.mode = PRO_NORM; p.src = VSrc{src, 1}; p.h_in = (float*)hin; p.h_out = (float*)hout; p.norm_w = (float*)nw; p.H = K; }
  else { p.mode = PRO_SILU; p.src = VSrc{src, 1}; p.H = K; }
  BzTimingSink sink;
  long long* stp = nullptr;
  if (flags & 16) { p.dbg = 0; if (hipMalloc((void**)&stp, 2 * 12 * 8 * 8) == hipSuccess) { hipMemsetAsync(stp, 0, 2 * 12 * 8 * 8, st); p.stamps = stp; bufs.push_back(stp); } }
  for (int i = 0; i < iters + 2 && rc == BZ_OK; i++) {
    GemvOut o{}; o.acc = (long long*)acc;
    if (i == 2) bzk_set_timing_sink(&sink);
    rc = bzk_gemv(st, Ls[i % nbuf], p, o, BZ_F16);
  }
  bzk_set_timing_sink(nullptr);
  hipStreamSynchronize(st);
  if (stp) {   // diagnostic: per-wave phase stamps of the slim kernel's workgroups 0 and 97 (10 ns units, relative to the workgroup's first)
    long long h[192]; hipMemcpy(h, stp, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; b++) {
      long long t0 = INT64_MAX;
      for (int i = 0; i < 96; i++) if (h[b * 96 + i] > 0) t0 = std::min(t0, h[b * 96 + i]);
      fprintf(stderr, "[bz] slim stamps, workgroup %s (us): rows = waves 0..11 (0-3 row waves, 4-11 tile waves); entry, loads issued, ssd, rs known, planes published / seen, group 0 done, atomics issued, drained\n", b ? "97" : "0");
      for (int w = 0; w < 12; w++) { fprintf(stderr, "   "); for (int i = 0; i < 8; i++) fprintf(stderr, " %6.2f", h[(b * 12 + w) * 8 + i] > 0 ? (h[(b * 12 + w) * 8 + i] - t0) / 100.0 : -1.0); fprintf(stderr, "\n"); }
    }
  }
  double tot = 0; int n = 0;
  for (auto& r : sink.recs) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { tot += ms; n++; } hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  *avg_us = n ? 1e3 * tot / n : 0.0;
  for (void* q : bufs) hipFree(q);
  return rc;
  BZ_API_END
}

// Tuning aid for the dense row GEMV (k_gemv_rows): [N,K] 16-bit weights in `nbuf` rotating buffers (cold HBM), the real launcher.
// mode: 0 plain f32 x, 1 fused residual + RMSNorm prologue (fixed-point prev), 2 SiLU*up prologue (fixed-point gate/up); sk: split-K count
// (0 = the loader's choice, bzk_rows_choose_sk).  Returns the mean dispatch time.
extern "C" int bz_tune_rows(bz_device* dev, int N, int K, int wdt, int mode, int sk, int nbuf, int iters, double* avg_us) {
  BZ_API_BEGIN
  if (!dev || !avg_us || N <= 0 || K % 8 || K <= 0 || nbuf <= 0 || iters <= 0 || mode < 0 || mode > 2 || (wdt != BZ_F16 && wdt != BZ_BF16)) BZ_FAIL(BZ_E_INVALID, "tune_rows: bad argument");
  BZ_HIP(hipSetDevice(dev->id));
  hipStream_t st = dev->stream;
  const size_t wb = (size_t)N * K * 2;
  std::vector<void*> bufs;
  int rc = BZ_OK;
  auto alloc = [&](size_t bytes, void** p) { if (hipMalloc(p, bytes) != hipSuccess) { rc = BZ_E_OOM; *p = nullptr; } else bufs.push_back(*p); };
  std::vector<LinearDev> Ls(nbuf);
  const int SK = sk > 0 ? sk : bzk_rows_choose_sk(N, K);
  for (int b = 0; b < nbuf && rc == BZ_OK; b++) {
    void* w;
    alloc(wb, &w);
    if (rc != BZ_OK) break;
    hipMemsetAsync(w, wdt == BZ_F16 ? 0x1c : 0x3b, wb, st);   // small finite values in either format
    LinearDev& L = Ls[b];
    L.kind = LK_ROWS; L.N = N; L.K = K; L.wdt = wdt; L.w = w; L.sk = SK; L.algo_bytes = wb;
  }
  const int KX = mode == 2 ? 2 * K : K;
  void *xs, *acc, *hin, *hout, *nw, *src, *y;
  alloc((size_t)KX * 8, &src); alloc((size_t)N * 8, &acc); alloc((size_t)K * 4, &hin); alloc((size_t)K * 4, &hout); alloc((size_t)K * 4, &nw); alloc((size_t)KX * 4, &xs);
  alloc((size_t)N * 4, &y);
  if (rc != BZ_OK) { for (void* p : bufs) hipFree(p); BZ_FAIL(BZ_E_OOM, "tune_rows: out of memory"); }
  hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, st, (uint32_t*)src, (size_t)KX * 2, 5u);
  hipMemsetAsync(xs, 0x3c, (size_t)KX * 4, st); hipMemsetAsync(hin, 0x3c, (size_t)K * 4, st); hipMemsetAsync(nw, 0x3c, (size_t)K * 4, st);
  hipMemsetAsync(acc, 0, (size_t)N * 8, st);
  Pro p{};
  p.act = wdt; p.eps = 1e-5f;
  if (mode == 0) { p.mode = PRO_PLAIN; p.src = VSrc{xs, 0}; }
  else if (mode == 1) { p.mode = PRO_NORM; p.src = VSrc{src, 1}; p.h_in = (float*)hin; p.h_out = (float*)hout; p.norm_w = (float*)nw; p.H = K; }
  else { p.mode = PRO_SILU; p.src = VSrc{src, 1}; p.H = K; }
  BzTimingSink sink;
  for (int i = 0; i < iters + 2 && rc == BZ_OK; i++) {
    GemvOut o{};
    if (SK > 1) o.acc = (long long*)acc; else o.direct = (float*)y;
    if (i == 2) bzk_set_timing_sink(&sink);
    rc = bzk_gemv(st, Ls[i % nbuf], p, o, wdt);
  }
  bzk_set_timing_sink(nullptr);
  hipStreamSynchronize(st);
  double tot = 0; int n = 0;
  for (auto& r : sink.recs) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { tot += ms; n++; } hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  *avg_us = n ? 1e3 * tot / n : 0.0;
  for (void* q : bufs) hipFree(q);
  return rc;
  BZ_API_END
}

// Tuning aid for the fused MLP kernel alone: synthetic gate/up [2I,H] and down [H,I] int4 weights in `nbuf` rotating sets (cold HBM), the
// real launcher, mean dispatch time; stamps_out (optional, 2 x 16 x 16 values): the diagnostic build's per-wave s_memrealtime stamps
// (100 MHz) of workgroup 0 and workgroup 113, relative to the earliest stamp of each workgroup.
extern "C" int bz_tune_mlp(bz_device* dev, int H, int I, int nbuf, int iters, int flags, double* avg_us, long long* stamps_out) {
  BZ_API_BEGIN
  if (!dev || !avg_us || (H != 2048 && H != 4096) || I % 128 || nbuf <= 0 || iters <= 0) BZ_FAIL(BZ_E_INVALID, "tune_mlp: bad argument");
  BZ_HIP(hipSetDevice(dev->id));
  hipStream_t st = dev->stream;
  std::vector<void*> bufs;
  int rc = BZ_OK;
  auto alloc = [&](size_t bytes, void** p) { if (hipMalloc(p, bytes) != hipSuccess) { rc = BZ_E_OOM; *p = nullptr; } else bufs.push_back(*p); };
  auto mk = [&](int N, int K, LinearDev& L, unsigned seed) {
    const size_t G = (size_t)K / 128, wb = (size_t)N * K / 2, sb = (size_t)N * G * 2, zb = (size_t)N * G;
    void *w = nullptr, *s2 = nullptr, *z = nullptr;
    alloc(wb, &w); alloc(sb, &s2); alloc(zb, &z);
    if (rc != BZ_OK) return;
    hipLaunchKernelGGL(k_fill_u32, dim3(2048), dim3(256), 0, st, (uint32_t*)w, wb / 4, seed);
    hipMemsetAsync(s2, 0x1c, sb, st); hipMemsetAsync(z, 8, zb, st);
    L.kind = LK_Q4G; L.N = N; L.K = K; L.gs = 128; L.w = w; L.scales = s2; L.zeros = z; L.algo_bytes = wb + sb + zb / 2;
  };
  std::vector<LinearDev> GU(nbuf), DN(nbuf);
  for (int b = 0; b < nbuf && rc == BZ_OK; b++) { mk(2 * I, H, GU[b], 17u * b + 1u); mk(H, I, DN[b], 29u * b + 3u); }
  void *src = nullptr, *acc = nullptr, *hin = nullptr, *hout = nullptr, *nw = nullptr, *stp = nullptr;
  alloc((size_t)H * 8, &src); alloc((size_t)H * 8, &acc); alloc((size_t)H * 4, &hin); alloc((size_t)H * 4, &hout); alloc((size_t)H * 4, &nw); alloc(2 * 16 * 16 * 8, &stp);
  if (rc != BZ_OK) { for (void* p : bufs) hipFree(p); BZ_FAIL(BZ_E_OOM, "tune_mlp: out of memory"); }
  hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, st, (uint32_t*)src, (size_t)H * 2, 5u);
  hipMemsetAsync(hin, 0x3c, (size_t)H * 4, st); hipMemsetAsync(nw, 0x3c, (size_t)H * 4, st); hipMemsetAsync(acc, 0, (size_t)H * 8, st); hipMemsetAsync(stp, 0, 2 * 16 * 16 * 8, st);
  Pro p{};
  p.mode = PRO_NORM; p.act = BZ_F16; p.eps = 1e-5f; p.src = VSrc{src, 1}; p.h_in = (float*)hin; p.h_out = (float*)hout; p.norm_w = (float*)nw; p.H = H; p.dbg = flags;
  p.stamps = stamps_out ? (long long*)stp : nullptr;
  BzTimingSink sink;
  for (int i = 0; i < iters + 2 && rc == BZ_OK; i++) {
    if (i == 2) bzk_set_timing_sink(&sink);
    rc = bzk_mlp_q4g(st, GU[i % nbuf], DN[i % nbuf], H, I, p, (long long*)acc, nullptr, 0);
  }
  bzk_set_timing_sink(nullptr);
  hipStreamSynchronize(st);
  double tot = 0; int n = 0;
  for (auto& r : sink.recs) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { tot += ms; n++; } hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  *avg_us = n ? 1e3 * tot / n : 0.0;
  if (stamps_out) {
    hipMemcpy(stamps_out, stp, 2 * 16 * 16 * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; b++) {
      long long t0 = INT64_MAX;
      for (int i = 0; i < 256; i++) if (stamps_out[b * 256 + i] > 0) t0 = std::min(t0, stamps_out[b * 256 + i]);
      for (int i = 0; i < 256; i++) if (stamps_out[b * 256 + i] > 0) stamps_out[b * 256 + i] -= t0; else stamps_out[b * 256 + i] = -1;
    }
  }
  for (void* q : bufs) hipFree(q);
  return rc;
  BZ_API_END
}

// Measured HBM read ceiling of THIS device ("record the measured peak on the box and report against both"): a streaming read of
// `bytes` (spread over rotating buffers far beyond the 256 MiB Infinity Cache) with 16-byte non-temporal loads, 8 KiB in flight per wave,
// 256 x 512-thread workgroups -- the access pattern of the decode kernels without any arithmetic.  Best of `iters` launches, GB/s.
typedef unsigned int bz_u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_probe_read(const bz_u32x4_t* __restrict__ src, size_t per_wave_vec, unsigned* sink) {
  const int wave = (blockIdx.x * 512 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const bz_u32x4_t* p = src + (size_t)wave * per_wave_vec + lane;
  bz_u32x4_t acc = {0, 0, 0, 0};
  for (size_t i = 0; i < per_wave_vec; i += 64 * 8) {
    bz_u32x4_t v[8];
#pragma unroll
    for (int d = 0; d < 8; d++) v[d] = __builtin_nontemporal_load(p + i + 64 * d);
#pragma unroll
    for (int d = 0; d < 8; d++) acc ^= v[d];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
extern "C" int bz_probe_hbm_read(bz_device* dev, size_t bytes, int iters, double* gbs) {
  BZ_API_BEGIN
  if (!dev || !gbs || iters <= 0 || bytes < (64u << 20)) BZ_FAIL(BZ_E_INVALID, "probe_hbm_read: bad argument (>= 64 MiB)");
  BZ_HIP(hipSetDevice(dev->id));
  hipStream_t st = dev->stream;
  const int NBUF = 4, waves = 256 * 8;
  const size_t per_wave_vec = bytes / 16 / waves / 512 * 512;
  const size_t used = per_wave_vec * 16 * waves;
  void* bufs[NBUF] = {nullptr, nullptr, nullptr, nullptr}; unsigned* sink = nullptr;
  int rc = BZ_OK;
  for (int b = 0; b < NBUF; b++) if (hipMalloc(&bufs[b], used) != hipSuccess) { rc = BZ_E_OOM; bufs[b] = nullptr; break; } else hipMemsetAsync(bufs[b], b + 1, used, st);
  if (rc == BZ_OK && hipMalloc((void**)&sink, 64) != hipSuccess) rc = BZ_E_OOM;
  double best = 0.0;
  if (rc == BZ_OK) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < iters + 2; i++) {
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_probe_read, dim3(256), dim3(512), 0, st, (const bz_u32x4_t*)bufs[i % NBUF], per_wave_vec, sink);
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 2 && ms > 0.f) best = std::max(best, (double)used / 1e9 / (ms * 1e-3));
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  for (int b = 0; b < NBUF; b++) if (bufs[b]) hipFree(bufs[b]);
  if (sink) hipFree(sink);
  if (rc != BZ_OK) BZ_FAIL(rc, "probe_hbm_read: out of device memory");
  *gbs = best;
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// sampling
// ---------------------------------------------------------------------------------------------------------
extern "C" int bz_logits_to_token(bz_device* dev, const bz_tensor* logits, int64_t rows, int64_t vocab, const bz_tensor* ids, const bz_tensor* cnts,
                                  int n, float rp, float fp, float pp, float temperature, int top_k, float top_p, float min_p, uint64_t seed,
                                  bz_tensor* token_out) {
  BZ_API_BEGIN
  if (!dev || !logits || !token_out || rows <= 0 || vocab <= 0) BZ_FAIL(BZ_E_INVALID, "logits_to_token: bad argument");
  if (logits->dtype != BZ_F32 || logits->nbytes < (size_t)rows * vocab * 4) BZ_FAIL(BZ_E_INVALID, "logits_to_token: logits must be F32 [rows,vocab]");
  if (token_out->dtype != BZ_I64 || token_out->nbytes < 8) BZ_FAIL(BZ_E_INVALID, "logits_to_token: token_out must be I64[1]");
  if (n > 0 && (!ids || !cnts || ids->dtype != BZ_I64 || cnts->dtype != BZ_I32 || ids->nbytes < (size_t)n * 8 || cnts->nbytes < (size_t)n * 4))
    BZ_FAIL(BZ_E_INVALID, "logits_to_token: ids I64[n] / cnts I32[n] required");
  std::lock_guard<std::mutex> dlock__(dev->mu);
  BZ_HIP(hipSetDevice(dev->id));
  float* scratch = dev->scratch;
  if (temperature < 0.0f) BZ_FAIL(BZ_E_INVALID, "logits_to_token: negative temperature");
  if (temperature != 0.0f)   // generation.rs:262-264: greedy == temperature 0; otherwise the sampled path
    return bzk_sample(dev->stream, &dev->samp_ws, (const float*)logits->ptr + (size_t)(rows - 1) * vocab, vocab, n ? (const long long*)ids->ptr : nullptr,
                      n ? (const int*)cnts->ptr : nullptr, n, rp, fp, pp, temperature, top_k, top_p, min_p, seed, (long long*)token_out->ptr);
  int rc = bzk_logits_to_token(dev->stream, (const float*)logits->ptr + (size_t)(rows - 1) * vocab, vocab, n ? (const long long*)ids->ptr : nullptr,
                               n ? (const int*)cnts->ptr : nullptr, n, rp, fp, pp, temperature, top_k, top_p, min_p, seed, scratch,
                               (long long*)token_out->ptr);
  return rc;
  BZ_API_END
}
extern "C" int bz_argmax_to_buf(bz_device* dev, const bz_tensor* logits, int64_t rows, int64_t vocab, bz_tensor* token_out) {
  BZ_API_BEGIN
  return bz_logits_to_token(dev, logits, rows, vocab, nullptr, nullptr, 0, 1.0f, 0.f, 0.f, 0.f, 0, 1.f, 0.f, 0, token_out);
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// whole-step hipGraph
// ---------------------------------------------------------------------------------------------------------
struct bz_decode_graph {
  bz_model* m = nullptr;
  bz_device* dev = nullptr;
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  hipGraph_t graph_long = nullptr; hipGraphExec_t exec_long = nullptr;   // split-KV attention variant (captured on first need)
  KvView view{}; int capacity = 0;  // what the step was captured over; positions the long variant's grid covers
  int seed_pos = 0;                 // host copy of the seeded position (position of replay r = seed_pos + r)
  long long* tok_buf = nullptr;     // device: input token of the next replay
  int* pos = nullptr;               // device: position of the next replay
  int* step = nullptr;              // device: replay counter
  long long* tok_log = nullptr;     // pinned host, written by the final kernel: log[step % LOGCAP]
  int* block_table = nullptr; int max_blocks = 0;
  bz_kv* kv = nullptr; bz_paged_kv* pkv = nullptr; bz_ssm_state* ssm = nullptr;
  std::vector<hipEvent_t> evs;
  long long replays = 0;
  static const int LOGCAP = 4096;
};

// one capture of the decode step over the graph's device-resident token / position / step words; att_positions > 0 records the long-context
// (split-KV attention) form of the step
static int graph_capture_variant(bz_decode_graph* g, int att_positions, hipGraph_t* graph_out, hipGraphExec_t* exec_out) {
  bz_model* m = g->m;
  FinalArgs fa{};
  fa.tok_out = g->tok_buf; fa.tok_log = g->tok_log; fa.step = g->step; fa.logcap = bz_decode_graph::LOGCAP; fa.pos = g->pos;
  StepIO io{};
  io.kv = g->view; io.d_tok = g->tok_buf; io.d_pos = g->pos; io.final_args = &fa; io.ssm = g->ssm; io.att_positions = att_positions;
  BZ_TRACE("graph: begin capture");
  hipStream_t cap = nullptr;
  BZ_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  hipError_t eb = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (eb != hipSuccess) { hipStreamDestroy(cap); BZ_FAIL(BZ_E_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(eb)); }
  tl_capture_stream = cap;
  int rc = model_step(m, io);
  tl_capture_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(cap, &graph);
  hipStreamDestroy(cap);
  BZ_TRACE("graph: end capture rc=%d hip=%d", rc, (int)e);
  if (rc != BZ_OK) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) BZ_FAIL(BZ_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  *graph_out = graph;
  BZ_HIP(hipGraphInstantiate(exec_out, graph, nullptr, nullptr, 0));
  BZ_TRACE("graph: instantiated");
  return BZ_OK;
}

static int graph_capture_common(bz_decode_graph* g, const KvView& view) {
  bz_model* m = g->m;
  hipStream_t st = m->dev->stream;
  BZ_HIP(hipMalloc(&g->tok_buf, 64));
  BZ_HIP(hipMalloc(&g->pos, 64));
  BZ_HIP(hipMalloc(&g->step, 64));
  BZ_HIP(hipMemset(g->tok_buf, 0, 64)); BZ_HIP(hipMemset(g->pos, 0, 64)); BZ_HIP(hipMemset(g->step, 0, 64));
  BZ_HIP(hipHostMalloc(&g->tok_log, sizeof(long long) * bz_decode_graph::LOGCAP, hipHostMallocDefault));
  memset(g->tok_log, 0xff, sizeof(long long) * bz_decode_graph::LOGCAP);
  BZ_HIP(hipDeviceSynchronize());
  g->view = view;
  BZ_TRY(graph_capture_variant(g, 0, &g->graph, &g->exec));
  for (int i = 0; i < 8; i++) { hipEvent_t ev; BZ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); g->evs.push_back(ev); }
  return BZ_OK;
}

extern "C" int bz_decode_graph_capture(bz_model* m, bz_kv* kv, bz_decode_graph** out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || !out) BZ_FAIL(BZ_E_INVALID, "graph capture: bad argument");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch == BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_INVALID, "graph capture: model has no KV cache (use bz_decode_graph_capture_ssm)");
  if (kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "graph capture: cache does not match the model");
  BZ_HIP(hipSetDevice(m->dev->id));
  // stable addresses: the cache must sit at full capacity (cuda_graphs.rs:70)
  BZ_TRY(kv_grow(kv, kv->max_len));
  bz_decode_graph* g = new bz_decode_graph();
  bz_dev_retain(m->dev); g->dev = m->dev;
  g->m = m; g->kv = kv; g->capacity = kv->max_len;
  int rc = graph_capture_common(g, view_of(kv));
  if (rc != BZ_OK) { bz_decode_graph_free(g); return rc; }
  *out = g;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_capture_paged(bz_model* m, bz_paged_kv* kv, int max_blocks, bz_decode_graph** out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || !out || max_blocks <= 0) BZ_FAIL(BZ_E_INVALID, "graph capture: bad argument");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch == BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_INVALID, "graph capture: model has no KV cache (use bz_decode_graph_capture_ssm)");
  if (kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "graph capture: cache does not match the model");
  BZ_HIP(hipSetDevice(m->dev->id));
  bz_decode_graph* g = new bz_decode_graph();
  bz_dev_retain(m->dev); g->dev = m->dev;
  g->m = m; g->pkv = kv; g->max_blocks = max_blocks; g->capacity = std::min(max_blocks * kv->block_size, m->cfg.max_seq_len);
  BZ_HIP(hipMalloc(&g->block_table, (size_t)max_blocks * 4));
  BZ_HIP(hipMemset(g->block_table, 0, (size_t)max_blocks * 4));
  int rc = graph_capture_common(g, view_of(kv, g->block_table, nullptr));
  if (rc != BZ_OK) { bz_decode_graph_free(g); return rc; }
  *out = g;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_capture_ssm(bz_model* m, bz_ssm_state* st, bz_decode_graph** out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !out) BZ_FAIL(BZ_E_INVALID, "graph capture: bad argument");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  BZ_TRY(check_ssm(m, st));
  BZ_HIP(hipSetDevice(m->dev->id));
  bz_decode_graph* g = new bz_decode_graph();
  bz_dev_retain(m->dev); g->dev = m->dev;
  g->m = m; g->ssm = st;
  int rc = graph_capture_common(g, KvView{});
  if (rc != BZ_OK) { bz_decode_graph_free(g); return rc; }
  *out = g;
  return BZ_OK;
  BZ_API_END
}
// ---------------------------------------------------------------------------------------------------------
// Batched decode graph (cuda_graphs_batched.rs:43-257): ONE hipGraph decodes one token for N sequences over a shared paged cache -- the
// weight-sharing multi-row step of bz_forward_paged_batch between two bookkeeping kernels.  Stable-address device buffers, as the reference's
// BatchedGraphState: token_buf [N], slot_mapping [N], block_table [N, max_blocks], next_token_buf [N]; beyond the reference (one shared
// seq_len_k), every sequence has its own device-resident position, the argmax is fed back on the device and the slot comes from the block table,
// so consecutive replays need no host work until a sequence crosses into a block the table does not hold yet.
// ---------------------------------------------------------------------------------------------------------
struct bz_batch_graph {
  bz_model* m = nullptr; bz_device* dev = nullptr; bz_paged_kv* kv = nullptr;
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  int N = 0, max_blocks = 0, capacity = 0;
  long long* tok = nullptr; long long* next = nullptr; int* pos = nullptr; int* slot = nullptr; int* table = nullptr; int* step = nullptr;
  bz_tensor* logits = nullptr;        // [N, vocab] of the last replay
  long long* log = nullptr;           // pinned host [LOGCAP][N]
  std::vector<int> host_pos;          // host copy of the positions (limit checks, cache bookkeeping)
  long long replays = 0;
  static const int LOGCAP = 1024;
};
extern "C" int bz_decode_batch_graph_free(bz_batch_graph* g) {
  BZ_API_BEGIN
  if (!g) return BZ_OK;
  if (g->dev) hipSetDevice(g->dev->id);
  hipDeviceSynchronize();
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  for (void* p : {(void*)g->tok, (void*)g->next, (void*)g->pos, (void*)g->slot, (void*)g->table, (void*)g->step}) if (p) hipFree(p);
  if (g->log) hipHostFree(g->log);
  if (g->logits) bz_tensor_free(g->logits);
  if (g->dev) bz_dev_release(g->dev);
  delete g;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_batch_graph_capture(bz_model* m, bz_paged_kv* kv, int N, int max_blocks, bz_batch_graph** out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || !out || N < 2 || N > 512 || max_blocks <= 0) BZ_FAIL(BZ_E_INVALID, "batch graph capture: bad argument (2 <= N <= 512)");
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (m->cfg.arch != BZ_ARCH_LLAMA) BZ_FAIL(BZ_E_UNSUPPORTED, "batch graph capture: llama family only");
  if (kv->layers != m->cfg.n_layers || kv->n_kv != m->cfg.n_kv_heads || kv->hd != m->cfg.head_dim) BZ_FAIL(BZ_E_INVALID, "batch graph capture: cache does not match the model");
  const int capacity = std::min(max_blocks * kv->block_size, m->cfg.max_seq_len);
  const LinearDev& LH = m->lm_head.parts[0];
  if (!prefill_eligible(m, std::max(N, prefill_min_rows()), capacity, true) || LH.wdt != m->cfg.act_dtype || LH.K % 64)
    BZ_FAIL(BZ_E_UNSUPPORTED, "batch graph capture: the model does not take the weight-sharing multi-row step (int4 without act-order or dense 16-bit weights, 16-bit lm_head)");
  BZ_HIP(hipSetDevice(m->dev->id));
  BZ_TRY(prefill_ws(m, N));                      // workspace before the capture (allocation synchronises)
  bz_batch_graph* g = new bz_batch_graph();
  bz_dev_retain(m->dev); g->dev = m->dev;
  g->m = m; g->kv = kv; g->N = N; g->max_blocks = max_blocks; g->capacity = capacity; g->host_pos.assign(N, -1);
  int rc = BZ_OK;
  auto fail = [&](int code) { bz_decode_batch_graph_free(g); return code; };
  if (hipMalloc(&g->tok, (size_t)N * 8) != hipSuccess || hipMalloc(&g->next, (size_t)N * 8) != hipSuccess || hipMalloc(&g->pos, (size_t)N * 4) != hipSuccess ||
      hipMalloc(&g->slot, (size_t)N * 4) != hipSuccess || hipMalloc(&g->table, (size_t)N * max_blocks * 4) != hipSuccess || hipMalloc(&g->step, 64) != hipSuccess ||
      hipHostMalloc(&g->log, sizeof(long long) * bz_batch_graph::LOGCAP * N, hipHostMallocDefault) != hipSuccess)
    return fail(BZ_E_OOM);
  hipMemset(g->tok, 0, (size_t)N * 8); hipMemset(g->next, 0, (size_t)N * 8); hipMemset(g->pos, 0, (size_t)N * 4); hipMemset(g->slot, 0, (size_t)N * 4);
  hipMemset(g->table, 0, (size_t)N * max_blocks * 4); hipMemset(g->step, 0, 64);
  memset(g->log, 0xff, sizeof(long long) * bz_batch_graph::LOGCAP * N);
  const int64_t shp[2] = {N, m->cfg.vocab};
  rc = bz_tensor_zeros(m->dev, BZ_F32, shp, 2, &g->logits);
  if (rc != BZ_OK) return fail(rc);
  BZ_HIP(hipDeviceSynchronize());
  hipStream_t cap = nullptr;
  BZ_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  hipError_t eb = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (eb != hipSuccess) { hipStreamDestroy(cap); bz_decode_batch_graph_free(g); BZ_FAIL(BZ_E_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(eb)); }
  tl_capture_stream = cap;
  rc = bzk_batch_advance(cap, g->tok, g->next, g->pos, g->slot, g->table, max_blocks, kv->block_size, N);
  if (rc == BZ_OK) {
    RowsCtx rcx; rcx.row_pos = g->pos; rcx.table_stride = max_blocks; rcx.max_len = capacity;
    rc = prefill_dense(m, g->tok, N, view_of(kv, g->table, nullptr), 0, g->slot, true, g->logits, rcx);
  }
  if (rc == BZ_OK) rc = bzk_batch_argmax(cap, (const float*)g->logits->ptr, m->cfg.vocab, g->next, g->log, g->step, bz_batch_graph::LOGCAP, N);
  tl_capture_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(cap, &graph);
  hipStreamDestroy(cap);
  if (rc != BZ_OK) { if (graph) hipGraphDestroy(graph); return fail(rc); }
  if (e != hipSuccess) { bz_decode_batch_graph_free(g); BZ_FAIL(BZ_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e)); }
  g->graph = graph;
  if (hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) != hipSuccess) { bz_decode_batch_graph_free(g); BZ_FAIL(BZ_E_HIP, "hipGraphInstantiate failed"); }
  *out = g;
  return BZ_OK;
  BZ_API_END
}
// State BEFORE the first replay: tokens[i] = the token sequence i feeds next, seq_lens[i] = its length INCLUDING that token (its position is
// seq_lens[i] - 1), block_table = [N, max_blocks] rows (every block a sequence will reach before the next set_block_table call).
extern "C" int bz_decode_batch_graph_seed(bz_batch_graph* g, const int64_t* tokens, const int32_t* seq_lens, const int32_t* block_table) {
  BZ_API_BEGIN
  if (!g || !tokens || !seq_lens || !block_table) BZ_FAIL(BZ_E_INVALID, "batch graph seed: null argument");
  std::lock_guard<std::recursive_mutex> lock__(g->m->mu);
  std::vector<int> p0(g->N);
  for (int i = 0; i < g->N; i++) {
    if (seq_lens[i] <= 0 || seq_lens[i] > g->capacity) BZ_FAIL(BZ_E_INVALID, "batch graph seed: sequence %d has length %d (capacity %d)", i, seq_lens[i], g->capacity);
    p0[i] = seq_lens[i] - 2;                   // the advance kernel in front of the forward adds 1
    g->host_pos[i] = seq_lens[i] - 2;
  }
  hipStream_t st = g->m->dev->stream;
  int z = 0;
  BZ_HIP(hipMemcpyAsync(g->next, tokens, (size_t)g->N * 8, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(g->pos, p0.data(), (size_t)g->N * 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(g->table, block_table, (size_t)g->N * g->max_blocks * 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(g->step, &z, 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  g->replays = 0;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_batch_graph_set_block_table(bz_batch_graph* g, const int32_t* block_table) {
  BZ_API_BEGIN
  if (!g || !block_table) BZ_FAIL(BZ_E_INVALID, "batch graph set_block_table: null argument");
  hipStream_t st = g->m->dev->stream;
  BZ_HIP(hipMemcpyAsync(g->table, block_table, (size_t)g->N * g->max_blocks * 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  return BZ_OK;
  BZ_API_END
}
// one decode step for all N sequences: feeds back the previous argmax (or the seeded tokens), advances the positions, leaves the new argmax per sequence
extern "C" int bz_decode_batch_graph_replay(bz_batch_graph* g) {
  BZ_API_BEGIN
  if (!g || !g->exec) BZ_FAIL(BZ_E_INVALID, "null batch graph");
  std::lock_guard<std::recursive_mutex> lock__(g->m->mu);
  int maxlen = 0;
  for (int i = 0; i < g->N; i++) {
    if (g->host_pos[i] + 1 >= g->capacity) BZ_FAIL(BZ_E_INVALID, "batch graph replay: sequence %d would reach position %d, beyond the capacity %d the step was captured over", i, g->host_pos[i] + 1, g->capacity);
    maxlen = std::max(maxlen, g->host_pos[i] + 2);
  }
  BZ_HIP(hipGraphLaunch(g->exec, g->m->dev->stream));
  for (int i = 0; i < g->N; i++) g->host_pos[i]++;
  if (g->kv->seq_len < maxlen) g->kv->seq_len = maxlen;
  g->replays++;
  return BZ_OK;
  BZ_API_END
}
// tokens produced by replay number `step` (0-based since the seed), host [N]; waits for the device
extern "C" int bz_decode_batch_graph_read_tokens(bz_batch_graph* g, int64_t step, int64_t* tokens_out) {
  BZ_API_BEGIN
  if (!g || !tokens_out || step < 0 || step >= g->replays || step < g->replays - bz_batch_graph::LOGCAP) BZ_FAIL(BZ_E_INVALID, "batch graph read_tokens: step out of range");
  BZ_HIP(hipStreamSynchronize(g->m->dev->stream));
  const long long* row = g->log + (size_t)(step % bz_batch_graph::LOGCAP) * g->N;
  for (int i = 0; i < g->N; i++) tokens_out[i] = row[i];
  return BZ_OK;
  BZ_API_END
}
// logits [N, vocab] of the last replay (device tensor owned by the graph)
extern "C" int bz_decode_batch_graph_logits(bz_batch_graph* g, bz_tensor** logits_out) {
  BZ_API_BEGIN
  if (!g || !logits_out) BZ_FAIL(BZ_E_INVALID, "batch graph logits: null argument");
  *logits_out = g->logits;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_decode_graph_set_block_table(bz_decode_graph* g, const int32_t* bt, int n) {
  BZ_API_BEGIN
  if (!g || !g->block_table || !bt || n < 0 || n > g->max_blocks) BZ_FAIL(BZ_E_INVALID, "set_block_table: bad argument");
  BZ_HIP(hipMemcpyAsync(g->block_table, bt, (size_t)n * 4, hipMemcpyHostToDevice, g->m->dev->stream));
  BZ_HIP(hipStreamSynchronize(g->m->dev->stream));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_seed(bz_decode_graph* g, int64_t token, int position) {
  BZ_API_BEGIN
  if (!g) BZ_FAIL(BZ_E_INVALID, "null graph");
  // the captured kernels index K/V rows, the RoPE tables and the block table by the device-resident position without a guard of their own
  const int limit = g->ssm ? INT32_MAX : std::min(g->capacity, g->m->cfg.max_seq_len);
  if (position < 0 || position >= limit) BZ_FAIL(BZ_E_INVALID, "graph seed: position %d out of range (cache capacity %d, max_seq_len %d)", position, g->capacity, g->m->cfg.max_seq_len);
  hipStream_t st = g->m->dev->stream;
  long long t = token; int p = position, z = 0;
  BZ_HIP(hipMemcpyAsync(g->tok_buf, &t, 8, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(g->pos, &p, 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipMemcpyAsync(g->step, &z, 4, hipMemcpyHostToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  g->replays = 0; g->seed_pos = position;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_replay(bz_decode_graph* g) {
  BZ_API_BEGIN
  if (!g || !g->exec) BZ_FAIL(BZ_E_INVALID, "null graph");
  std::lock_guard<std::recursive_mutex> lock__(g->m->mu);
  hipStream_t st = g->m->dev->stream;
  hipGraphExec_t exec = g->exec;
  if (!g->ssm && g->seed_pos + g->replays >= (long long)std::min(g->capacity, g->m->cfg.max_seq_len))
    BZ_FAIL(BZ_E_INVALID, "graph replay: position %lld is beyond the cache capacity %d / max_seq_len %d the step was captured over", g->seed_pos + g->replays, g->capacity,
            g->m->cfg.max_seq_len);
  // the position of this replay is known on the host (seeded position + replays since): long contexts replay the split-KV variant,
  // captured on first need over the same device words and sized for the cache capacity
  if (!g->ssm && g->capacity > 0 && att_positions_for((int)(g->seed_pos + g->replays + 1)) > 0) {
    if (!g->exec_long) {
      BZ_HIP(hipSetDevice(g->m->dev->id));
      BZ_TRY(graph_capture_variant(g, g->capacity, &g->graph_long, &g->exec_long));
    }
    exec = g->exec_long;
  }
  BZ_HIP(hipGraphLaunch(exec, st));
  BZ_HIP(hipEventRecord(g->evs[g->replays % g->evs.size()], st));
  g->replays++;
  if (g->kv) g->kv->seq_len++;
  if (g->pkv) g->pkv->seq_len++;
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_read_token(bz_decode_graph* g, int64_t step, int64_t* out) {
  BZ_API_BEGIN
  if (!g || !out || step < 0 || step >= g->replays) BZ_FAIL(BZ_E_INVALID, "read_token: step %lld of %lld", (long long)step, g ? g->replays : 0);
  if (g->replays - step > (long long)g->evs.size()) BZ_HIP(hipStreamSynchronize(g->m->dev->stream));
  else BZ_HIP(hipEventSynchronize(g->evs[step % g->evs.size()]));
  if (g->replays - step > bz_decode_graph::LOGCAP) BZ_FAIL(BZ_E_INVALID, "read_token: step %lld fell out of the token log", (long long)step);
  *out = ((volatile long long*)g->tok_log)[step % bz_decode_graph::LOGCAP];
  return persist_check(g->m->dev);
  BZ_API_END
}
extern "C" int bz_decode_graph_read_logits(bz_decode_graph* g, float* host, size_t n) {
  BZ_API_BEGIN
  if (!g || !host || n > (size_t)g->m->cfg.vocab) BZ_FAIL(BZ_E_INVALID, "read_logits: bad argument");
  BZ_HIP(hipMemcpyAsync(host, g->m->logits, n * 4, hipMemcpyDeviceToHost, g->m->dev->stream));
  BZ_HIP(hipStreamSynchronize(g->m->dev->stream));
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_decode_graph_free(bz_decode_graph* g) {
  BZ_API_BEGIN
  if (!g) return BZ_OK;
  hipStreamSynchronize(g->dev->stream);
  for (auto ev : g->evs) hipEventDestroy(ev);
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  if (g->exec_long) hipGraphExecDestroy(g->exec_long);
  if (g->graph_long) hipGraphDestroy(g->graph_long);
  if (g->tok_buf) hipFree(g->tok_buf);
  if (g->pos) hipFree(g->pos);
  if (g->step) hipFree(g->step);
  if (g->block_table) hipFree(g->block_table);
  if (g->tok_log) hipHostFree(g->tok_log);
  bz_dev_release(g->dev);
  delete g;
  return BZ_OK;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// host decode loop: restates Executor::generate (/root/reference/src/engine/executor_generate.rs:341-410, contiguous
// branch; :182-340 paged branch) and generate_with_graphs (/root/reference/src/engine/cuda_graphs.rs:34-190)
// ---------------------------------------------------------------------------------------------------------
// sampling.rs:169-191 penalty_window
static int penalty_window(const std::vector<uint32_t>& hist, int last_n, std::vector<int64_t>& ids, std::vector<int32_t>& cnts) {
  size_t b = 0;
  if (last_n > 0 && (size_t)last_n < hist.size()) b = hist.size() - last_n;
  ids.clear(); cnts.clear();
  for (size_t i = b; i < hist.size(); i++) {
    size_t j = 0;
    for (; j < ids.size(); j++) if (ids[j] == (int64_t)hist[i]) { cnts[j]++; break; }
    if (j == ids.size()) { ids.push_back(hist[i]); cnts.push_back(1); }
  }
  return (int)ids.size();
}

extern "C" int bz_generate(bz_model* m, const int64_t* prompt, int n_prompt, const bz_gen_config* gc, int64_t* out_tokens, bz_gen_stats* stats) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !prompt || !gc || !out_tokens) BZ_FAIL(BZ_E_INVALID, "generate: bad argument");
  if (n_prompt <= 0) { if (stats) memset(stats, 0, sizeof(*stats)); return BZ_OK; }  // executor_generate.rs:75-77
  const bz_model_config& c = m->cfg;
  bz_device* dev = m->dev;
  BZ_HIP(hipSetDevice(dev->id));
  for (int i = 0; i < n_prompt; i++) if (prompt[i] < 0 || prompt[i] >= c.vocab) BZ_FAIL(BZ_E_INVALID, "generate: prompt token %lld out of vocab", (long long)prompt[i]);
  int max_tokens = std::min(gc->max_tokens, std::max(0, c.max_seq_len - n_prompt));  // :79-82
  const bool greedy = gc->temperature == 0.0f;
  if (gc->use_graph && !greedy) BZ_FAIL(BZ_E_INVALID, "generate: graph mode is greedy-only (cli/run.rs:144-157)");
  if (gc->use_graph && (gc->dry_multiplier > 0.f || gc->typical_p > 0.f || gc->n_logit_bias > 0 || gc->mirostat_mode >= 2))
    BZ_FAIL(BZ_E_INVALID, "generate: graph mode has no host-side sampler options");
  if (gc->n_logit_bias < 0 || (gc->n_logit_bias > 0 && (!gc->logit_bias_ids || !gc->logit_bias_vals))) BZ_FAIL(BZ_E_INVALID, "generate: bad logit_bias arrays");
  const int kv_dt = c.act_dtype;
  int rc = BZ_OK;
  bz_tensor *t_prompt = nullptr, *t_logits = nullptr, *t_tok = nullptr, *t_ids = nullptr, *t_cnts = nullptr, *t_slot = nullptr, *t_bt = nullptr;
  bz_kv* kv = nullptr; bz_paged_kv* pkv = nullptr; bz_decode_graph* graph = nullptr; bz_ssm_state* ssm = nullptr; bz_mirostat* mstate = nullptr;
  const bool mamba = c.arch == BZ_ARCH_MAMBA2;   // executor_generate.rs:123-181
  auto mamba_arch = [](const bz_model_config& cc) { return cc.arch == BZ_ARCH_MAMBA2; };
  std::vector<uint32_t> history(prompt, prompt + n_prompt);
  std::vector<int32_t> bt;
  int n_out = 0, finish = 0;
  auto T0 = std::chrono::steady_clock::now();
  auto T1 = T0;
  std::vector<std::chrono::steady_clock::time_point> tok_t;   // host arrival time of every generated token (cli/bench.rs:285-292: TTFT, inter-token latency)
  tok_t.reserve((size_t)std::max(max_tokens, 0));
  const char* backend = mamba_arch(c) ? "mamba2" : (gc->paged ? "paged" : "contiguous");
  BZ_TRACE("phase=\"prefill_start\" backend=\"%s\" prompt_tokens=%d", backend, n_prompt);   // executor_generate.rs:136,252,355
  int64_t sh1[1] = {1}, shp[1] = {n_prompt}, shv[2] = {1, c.vocab}, sh64[1] = {4096};
#define GEN_TRY(x) do { rc = (x); if (rc != BZ_OK) goto done; } while (0)
  GEN_TRY(bz_tensor_from_host(dev, BZ_I64, shp, 1, prompt, &t_prompt));
  GEN_TRY(bz_tensor_zeros(dev, BZ_F32, shv, 2, &t_logits));
  GEN_TRY(bz_tensor_zeros(dev, BZ_I64, sh1, 1, &t_tok));
  GEN_TRY(bz_tensor_zeros(dev, BZ_I64, sh64, 1, &t_ids));
  GEN_TRY(bz_tensor_zeros(dev, BZ_I32, sh64, 1, &t_cnts));
  if (mamba) {
    GEN_TRY(bz_ssm_state_create(m, 1, c.act_dtype, &ssm));                       // :131-133 LayeredSsmState::new
    GEN_TRY(bz_forward_ssm(m, t_prompt, n_prompt, ssm, t_logits, 0));            // :137
  } else if (gc->paged) {
    const int bs = gc->block_size > 0 ? gc->block_size : 16;
    const int total = n_prompt + max_tokens;
    const int nblocks = (total + bs - 1) / bs + 4;  // executor_generate.rs:191-196
    GEN_TRY(bz_paged_kv_create(dev, c.n_layers, nblocks, bs, c.n_kv_heads, c.head_dim, kv_dt, &pkv));
    // CpuBlockAllocator hands out blocks in order; a private allocator gives 0,1,2,...
    bt.resize(nblocks);
    for (int i = 0; i < nblocks; i++) bt[i] = i;
    int64_t shb[1] = {nblocks}, shs[1] = {n_prompt};
    GEN_TRY(bz_tensor_from_host(dev, BZ_I32, shb, 1, bt.data(), &t_bt));
    std::vector<int32_t> slots(n_prompt);
    for (int i = 0; i < n_prompt; i++) slots[i] = bt[i / bs] * bs + i % bs;  // compute_slot_mapping (batch_decode.rs:81-88)
    GEN_TRY(bz_tensor_from_host(dev, BZ_I32, shs, 1, slots.data(), &t_slot));
    GEN_TRY(bz_forward_paged(m, t_prompt, n_prompt, pkv, t_slot, t_bt, nblocks, n_prompt, 0, t_logits, 0));
  } else {
    const int cap = std::min(n_prompt + max_tokens, c.max_seq_len);  // :346
    GEN_TRY(bz_kv_create(dev, c.n_layers, 1, c.n_kv_heads, std::max(cap, 1), c.max_seq_len, c.head_dim, kv_dt, &kv));
    GEN_TRY(bz_forward_kv(m, t_prompt, n_prompt, kv, 0, t_logits, 0));
  }
  GEN_TRY(bz_device_synchronize(dev));
  T1 = std::chrono::steady_clock::now();
  BZ_TRACE("phase=\"prefill_end\" backend=\"%s\"", backend);                                   // :139,264,360
  BZ_TRACE("phase=\"decode_start\" backend=\"%s\" max_tokens=%d graph=%d", backend, max_tokens, gc->use_graph);   // :140,265,361

  if (gc->use_graph) {
    // cuda_graphs.rs:149-189: first token from the prefill logits, then one graph launch per token
    int64_t tok;
    GEN_TRY(bz_argmax_to_buf(dev, t_logits, 1, c.vocab, t_tok));
    GEN_TRY(bz_tensor_to_host(t_tok, &tok, 8));
    const auto t_first = std::chrono::steady_clock::now();   // the first token is on the host here; capturing the graph comes after it
    if (mamba) GEN_TRY(bz_decode_graph_capture_ssm(m, ssm, &graph));
    else if (gc->paged) { GEN_TRY(bz_decode_graph_capture_paged(m, pkv, (int)bt.size(), &graph)); GEN_TRY(bz_decode_graph_set_block_table(graph, bt.data(), (int)bt.size())); }
    else GEN_TRY(bz_decode_graph_capture(m, kv, &graph));
    GEN_TRY(bz_decode_graph_seed(graph, tok, n_prompt));
    BZ_TRACE("generate: graph captured and seeded with token %lld at position %d", (long long)tok, n_prompt);
    for (int i = 0; i < max_tokens; i++) {
      out_tokens[n_out++] = tok; history.push_back((uint32_t)tok); tok_t.push_back(i == 0 ? t_first : std::chrono::steady_clock::now());
      if (tok == gc->eos_id) { finish = 1; break; }
      if (i + 1 == max_tokens) break;
      auto tl0 = std::chrono::steady_clock::now();
      GEN_TRY(bz_decode_graph_replay(graph));
      auto tl1 = std::chrono::steady_clock::now();
      GEN_TRY(bz_decode_graph_read_token(graph, i, &tok));
      BZ_TRACE("step=%d token=%lld fwd_launch_us=%.1f sync_us=%.1f", i, (long long)tok, std::chrono::duration<double, std::micro>(tl1 - tl0).count(),
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl1).count());   // :313,394 fwd_launch / sync split
    }
  } else {
    std::vector<int64_t> ids; std::vector<int32_t> cnts;
    const bool has_pen = gc->repeat_penalty != 1.0f || gc->frequency_penalty != 0.f || gc->presence_penalty != 0.f;
    // host-side options (sampling.rs:393-437): the reference pulls the logits to the CPU for these, and so does this loop
    const bool needs_cpu = gc->dry_multiplier > 0.f || gc->typical_p > 0.f;
    const bool dyn = !greedy && gc->dynatemp_range > 0.f, miro = gc->mirostat_mode >= 2;
    const bool host_row = needs_cpu || gc->n_logit_bias > 0 || dyn || miro;
    std::vector<float> row(host_row ? (size_t)c.vocab : 0);
    for (int i = 0; i < max_tokens; i++) {
      float temperature = greedy ? 0.0f : gc->temperature;
      if (host_row) {
        GEN_TRY(bz_tensor_to_host(t_logits, row.data(), (size_t)c.vocab * 4));
        if (gc->dry_multiplier > 0.f) GEN_TRY(bz_apply_dry_penalty(row.data(), c.vocab, history.data(), (int64_t)history.size(), gc->dry_multiplier, gc->dry_base > 0 ? gc->dry_base : 2, gc->dry_allowed_length));
        if (gc->typical_p > 0.f) GEN_TRY(bz_apply_typical_filter(row.data(), c.vocab, gc->typical_p));
        if (!miro && gc->n_logit_bias > 0) GEN_TRY(bz_apply_logit_bias(row.data(), c.vocab, gc->logit_bias_ids, gc->logit_bias_vals, gc->n_logit_bias));
        if (!miro && dyn) temperature = bz_compute_dynamic_temperature(row.data(), c.vocab, gc->temperature, gc->dynatemp_range, gc->dynatemp_exponent > 0.f ? gc->dynatemp_exponent : 1.0f);
        if (!miro) GEN_TRY(bz_tensor_copy_from_host(t_logits, row.data(), (size_t)c.vocab * 4));
      }
      if (miro) {       // sampling.rs:96-110,118-150: Mirostat v2 on the CPU row, token goes back to the device
        if (!mstate) GEN_TRY(bz_mirostat_create(gc->mirostat_tau, gc->mirostat_eta, gc->seed, &mstate));
        uint32_t mt = 0;
        GEN_TRY(bz_mirostat_sample(mstate, row.data(), c.vocab, gc->temperature, &mt, nullptr));
        const int64_t mt64 = mt;
        GEN_TRY(bz_tensor_copy_from_host(t_tok, &mt64, 8));
      } else {
      int n = has_pen ? penalty_window(history, gc->repeat_last_n, ids, cnts) : 0;  // sampling.rs:431
      if (n > 4096) { n = 4096; }
      if (n) { GEN_TRY(bz_tensor_copy_from_host(t_ids, ids.data(), (size_t)n * 8)); GEN_TRY(bz_tensor_copy_from_host(t_cnts, cnts.data(), (size_t)n * 4)); }
      GEN_TRY(bz_logits_to_token(dev, t_logits, 1, c.vocab, t_ids, t_cnts, n, gc->repeat_penalty, gc->frequency_penalty, gc->presence_penalty,
                                 temperature, gc->top_k, gc->top_p, gc->min_p, gc->seed + (uint64_t)i, t_tok));
      }
      uint64_t ev;
      GEN_TRY(bz_event_record(dev, &ev));                                           // :367 record_event
      const bool last = i + 1 == max_tokens;
      // :372 the next forward is launched BEFORE the token is read back (token stays on device)
      if (!last) {
        if (mamba) {
          GEN_TRY(bz_forward_ssm(m, t_tok, 1, ssm, t_logits, 0));                  // :148
        } else if (gc->paged) {
          const int bs = pkv->block_size, cur = pkv->seq_len;
          int32_t slot = bt[cur / bs] * bs + cur % bs;
          GEN_TRY(bz_tensor_copy_from_host(t_slot, &slot, 4));
          GEN_TRY(bz_forward_paged(m, t_tok, 1, pkv, t_slot, t_bt, (int)bt.size(), cur + 1, cur, t_logits, 0));
        } else {
          GEN_TRY(bz_forward_kv(m, t_tok, 1, kv, kv->seq_len, t_logits, 0));
        }
      }
      int64_t tok;
      auto ts0 = std::chrono::steady_clock::now();
      GEN_TRY(bz_tensor_to_host_pipelined(t_tok, ev, &tok, 8));                      // :378 read_token_id
      BZ_TRACE("step=%d token=%lld sync_us=%.1f", i, (long long)tok, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ts0).count());
      out_tokens[n_out++] = tok; history.push_back((uint32_t)tok); tok_t.push_back(std::chrono::steady_clock::now());
      if (tok == gc->eos_id) { finish = 1; break; }
    }
  }
  GEN_TRY(bz_device_synchronize(dev));
done:
  {
    auto T2 = std::chrono::steady_clock::now();
    BZ_TRACE("phase=\"decode_end\" backend=\"%s\" generated=%d", backend, n_out);                // :181,340,409
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->prefill_ms = std::chrono::duration<double, std::milli>(T1 - T0).count();
      stats->decode_ms = std::chrono::duration<double, std::milli>(T2 - T1).count();
      stats->n_generated = n_out; stats->finish_reason = finish;
      if (!tok_t.empty()) {   // cli/bench.rs:285-306
        auto ms = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - T0).count(); };
        stats->ttft_ms = ms(tok_t.front()); stats->total_ms = ms(tok_t.back());
        std::vector<double> itl;
        for (size_t i = 1; i < tok_t.size(); i++) itl.push_back(std::chrono::duration<double, std::milli>(tok_t[i] - tok_t[i - 1]).count());
        if (!itl.empty()) {
          std::sort(itl.begin(), itl.end());
          auto pct = [&](double p) { size_t k = (size_t)std::ceil(p / 100.0 * (double)itl.size()); k = k ? k - 1 : 0; return itl[std::min(k, itl.size() - 1)]; };
          stats->itl_p50_ms = pct(50.0); stats->itl_p99_ms = pct(99.0); stats->itl_max_ms = itl.back();
          const double dec = stats->total_ms - stats->ttft_ms;
          stats->decode_tok_per_s = dec > 0.0 ? (double)itl.size() / (dec / 1e3) : 0.0;
        }
      }
    }
  }
  bz_decode_graph_free(graph);
  bz_tensor_free(t_prompt); bz_tensor_free(t_logits); bz_tensor_free(t_tok); bz_tensor_free(t_ids); bz_tensor_free(t_cnts);
  bz_tensor_free(t_slot); bz_tensor_free(t_bt);
  bz_kv_free(kv); bz_paged_kv_free(pkv); bz_ssm_state_free(ssm); bz_mirostat_free(mstate);
  return rc;
#undef GEN_TRY
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// op-level entry points
// ---------------------------------------------------------------------------------------------------------
static int find_linear(bz_model* m, const char* name, LinearDev* out) {
  if (!m || !m->finalized || !name) BZ_FAIL(BZ_E_INVALID, "model not finalized");
  auto it = m->named.find(name);
  if (it == m->named.end()) BZ_FAIL(BZ_E_NOTFOUND, "no linear weight named '%s'", name);
  *out = it->second;
  return BZ_OK;
}

extern "C" int bz_quant_matmul(bz_model* m, const char* name, const bz_tensor* x, int S, bz_tensor* y) {
  BZ_API_BEGIN
  LinearDev L;
  BZ_TRY(find_linear(m, name, &L));
  std::lock_guard<std::recursive_mutex> lock__(m->mu);
  if (!x || !y || x->dtype != BZ_F32 || y->dtype != BZ_F32 || S <= 0 || x->nbytes < (size_t)S * L.K * 4 || y->nbytes < (size_t)S * L.N * 4)
    BZ_FAIL(BZ_E_INVALID, "quant_matmul: x must be F32 [S,%d], y F32 [S,%d]", L.K, L.N);
  BZ_HIP(hipSetDevice(m->dev->id));
  hipStream_t st = m->dev->stream;
  // op-level results are NOT rounded to the activation dtype: the caller sees the f32 accumulator
  long long* acc = nullptr;
  if (L.kind != LK_ROWS) BZ_HIP(hipMalloc(&acc, (size_t)L.N * 8));
  int rc = BZ_OK;
  for (int s = 0; s < S && rc == BZ_OK; s++) {
    Pro p{}; p.mode = PRO_PLAIN; p.src = VSrc{(const float*)x->ptr + (size_t)s * L.K, 0}; p.act = BZ_F32; p.perm = L.perm;
    GemvOut o{};
    if (L.kind == LK_ROWS) { o.direct = (float*)y->ptr + (size_t)s * L.N; rc = bzk_gemv(st, L, p, o, BZ_F32); }
    else {
      rc = bzk_zero64(st, acc, L.N);
      o.acc = acc;
      if (rc == BZ_OK) rc = bzk_gemv(st, L, p, o, BZ_F32);
      if (rc == BZ_OK) rc = bzk_fix_to_f32(st, acc, L.N, BZ_F32, (float*)y->ptr + (size_t)s * L.N);
    }
  }
  hipStreamSynchronize(st);
  if (acc) hipFree(acc);
  return rc;
  BZ_API_END
}

extern "C" int bz_dequant(bz_model* m, const char* name, float* host) {
  BZ_API_BEGIN
  LinearDev L;
  BZ_TRY(find_linear(m, name, &L));
  if (!host) BZ_FAIL(BZ_E_INVALID, "dequant: null output");
  BZ_HIP(hipSetDevice(m->dev->id));
  const size_t n = (size_t)L.N * L.K;
  float* d;
  BZ_HIP(hipMalloc(&d, n * 4));
  int rc = L.kind == LK_Q4G ? bzk_dequant_q4g(m->dev->stream, L, d) : (L.kind == LK_ROWS ? bzk_dequant_rows(m->dev->stream, L, d) : bzk_dequant_gq(m->dev->stream, L, d));
  std::vector<float> tmp;
  if (rc == BZ_OK) rc = hipStreamSynchronize(m->dev->stream) == hipSuccess ? BZ_OK : BZ_E_HIP;  // the stream is non-blocking
  if (rc == BZ_OK) {
    if (L.perm) {
      // kernel order k' -> original k = perm[k']
      tmp.resize(n);
      hipMemcpy(tmp.data(), d, n * 4, hipMemcpyDeviceToHost);
      std::vector<int> perm(L.K);
      hipMemcpy(perm.data(), L.perm, (size_t)L.K * 4, hipMemcpyDeviceToHost);
      for (int r = 0; r < L.N; r++) for (int k = 0; k < L.K; k++) host[(size_t)r * L.K + perm[k]] = tmp[(size_t)r * L.K + k];
    } else {
      hipMemcpy(host, d, n * 4, hipMemcpyDeviceToHost);
    }
  }
  hipFree(d);
  return rc;
  BZ_API_END
}

// ---------------------------------------------------------------------------------------------------------
// op-level entry points for the Mamba2 mixer and the MoE block (bz_ssm_step, bz_conv1d_step, bz_moe_route, bz_moe_grouped_gemv).
// Each runs the kernel the decode step uses, on caller-provided rows, so that the state update / router / grouped GEMV can be compared with
// the oracle on their own (ConvOps and the MoE ops of the trait bound at /root/reference/src/engine/executor.rs:67-80).
// ---------------------------------------------------------------------------------------------------------
static int ssm_op_args(bz_model* m, int layer, bz_ssm_state* st, const bz_tensor* zx, SsmArgs* sa, int* conv_dim_out) {
  if (!m || !m->finalized || m->cfg.arch != BZ_ARCH_MAMBA2) BZ_FAIL(BZ_E_INVALID, "ssm op: not a finalized Mamba2 model");
  BZ_TRY(check_ssm(m, st));
  const bz_model_config& c = m->cfg;
  if (layer < 0 || layer >= c.n_layers) BZ_FAIL(BZ_E_INVALID, "ssm op: layer %d out of range", layer);
  const int DI = c.ssm_d_inner, NH = c.ssm_n_heads, NS = c.ssm_d_state, G = c.ssm_n_groups, KC = c.ssm_conv_kernel;
  const int conv_dim = DI + 2 * G * NS, d_in = 2 * DI + 2 * G * NS + NH;
  if (!zx || zx->dtype != BZ_F32 || zx->nbytes < (size_t)d_in * 4) BZ_FAIL(BZ_E_INVALID, "ssm op: zx must be an F32 tensor of %d values ([z | x B C | dt], the in_proj row)", d_in);
  BZ_HIP(hipSetDevice(m->dev->id));
  const MambaLayerDev& L = m->mlayers[layer];
  SsmArgs a{};
  a.zx = VSrc{zx->ptr, 0}; a.z_off = 0; a.x_off = DI; a.dt_off = DI + conv_dim; a.dt_bias = L.dt_bias; a.A_log = L.A_log; a.D = L.D;
  a.conv_w = L.conv_w; a.conv_b = L.conv_b; a.conv_state = st->conv + (size_t)layer * conv_dim * (KC - 1); a.conv_kernel = KC;
  a.state = (char*)st->ssm + (size_t)layer * NH * c.ssm_head_dim * NS * bz_dtype_size(st->dtype); a.sdt = st->dtype;
  a.n_heads = NH; a.head_dim = c.ssm_head_dim; a.d_state = NS; a.n_groups = G; a.d_inner = DI; a.act = c.act_dtype; a.y = m->ybuf; a.gate = 1; a.vss = m->vss;
  *sa = a; *conv_dim_out = conv_dim;
  return BZ_OK;
}

extern "C" int bz_conv1d_step(bz_model* m, int layer, bz_ssm_state* state, const bz_tensor* zx, bz_tensor* xbc_out) {
  BZ_API_BEGIN
  SsmArgs a; int conv_dim = 0;
  BZ_TRY(ssm_op_args(m, layer, state, zx, &a, &conv_dim));
  if (!xbc_out || xbc_out->dtype != BZ_F32 || xbc_out->nbytes < (size_t)conv_dim * 4) BZ_FAIL(BZ_E_INVALID, "conv1d_step: xbc_out must hold %d F32 values", conv_dim);
  std::lock_guard<std::recursive_mutex> lk(m->mu);
  hipStream_t st = m->dev->stream;
  a.conv_out = (float*)xbc_out->ptr; a.conv_only = 1;
  BZ_TRY(bzk_ssm_step(st, a));                        // the x channels' window moves inside the launch ...
  const ConvShift shf{a.conv_state, a.zx, a.x_off, a.d_inner, conv_dim - a.d_inner, a.conv_kernel};
  BZ_TRY(bzk_conv_shift(st, shf, a.act));             // ... the B / C channels' behind it (the decode step: a side duty of the out_proj launch)
  BZ_HIP(hipStreamSynchronize(st));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_ssm_step(bz_model* m, int layer, bz_ssm_state* state, const bz_tensor* zx, bz_tensor* y_out) {
  BZ_API_BEGIN
  SsmArgs a; int conv_dim = 0;
  BZ_TRY(ssm_op_args(m, layer, state, zx, &a, &conv_dim));
  if (!y_out || y_out->dtype != BZ_F32 || y_out->nbytes < (size_t)a.d_inner * 4) BZ_FAIL(BZ_E_INVALID, "ssm_step: y_out must hold %d F32 values", a.d_inner);
  std::lock_guard<std::recursive_mutex> lk(m->mu);
  hipStream_t st = m->dev->stream;
  a.y = (float*)y_out->ptr;
  BZ_TRY(bzk_ssm_step(st, a));
  const ConvShift shf{a.conv_state, a.zx, a.x_off, a.d_inner, conv_dim - a.d_inner, a.conv_kernel};
  BZ_TRY(bzk_conv_shift(st, shf, a.act));
  BZ_HIP(hipStreamSynchronize(st));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_ssm_state_read(const bz_ssm_state* st, int layer, int which, float* host, size_t n) {
  BZ_API_BEGIN
  if (!st || !host || layer < 0 || layer >= st->layers || (which != 0 && which != 1)) BZ_FAIL(BZ_E_INVALID, "ssm_state_read: bad argument");
  BZ_HIP(hipSetDevice(st->dev->id));
  BZ_HIP(hipStreamSynchronize(st->dev->stream));
  if (which == 1) {
    const size_t per = (size_t)st->conv_dim * (st->kc - 1);
    if (n != per) BZ_FAIL(BZ_E_INVALID, "ssm_state_read: the conv window of a layer has %zu values", per);
    BZ_HIP(hipMemcpy(host, st->conv + (size_t)layer * per, per * 4, hipMemcpyDeviceToHost));
    return BZ_OK;
  }
  const size_t per = (size_t)st->n_heads * st->head_dim * st->d_state, es = bz_dtype_size(st->dtype);
  if (n != per) BZ_FAIL(BZ_E_INVALID, "ssm_state_read: the SSM state of a layer has %zu values", per);
  if (st->dtype == BZ_F32) { BZ_HIP(hipMemcpy(host, (const char*)st->ssm + (size_t)layer * per * 4, per * 4, hipMemcpyDeviceToHost)); return BZ_OK; }
  std::vector<uint16_t> raw(per);
  BZ_HIP(hipMemcpy(raw.data(), (const char*)st->ssm + (size_t)layer * per * es, per * es, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < per; i++) {
    if (st->dtype == BZ_BF16) { const uint32_t u = (uint32_t)raw[i] << 16; memcpy(&host[i], &u, 4); }
    else host[i] = __half2float(__ushort_as_half(raw[i]));
  }
  return BZ_OK;
  BZ_API_END
}

static int moe_op_layer(bz_model* m, int layer, const DsLayerDev** L) {
  if (!m || !m->finalized || m->cfg.arch != BZ_ARCH_DEEPSEEK2) BZ_FAIL(BZ_E_INVALID, "moe op: not a finalized DeepSeek-V2 model");
  if (layer < 0 || layer >= m->cfg.n_layers || !m->dlayers[layer].is_moe) BZ_FAIL(BZ_E_INVALID, "moe op: layer %d is not a MoE layer", layer);
  BZ_HIP(hipSetDevice(m->dev->id));
  *L = &m->dlayers[layer];
  return BZ_OK;
}

extern "C" int bz_moe_route(bz_model* m, int layer, const bz_tensor* hidden, bz_tensor* sel_out, bz_tensor* w_out, bz_tensor* xn_out) {
  BZ_API_BEGIN
  const DsLayerDev* L = nullptr;
  BZ_TRY(moe_op_layer(m, layer, &L));
  const bz_model_config& c = m->cfg;
  const int H = c.hidden, slots = c.moe_top_k + c.moe_n_shared;
  if (!hidden || hidden->dtype != BZ_F32 || hidden->nbytes < (size_t)H * 4) BZ_FAIL(BZ_E_INVALID, "moe_route: hidden must be F32 [%d]", H);
  if (!sel_out || sel_out->dtype != BZ_I32 || sel_out->nbytes < (size_t)slots * 4 || !w_out || w_out->dtype != BZ_F32 || w_out->nbytes < (size_t)slots * 4)
    BZ_FAIL(BZ_E_INVALID, "moe_route: sel_out I32 / w_out F32 must hold top_k + n_shared = %d values", slots);
  if (xn_out && (xn_out->dtype != BZ_F32 || xn_out->nbytes < (size_t)H * 4)) BZ_FAIL(BZ_E_INVALID, "moe_route: xn_out must be F32 [%d]", H);
  std::lock_guard<std::recursive_mutex> lk(m->mu);
  hipStream_t st = m->dev->stream;
  Pro pf{}; pf.mode = PRO_NORM; pf.src = VSrc{nullptr, 0}; pf.h_in = (const float*)hidden->ptr; pf.h_out = nullptr; pf.norm_w = L->ffn_norm; pf.eps = c.rms_eps; pf.H = H; pf.act = c.act_dtype;
  BZ_TRY(bzk_moe_router(st, pf, L->router, L->router_dt, c.moe_n_experts, c.moe_top_k, c.moe_n_shared, c.moe_routed_scale, c.moe_norm_topk, m->moe_xn, m->moe_sel, m->moe_w,
                        m->moe_lg, m->moe_cnt));
  BZ_HIP(hipMemcpyAsync(sel_out->ptr, m->moe_sel, (size_t)slots * 4, hipMemcpyDeviceToDevice, st));
  BZ_HIP(hipMemcpyAsync(w_out->ptr, m->moe_w, (size_t)slots * 4, hipMemcpyDeviceToDevice, st));
  if (xn_out) BZ_HIP(hipMemcpyAsync(xn_out->ptr, m->moe_xn, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
  BZ_HIP(hipStreamSynchronize(st));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_moe_grouped_gemv(bz_model* m, int layer, int which, const bz_tensor* sel, int n_slots, const bz_tensor* x, bz_tensor* y) {
  BZ_API_BEGIN
  const DsLayerDev* L = nullptr;
  BZ_TRY(moe_op_layer(m, layer, &L));
  const bz_model_config& c = m->cfg;
  const int H = c.hidden, MI = c.moe_inter, act = c.act_dtype;
  if (which != 0 && which != 1) BZ_FAIL(BZ_E_INVALID, "moe_grouped_gemv: which must be 0 (gate/up) or 1 (down)");
  if (n_slots <= 0 || n_slots > 128 || !sel || sel->dtype != BZ_I32 || sel->nbytes < (size_t)n_slots * 4) BZ_FAIL(BZ_E_INVALID, "moe_grouped_gemv: sel must be I32 [n_slots <= 128]");
  const int N = which == 0 ? 2 * MI : H, K = which == 0 ? H : MI;
  const size_t xin = which == 0 ? (size_t)H : (size_t)n_slots * 2 * MI;
  if (!x || x->dtype != BZ_F32 || x->nbytes < xin * 4) BZ_FAIL(BZ_E_INVALID, "moe_grouped_gemv: x must hold %zu F32 values", xin);
  if (!y || y->dtype != BZ_F32 || y->nbytes < (size_t)n_slots * N * 4) BZ_FAIL(BZ_E_INVALID, "moe_grouped_gemv: y must hold n_slots x %d F32 values", N);
  {
    std::vector<int> hs(n_slots);
    BZ_HIP(hipMemcpy(hs.data(), sel->ptr, (size_t)n_slots * 4, hipMemcpyDeviceToHost));
    for (int e : hs) if (e < 0 || e >= c.moe_n_experts + c.moe_n_shared) BZ_FAIL(BZ_E_INVALID, "moe_grouped_gemv: expert index %d out of range", e);
  }
  std::lock_guard<std::recursive_mutex> lk(m->mu);
  hipStream_t st = m->dev->stream;
  const size_t es = bz_dtype_size(L->e_dt);
  const bool r2 = bzk_moe_rows2_ok(L->e_dt, K);
  long long* acc = nullptr;
  BZ_HIP(hipMalloc((void**)&acc, (size_t)n_slots * N * 8));
  hipError_t e0 = hipMemsetAsync(acc, 0, (size_t)n_slots * N * 8, st);
  int rc = e0 == hipSuccess ? BZ_OK : BZ_E_HIP;
  MoeGemvArgs g{};
  g.w = which == 0 ? L->e_gu : L->e_dn; g.expert_stride = (long long)N * K; g.sel = (const int*)sel->ptr; g.N = N; g.K = K;
  g.src_stride = which == 0 ? 0 : 2 * MI;
  g.out = (float*)y->ptr; g.out_stride = N;
  g.acc = acc; g.acc_stride = N; g.acc_slots = n_slots;
  Pro p{}; p.mode = which == 0 ? PRO_PLAIN : PRO_SILU; p.src = VSrc{x->ptr, 0}; p.H = which == 0 ? 0 : MI; p.act = act;
  // the decode step's form: the balanced role kernel into fixed-point accumulators (16-bit experts), converted with one rounding; else the 16-row workgroup form
  const bool split = r2 || which == 1;
  if (rc == BZ_OK) rc = bzk_moe_gemv(st, g, L->e_dt, n_slots, p, act, split, (double)n_slots * N * K * es);
  if (rc == BZ_OK && split) rc = bzk_fix_to_f32(st, acc, n_slots * N, act, (float*)y->ptr);
  hipStreamSynchronize(st);
  hipFree(acc);
  return rc;
  BZ_API_END
}

extern "C" int bz_expf_spec(const float* x, int n, float* y) {
  BZ_API_BEGIN
  if (!x || !y || n < 0) BZ_FAIL(BZ_E_INVALID, "bad argument");
  for (int i = 0; i < n; i++) y[i] = bz_expf(x[i]);   // the host compilation of the device function (bz_internal.h)
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_rms_norm(bz_device* dev, const bz_tensor* x, const bz_tensor* prev, const bz_tensor* w, int rows, int n, float eps, int act,
                           bz_tensor* y, bz_tensor* h_out) {
  BZ_API_BEGIN
  if (!dev || !x || !w || !y || rows <= 0 || n <= 0) BZ_FAIL(BZ_E_INVALID, "rms_norm: bad argument");
  const size_t need = (size_t)rows * n * 4;
  if (x->dtype != BZ_F32 || y->dtype != BZ_F32 || w->dtype != BZ_F32 || x->nbytes < need || y->nbytes < need || w->nbytes < (size_t)n * 4 ||
      (prev && prev->nbytes < need) || (h_out && h_out->nbytes < need))
    BZ_FAIL(BZ_E_INVALID, "rms_norm: F32 tensors of [rows,n] / [n] required");
  BZ_HIP(hipSetDevice(dev->id));
  BZ_TRY(bzk_rms_norm(dev->stream, (const float*)x->ptr, prev ? (const float*)prev->ptr : nullptr, (const float*)w->ptr, rows, n, eps, act,
                      (float*)y->ptr, h_out ? (float*)h_out->ptr : nullptr));
  BZ_HIP(hipStreamSynchronize(dev->stream));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_rope(bz_model* m, bz_tensor* x, int S, int n_heads, int position) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !x || x->dtype != BZ_F32 || S <= 0 || n_heads <= 0) BZ_FAIL(BZ_E_INVALID, "rope: bad argument");
  if (x->nbytes < (size_t)S * n_heads * m->cfg.head_dim * 4) BZ_FAIL(BZ_E_INVALID, "rope: x must be F32 [S,n_heads,head_dim]");
  if (position < 0 || position + S > m->cfg.max_seq_len) BZ_FAIL(BZ_E_INVALID, "rope: positions out of range");
  BZ_HIP(hipSetDevice(m->dev->id));
  BZ_TRY(bzk_rope(m->dev->stream, (float*)x->ptr, S, n_heads, m->cfg.head_dim, position, m->cos_t, m->sin_t, m->cfg.rope_interleaved, m->cfg.act_dtype));
  BZ_HIP(hipStreamSynchronize(m->dev->stream));
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_silu_mul(bz_device* dev, const bz_tensor* g, const bz_tensor* u, int64_t n, int act, bz_tensor* y) {
  BZ_API_BEGIN
  if (!dev || !g || !u || !y || n <= 0 || g->nbytes < (size_t)n * 4 || u->nbytes < (size_t)n * 4 || y->nbytes < (size_t)n * 4) BZ_FAIL(BZ_E_INVALID, "silu_mul: bad argument");
  BZ_HIP(hipSetDevice(dev->id));
  BZ_TRY(bzk_silu_mul(dev->stream, (const float*)g->ptr, (const float*)u->ptr, n, act, (float*)y->ptr));
  BZ_HIP(hipStreamSynchronize(dev->stream));
  return BZ_OK;
  BZ_API_END
}

static int attn_common(bz_model* m, const bz_tensor* q, const KvView& view, int layer, int len, bz_tensor* out) {
  const bz_model_config& c = m->cfg;
  const size_t need = (size_t)c.n_heads * c.head_dim * 4;
  if (!q || !out || q->dtype != BZ_F32 || out->dtype != BZ_F32 || q->nbytes < need || out->nbytes < need) BZ_FAIL(BZ_E_INVALID, "attn_decode: q/out must be F32 [n_heads,head_dim]");
  if (layer < 0 || layer >= c.n_layers || len <= 0) BZ_FAIL(BZ_E_INVALID, "attn_decode: bad layer/len");
  hipStream_t st = m->dev->stream;
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, m->pos_tmp, len);
  AttnArgs aa{};
  aa.qkv = VSrc{q->ptr, 0}; aa.cos_t = m->cos_t; aa.sin_t = m->sin_t; aa.interleaved = c.rope_interleaved; aa.pos = m->pos_tmp;
  aa.nq = c.n_heads; aa.nkv = c.n_kv_heads; aa.hd = c.head_dim; aa.act = c.act_dtype; aa.kv = view; aa.layer = layer; aa.out = (float*)out->ptr;
  aa.q_only = 1;
  BZ_TRY(bzk_attn_decode(st, aa));
  BZ_HIP(hipStreamSynchronize(st));
  return BZ_OK;
}
extern "C" int bz_attn_decode(bz_model* m, const bz_tensor* q, bz_kv* kv, int layer, int len, bz_tensor* out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || len > kv->cap) BZ_FAIL(BZ_E_INVALID, "attn_decode: bad argument");
  BZ_HIP(hipSetDevice(m->dev->id));
  return attn_common(m, q, view_of(kv), layer, len, out);
  BZ_API_END
}
extern "C" int bz_paged_attn_decode(bz_model* m, const bz_tensor* q, bz_paged_kv* kv, int layer, const bz_tensor* block_table, int len, bz_tensor* out) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || !block_table || block_table->dtype != BZ_I32) BZ_FAIL(BZ_E_INVALID, "paged_attn_decode: bad argument");
  if ((size_t)((len + kv->block_size - 1) / kv->block_size) * 4 > block_table->nbytes) BZ_FAIL(BZ_E_INVALID, "paged_attn_decode: block_table too short");
  BZ_HIP(hipSetDevice(m->dev->id));
  return attn_common(m, q, view_of(kv, (const int*)block_table->ptr, nullptr), layer, len, out);
  BZ_API_END
}
extern "C" int bz_kv_insert(bz_model* m, bz_kv* kv, int layer, int position, const bz_tensor* k, const bz_tensor* v) {
  BZ_API_BEGIN
  if (!m || !m->finalized || !kv || !k || !v) BZ_FAIL(BZ_E_INVALID, "kv_insert: bad argument");
  const size_t need = (size_t)kv->n_kv * kv->hd * 4;
  if (k->dtype != BZ_F32 || v->dtype != BZ_F32 || k->nbytes < need || v->nbytes < need) BZ_FAIL(BZ_E_INVALID, "kv_insert: k/v must be F32 [n_kv_heads,head_dim]");
  if (layer < 0 || layer >= kv->layers || position < 0 || position >= kv->max_len) BZ_FAIL(BZ_E_INVALID, "kv_insert: bad layer/position");
  BZ_HIP(hipSetDevice(m->dev->id));
  BZ_TRY(kv_grow(kv, position + 1));
  hipStream_t st = m->dev->stream;
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, st, m->pos_tmp, position);
  BZ_TRY(bzk_kv_insert(st, view_of(kv), layer, (const float*)k->ptr, (const float*)v->ptr, m->pos_tmp, kv->n_kv, kv->hd));
  BZ_HIP(hipStreamSynchronize(st));
  if (position + 1 > kv->seq_len) kv->seq_len = position + 1;
  return BZ_OK;
  BZ_API_END
}
