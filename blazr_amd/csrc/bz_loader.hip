// bz_loader.hip -- checkpoint ingestion: the host-side loader of the reference, restated in C++ above the C ABI.
//
//   detect_model_source            /root/reference/src/loader/detect.rs:34-150          -> bz_detect_model_source
//   SafeTensorsLoader (boostr)     used at loader/safetensors/regular.rs:38-117         -> StLoader (header JSON + mmap, sharded index)
//   load_or_create_config          loader/safetensors/config.rs:14-70, HF config.json   -> bz_config_from_hf_json
//   detect_architecture_from_loader loader/safetensors/detect_arch.rs:13-63             -> detect_from_tensors
//   detect_awq / detect_gptq / group size  detect_arch.rs:66-196                         -> detect_quant
//   load_safetensors_awq           loader/safetensors/awq.rs:40-262                      -> load_safetensors (AWQ branch)
//   load_safetensors_gptq          loader/safetensors/gptq.rs:40-262                     -> load_safetensors (GPTQ branch)
//   config_from_gguf_metadata      loader/gguf.rs:101-306                                -> gguf_config
//   VarMap::from_gguf              loader/gguf.rs:33                                     -> load_gguf (llama-family tensor names)
// Pure host code: it only calls bz_model_create / bz_model_add_* / bz_model_finalize.
#include <fcntl.h>
#include <glob.h>
#include <stdarg.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "bz_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// minimal JSON DOM
// ---------------------------------------------------------------------------------------------------------
struct JVal {
  enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  bool b = false; double n = 0; std::string s;
  std::vector<JVal> a;
  std::vector<std::pair<std::string, JVal>> o;
  const JVal* get(const char* k) const {
    if (t != OBJ) return nullptr;
    for (auto& kv : o) if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool num(const char* k, double* out) const { const JVal* v = get(k); if (!v || v->t != NUM) return false; *out = v->n; return true; }
  long long i64(const char* k, long long dflt) const { double d; return num(k, &d) ? (long long)d : dflt; }
  double f64(const char* k, double dflt) const { double d; return num(k, &d) ? d : dflt; }
  std::string str(const char* k, const char* dflt = "") const { const JVal* v = get(k); return (v && v->t == STR) ? v->s : std::string(dflt); }
  bool boolean(const char* k, bool dflt) const { const JVal* v = get(k); return (v && v->t == BOOL) ? v->b : dflt; }
};

struct JParser {
  const char* p; const char* e; bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(e - p) >= n && !memcmp(p, s, n)) { p += n; return true; } return false; }
  std::string string() {
    std::string out;
    if (p >= e || *p != '"') { ok = false; return out; }
    p++;
    while (p < e && *p != '"') {
      if (*p == '\\' && p + 1 < e) {
        p++;
        switch (*p) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (p + 4 >= e) { ok = false; return out; }
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += *p;
        }
        p++;
      } else out += *p++;
    }
    if (p >= e) { ok = false; return out; }
    p++;
    return out;
  }
  JVal value(int depth = 0) {
    JVal v;
    ws();
    if (p >= e || depth > 64) { ok = false; return v; }
    if (*p == '{') {
      v.t = JVal::OBJ; p++; ws();
      if (p < e && *p == '}') { p++; return v; }
      while (ok) {
        ws();
        std::string k = string();
        ws();
        if (!ok || p >= e || *p != ':') { ok = false; break; }
        p++;
        v.o.emplace_back(k, value(depth + 1));
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == '}') { p++; break; }
        ok = false;
      }
    } else if (*p == '[') {
      v.t = JVal::ARR; p++; ws();
      if (p < e && *p == ']') { p++; return v; }
      while (ok) {
        v.a.push_back(value(depth + 1));
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == ']') { p++; break; }
        ok = false;
      }
    } else if (*p == '"') { v.t = JVal::STR; v.s = string(); }
    else if (lit("true")) { v.t = JVal::BOOL; v.b = true; }
    else if (lit("false")) { v.t = JVal::BOOL; v.b = false; }
    else if (lit("null")) { v.t = JVal::NUL; }
    else {
      // the header is a mapped file region with no terminating NUL: parse the number from a bounded copy
      char buf[64]; size_t n = 0;
      while (p + n < e && n < sizeof(buf) - 1 && (isdigit((unsigned char)p[n]) || p[n] == '-' || p[n] == '+' || p[n] == '.' || p[n] == 'e' || p[n] == 'E')) { buf[n] = p[n]; n++; }
      buf[n] = 0;
      char* end = nullptr;
      v.t = JVal::NUM; v.n = strtod(buf, &end);
      if (n == 0 || end == buf) ok = false; else p += (end - buf);
    }
    return v;
  }
};
bool json_parse(const char* text, size_t n, JVal* out) {
  JParser ps{text, text + n};
  *out = ps.value();
  ps.ws();
  return ps.ok && ps.p == ps.e;
}

// ---------------------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------------------
bool is_file(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
std::string dirname_of(const std::string& p) { size_t i = p.find_last_of('/'); return i == std::string::npos ? "." : (i == 0 ? "/" : p.substr(0, i)); }
std::string basename_of(const std::string& p) { size_t i = p.find_last_of('/'); return i == std::string::npos ? p : p.substr(i + 1); }
std::string ext_of(const std::string& p) { std::string b = basename_of(p); size_t i = b.find_last_of('.'); return i == std::string::npos ? "" : b.substr(i + 1); }
bool ends_with(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); }
std::vector<std::string> glob_list(const std::string& pattern) {
  std::vector<std::string> out;
  glob_t g;
  if (glob(pattern.c_str(), 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; i++) out.push_back(g.gl_pathv[i]);
  globfree(&g);
  return out;
}
bool read_text(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  out->resize(n > 0 ? n : 0);
  bool ok = n <= 0 || fread(&(*out)[0], 1, n, f) == (size_t)n;
  fclose(f);
  return ok;
}
struct Mapped {
  void* p = nullptr; size_t n = 0;
  ~Mapped() { if (p) munmap(p, n); }
  int open(const std::string& path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) BZ_FAIL(BZ_E_NOTFOUND, "cannot open '%s'", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); BZ_FAIL(BZ_E_INVALID, "cannot stat '%s' (not a regular file?)", path.c_str()); }
    n = (size_t)st.st_size;
    p = n ? mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    ::close(fd);
    if (n && p == MAP_FAILED) { p = nullptr; BZ_FAIL(BZ_E_INVALID, "mmap of '%s' failed", path.c_str()); }
    return BZ_OK;
  }
};

// loader/detect.rs:103-113
std::string find_config_in_dir(const std::string& dir) {
  for (const char* nm : {"config.json", "config.yaml", "config.yml"}) if (is_file(dir + "/" + nm)) return dir + "/" + nm;
  return "";
}

// ---------------------------------------------------------------------------------------------------------
// SafeTensors: 8-byte little-endian header length, JSON header {name: {dtype, shape, data_offsets}}, raw data
// ---------------------------------------------------------------------------------------------------------
// overflow-checked size arithmetic for header-controlled values
bool mul_ok(size_t a, size_t b, size_t* out) { return !__builtin_mul_overflow(a, b, out); }
bool shape_elems(const std::vector<int64_t>& shape, size_t* out) {
  size_t n = 1;
  for (int64_t d : shape) { if (d <= 0 || !mul_ok(n, (size_t)d, &n)) return false; }
  *out = n;
  return true;
}
size_t st_dtype_size(const std::string& d) {
  if (d == "F64" || d == "I64" || d == "U64") return 8;
  if (d == "F32" || d == "I32" || d == "U32") return 4;
  if (d == "F16" || d == "BF16" || d == "I16" || d == "U16") return 2;
  if (d == "I8" || d == "U8" || d == "BOOL" || d == "F8_E4M3" || d == "F8_E5M2") return 1;
  return 0;
}

struct StTensor { std::string dtype; std::vector<int64_t> shape; const unsigned char* data = nullptr; size_t bytes = 0; };
struct StLoader {
  std::vector<std::unique_ptr<Mapped>> files;
  std::map<std::string, StTensor> tensors;      // BTreeMap order == tensor_names() order of a sorted listing
  size_t total = 0;
  bool sharded() const { return files.size() > 1; }
  int add_file(const std::string& path) {
    std::unique_ptr<Mapped> mf(new Mapped());
    BZ_TRY(mf->open(path));
    if (mf->n < 8) BZ_FAIL(BZ_E_INVALID, "'%s' is not a SafeTensors file (too small)", path.c_str());
    uint64_t hl; memcpy(&hl, mf->p, 8);
    if (hl > mf->n - 8) BZ_FAIL(BZ_E_INVALID, "'%s': SafeTensors header length %llu exceeds the file", path.c_str(), (unsigned long long)hl);
    JVal hdr;
    if (!json_parse((const char*)mf->p + 8, (size_t)hl, &hdr) || hdr.t != JVal::OBJ) BZ_FAIL(BZ_E_INVALID, "'%s': malformed SafeTensors header", path.c_str());
    const unsigned char* base = (const unsigned char*)mf->p + 8 + hl;
    const size_t data_n = mf->n - 8 - (size_t)hl;
    for (auto& kv : hdr.o) {
      if (kv.first == "__metadata__") continue;
      const JVal& t = kv.second;
      const JVal* sh = t.get("shape"); const JVal* off = t.get("data_offsets");
      if (!sh || sh->t != JVal::ARR || !off || off->t != JVal::ARR || off->a.size() != 2) BZ_FAIL(BZ_E_INVALID, "'%s': bad entry for tensor '%s'", path.c_str(), kv.first.c_str());
      StTensor st; st.dtype = t.str("dtype");
      for (auto& d : sh->a) {
        if (d.t != JVal::NUM || !(d.n >= 0) || d.n > 9.0e15 || d.n != (double)(int64_t)d.n) BZ_FAIL(BZ_E_INVALID, "'%s': tensor '%s' has a bad dimension", path.c_str(), kv.first.c_str());
        st.shape.push_back((int64_t)d.n);
      }
      if (off->a[0].t != JVal::NUM || off->a[1].t != JVal::NUM || !(off->a[0].n >= 0) || !(off->a[1].n >= 0) || off->a[0].n > 9.0e15 || off->a[1].n > 9.0e15)
        BZ_FAIL(BZ_E_INVALID, "'%s': tensor '%s' data_offsets malformed", path.c_str(), kv.first.c_str());
      const size_t b = (size_t)off->a[0].n, e = (size_t)off->a[1].n;
      if (b > e || e > data_n) BZ_FAIL(BZ_E_INVALID, "'%s': tensor '%s' data_offsets out of range", path.c_str(), kv.first.c_str());
      st.data = base + b; st.bytes = e - b;
      // the safetensors crate the reference uses rejects a header whose byte range disagrees with dtype x shape (TensorInvalidInfo)
      size_t ne = 0, want = 0;
      const size_t es = st_dtype_size(st.dtype);
      bool empty_dim = false;
      for (int64_t d : st.shape) if (d == 0) empty_dim = true;
      if (empty_dim) { if (st.bytes != 0) BZ_FAIL(BZ_E_INVALID, "'%s': tensor '%s' has an empty shape but %zu bytes", path.c_str(), kv.first.c_str(), st.bytes); }
      else if (es && (!shape_elems(st.shape, &ne) || !mul_ok(ne, es, &want) || want != st.bytes))
        BZ_FAIL(BZ_E_INVALID, "'%s': tensor '%s' (%s): %zu bytes in the file but shape x dtype needs %zu", path.c_str(), kv.first.c_str(), st.dtype.c_str(), st.bytes, want);
      tensors[kv.first] = st;
    }
    total += mf->n;
    files.push_back(std::move(mf));
    return BZ_OK;
  }
  // a single file, the first shard of a sharded model, or a directory
  int open(const std::string& path) {
    std::string dir = is_dir(path) ? path : dirname_of(path);
    std::vector<std::string> shard_files;
    if (is_file(path) && basename_of(path).find("-of-") == std::string::npos) { shard_files.push_back(path); }
    else {
      std::string idx = dir + "/model.safetensors.index.json", text;
      JVal j;
      if (is_dir(path)) for (const char* nm : {"model.safetensors", "pytorch_model.safetensors"}) if (shard_files.empty() && is_file(dir + "/" + nm)) shard_files.push_back(dir + "/" + nm);
      if (shard_files.empty() && read_text(idx, &text) && json_parse(text.data(), text.size(), &j) && j.get("weight_map")) {
        std::set<std::string> uniq;
        for (auto& kv : j.get("weight_map")->o) uniq.insert(kv.second.s);
        for (auto& f : uniq) shard_files.push_back(dir + "/" + f);
      }
      if (shard_files.empty()) shard_files = glob_list(dir + "/model-*-of-*.safetensors");
      if (shard_files.empty() && is_file(path)) shard_files.push_back(path);
    }
    if (shard_files.empty()) BZ_FAIL(BZ_E_NOTFOUND, "no SafeTensors files under '%s'", path.c_str());
    for (auto& f : shard_files) BZ_TRY(add_file(f));
    return BZ_OK;
  }
  const StTensor* find(const std::string& n) const { auto it = tensors.find(n); return it == tensors.end() ? nullptr : &it->second; }
};
int st_dtype(const std::string& d) { return d == "F32" ? BZ_F32 : d == "F16" ? BZ_F16 : d == "BF16" ? BZ_BF16 : d == "I32" ? BZ_I32 : d == "I64" ? BZ_I64 : -1; }

// ---------------------------------------------------------------------------------------------------------
// config.json (HF) -> bz_model_config
// ---------------------------------------------------------------------------------------------------------
struct QuantInfo { int method = 0; int group_size = 128; int torch_dtype = BZ_F16; bool has_dtype = false; std::string model_type; };

void config_defaults(bz_model_config* c) {
  memset(c, 0, sizeof(*c));
  c->abi_version = BZ_ABI_VERSION; c->arch = BZ_ARCH_LLAMA;
  c->rms_eps = 1e-5f;              // model/config.rs:119-121
  c->rope_theta = 10000.0f;        // :123-125
  c->max_seq_len = 4096;           // :143-145
  c->rope_scaling = BZ_ROPE_NONE; c->rope_factor = 1.0f; c->rope_low_freq_factor = 1.0f; c->rope_high_freq_factor = 4.0f; c->rope_original_max_pos = 8192;   // :127-141
  c->act_dtype = BZ_F16;           // safetensors/config.rs:27 default dtype
  c->ssm_n_groups = 1; c->ssm_conv_kernel = 4;
  c->moe_routed_scale = 1.0f;
}

int hf_config_to_pod(const JVal& j, bz_model_config* c, QuantInfo* q) {
  if (j.t != JVal::OBJ) BZ_FAIL(BZ_E_INVALID, "config.json: not a JSON object");
  config_defaults(c);
  std::string mt = j.str("model_type");
  if (mt.empty()) { const JVal* a = j.get("architectures"); if (a && a->t == JVal::ARR && !a->a.empty()) mt = a->a[0].s; }
  std::string lower = mt; std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
  q->model_type = lower;
  const std::string td = j.str("torch_dtype");       // safetensors/config.rs:15-28
  if (td == "bfloat16") { q->torch_dtype = BZ_BF16; q->has_dtype = true; } else if (td == "float16") { q->torch_dtype = BZ_F16; q->has_dtype = true; }
  else if (td == "float32") { q->torch_dtype = BZ_F32; q->has_dtype = true; }
  c->act_dtype = q->torch_dtype;
  if (const JVal* qc = j.get("quantization_config")) {   // detect_arch.rs:79-90,118-131
    std::string m = qc->str("quant_method"); std::transform(m.begin(), m.end(), m.begin(), ::tolower);
    if (m == "awq") q->method = 1; else if (m == "gptq") q->method = 2;
    q->group_size = (int)qc->i64("group_size", 128);
    if (q->method && qc->i64("bits", 4) != 4) BZ_FAIL(BZ_E_UNSUPPORTED, "config.json: %lld-bit %s is not implemented (4-bit only)", qc->i64("bits", 4), m.c_str());
  }
  c->vocab = (int)j.i64("vocab_size", 0);
  c->hidden = (int)j.i64("hidden_size", 0);
  c->n_layers = (int)j.i64("num_hidden_layers", 0);
  c->tie_embeddings = j.boolean("tie_word_embeddings", false) ? 1 : 0;
  if (lower.find("mamba") != std::string::npos) {
    c->arch = BZ_ARCH_MAMBA2;
    c->ssm_d_state = (int)j.i64("state_size", 64);
    c->ssm_head_dim = (int)j.i64("head_dim", 64);
    const int expand = (int)j.i64("expand", 2);
    c->ssm_d_inner = expand * c->hidden;
    c->ssm_n_heads = (int)j.i64("num_heads", c->ssm_head_dim ? c->ssm_d_inner / c->ssm_head_dim : 0);
    c->ssm_n_groups = (int)j.i64("n_groups", 1);
    c->ssm_conv_kernel = (int)j.i64("conv_kernel", 4);
    c->rms_eps = (float)j.f64("layer_norm_epsilon", j.f64("rms_norm_eps", 1e-5));
    c->max_seq_len = (int)j.i64("max_position_embeddings", 1 << 20);
    return BZ_OK;
  }
  c->n_heads = (int)j.i64("num_attention_heads", 32);
  c->n_kv_heads = (int)j.i64("num_key_value_heads", c->n_heads);
  c->head_dim = (int)j.i64("head_dim", c->n_heads ? c->hidden / c->n_heads : 0);
  c->inter = (int)j.i64("intermediate_size", 0);
  c->max_seq_len = (int)j.i64("max_position_embeddings", 4096);
  c->rms_eps = (float)j.f64("rms_norm_eps", 1e-5);
  c->rope_theta = (float)j.f64("rope_theta", 10000.0);
  if (const JVal* rs = j.get("rope_scaling")) {
    if (rs->t == JVal::OBJ) {   // safetensors/config.rs:83-95
      std::string ty = rs->str("rope_type", rs->str("type", "llama3").c_str());
      c->rope_factor = (float)rs->f64("factor", 1.0);
      c->rope_low_freq_factor = (float)rs->f64("low_freq_factor", 1.0);
      c->rope_high_freq_factor = (float)rs->f64("high_freq_factor", 4.0);
      c->rope_original_max_pos = (int)rs->i64("original_max_position_embeddings", 8192);
      if (ty == "llama3") c->rope_scaling = BZ_ROPE_LLAMA3;
      else if (ty == "linear") c->rope_scaling = BZ_ROPE_LINEAR;
      else if (ty == "yarn") {
        // HF _compute_yarn_parameters / DeepseekV2YarnRotaryEmbedding: beta_fast / beta_slow (0 = defaults 32 / 1), the cos / sin factor, and -- DeepSeek-V2 --
        // the softmax mscale from mscale_all_dim
        c->rope_scaling = BZ_ROPE_YARN;
        c->rope_beta_fast = (float)rs->f64("beta_fast", 0.0); c->rope_beta_slow = (float)rs->f64("beta_slow", 0.0);
        auto mscale = [&](double ms) { return c->rope_factor <= 1.0f ? 1.0 : 0.1 * ms * log((double)c->rope_factor) + 1.0; };
        const double ms = rs->f64("mscale", 0.0), msa = rs->f64("mscale_all_dim", 0.0);
        if (rs->get("attention_factor")) c->rope_attn_factor = (float)rs->f64("attention_factor", 0.0);
        else if (ms > 0.0 && msa > 0.0) c->rope_attn_factor = (float)(mscale(ms) / mscale(msa));
        if (lower.find("deepseek") != std::string::npos && msa > 0.0) c->mla_softmax_mscale = (float)mscale(msa);
      }
      else if (ty == "default" || ty.empty()) c->rope_scaling = BZ_ROPE_NONE;
      else BZ_FAIL(BZ_E_UNSUPPORTED, "config.json: rope_scaling type '%s' is not implemented", ty.c_str());
    }
  }
  if (lower.find("deepseek") != std::string::npos) {
    c->arch = BZ_ARCH_DEEPSEEK2;
    c->mla_kv_lora_rank = (int)j.i64("kv_lora_rank", 0);
    c->mla_q_lora_rank = (int)j.i64("q_lora_rank", 0);      // null -> 0
    c->mla_nope_dim = (int)j.i64("qk_nope_head_dim", 0);
    c->mla_rope_dim = (int)j.i64("qk_rope_head_dim", 0);
    c->mla_v_dim = (int)j.i64("v_head_dim", 0);
    c->moe_n_experts = (int)j.i64("n_routed_experts", 0);
    c->moe_top_k = (int)j.i64("num_experts_per_tok", 2);     // model/config.rs:151-153 default 2
    c->moe_n_shared = (int)j.i64("n_shared_experts", 0);
    c->moe_inter = (int)j.i64("moe_intermediate_size", 0);
    c->moe_first_dense = (int)j.i64("first_k_dense_replace", 0);
    c->moe_norm_topk = j.boolean("norm_topk_prob", false) ? 1 : 0;
    c->moe_routed_scale = (float)j.f64("routed_scaling_factor", 1.0);
    c->rope_interleaved = 1;
    c->rms_eps = (float)j.f64("rms_norm_eps", 1e-6);
  }
  return BZ_OK;
}

// detect_arch.rs:66-196: quant_config.json / quantize_config.json / config.json quantization_config / tensor-name suffixes
void detect_quant(const std::string& dir, const StLoader& st, QuantInfo* q) {
  auto from_file = [&](const char* nm, int* method, int* gs) {
    std::string text; JVal j;
    if (!read_text(dir + "/" + nm, &text) || !json_parse(text.data(), text.size(), &j)) return;
    std::string m = j.str("quant_method"); std::transform(m.begin(), m.end(), m.begin(), ::tolower);
    if (m == "awq") *method = 1; else if (m == "gptq") *method = 2;
    double g; if (j.num("group_size", &g)) *gs = (int)g;
  };
  int m = 0, gs = q->group_size;
  from_file("quantize_config.json", &m, &gs);
  if (!m) from_file("quant_config.json", &m, &gs);
  if (m) { q->method = m; q->group_size = gs; return; }
  if (q->method) return;    // config.json quantization_config
  bool has_qw = false, has_gidx = false;
  for (auto& kv : st.tensors) { if (ends_with(kv.first, ".qweight")) has_qw = true; if (ends_with(kv.first, ".g_idx")) has_gidx = true; }
  if (has_gidx) q->method = 2; else if (has_qw) q->method = 1;    // regular.rs:42-49: GPTQ is probed first
}

// boostr::model::detection::detect_architecture_from_names, as pinned by the reference's tests (loader/safetensors/detect_arch.rs:200-315):
// format from the "model." prefix, layer count from the highest "layers.N." index, tied embeddings when lm_head.weight is absent, per-layer type
// from the tensor names of that layer (mamba mixer tensors -> Mamba2; MLA latent projections -> MlaWithMoe / MlaWithMlp; else StandardTransformer).
int detect_from_names(const std::vector<std::string>& names, bz_detected_arch* out) {
  memset(out, 0, sizeof(*out));
  bool hf = false, lm_head = false;
  int layers = 0;
  for (auto& n : names) {
    if (n.compare(0, 6, "model.") == 0 || n.compare(0, 9, "backbone.") == 0) hf = true;
    if (n == "lm_head.weight" || n == "lm_head.qweight") lm_head = true;
    size_t p = n.find("layers.");
    if (p != std::string::npos && isdigit((unsigned char)n[p + 7])) layers = std::max(layers, atoi(n.c_str() + p + 7) + 1);
  }
  if (layers == 0) BZ_FAIL(BZ_E_INVALID, "no transformer / mamba layers found in the tensor names");
  out->format = hf ? 0 : 1;
  out->num_layers = layers;
  out->tie_word_embeddings = lm_head ? 0 : 1;
  for (int l = 0; l < layers && l < 512; l++) {
    const std::string key = "layers." + std::to_string(l) + ".";
    bool mamba = false, mamba3 = false, mla = false, moe = false;
    for (auto& n : names) {
      size_t p = n.find(key);
      if (p == std::string::npos) continue;
      const std::string rest = n.substr(p + key.size());
      if (rest.find("mamba3") != std::string::npos) mamba3 = true;
      if (rest.find("mixer.") != std::string::npos || rest.find("mamba2") != std::string::npos || rest.find("A_log") != std::string::npos) mamba = true;
      if (rest.find("w_dkv") != std::string::npos || rest.find("kv_a_proj") != std::string::npos || rest.find("kv_b_proj") != std::string::npos) mla = true;
      if (rest.find("experts.") != std::string::npos || rest.find("moe.") != std::string::npos) moe = true;
    }
    out->layer_types[l] = mamba3 ? BZ_LAYER_MAMBA3 : mamba ? BZ_LAYER_MAMBA2 : (mla ? (moe ? BZ_LAYER_MLA_MOE : BZ_LAYER_MLA_MLP) : BZ_LAYER_TRANSFORMER);
  }
  return BZ_OK;
}

// detect_arch.rs:13-63 (+ boostr's name scan for the layer count / tied embeddings): used when no config.json is present
int detect_from_tensors(const StLoader& st, bz_model_config* c) {
  config_defaults(c);
  std::vector<std::string> names;
  for (auto& kv : st.tensors) names.push_back(kv.first);
  bz_detected_arch da;
  BZ_TRY(detect_from_names(names, &da));
  for (int l = 0; l < da.num_layers && l < 512; l++)
    if (da.layer_types[l] != BZ_LAYER_TRANSFORMER) BZ_FAIL(BZ_E_UNSUPPORTED, "layer %d is not a standard transformer layer: a config.json is required for this architecture", l);
  c->n_layers = da.num_layers;
  auto rows_of = [&](const std::string& base, int64_t* rows) {
    if (const StTensor* t = st.find(base + ".weight")) { if (t->shape.size() == 2) { *rows = t->shape[0]; return true; } }
    if (const StTensor* t = st.find(base + ".qweight")) {
      if (t->shape.size() == 2) { *rows = st.find(base + ".g_idx") ? t->shape[1] : t->shape[1] * 8; return true; }   // GPTQ [K/8,N] / AWQ [K,N/8]
    }
    return false;
  };
  if (const StTensor* e = st.find("model.embed_tokens.weight")) if (e->shape.size() == 2) { c->vocab = (int)e->shape[0]; c->hidden = (int)e->shape[1]; }
  int64_t r;
  if (rows_of("model.layers.0.mlp.gate_proj", &r)) c->inter = (int)r;
  const int head_dim = 128;   // detect_arch.rs:42 "Default head dimension for most models"
  if (rows_of("model.layers.0.self_attn.q_proj", &r) && c->hidden > 0) { c->n_heads = (int)(r / head_dim); c->head_dim = head_dim; }
  if (rows_of("model.layers.0.self_attn.k_proj", &r)) c->n_kv_heads = (int)(r / (c->head_dim ? c->head_dim : 128));
  c->tie_embeddings = st.find("lm_head.weight") ? 0 : 1;
  if (!c->vocab || !c->hidden || !c->n_layers || !c->n_heads) BZ_FAIL(BZ_E_INVALID, "cannot detect the architecture from tensor names / shapes (no config.json)");
  if (!c->n_kv_heads) c->n_kv_heads = c->n_heads;
  return BZ_OK;
}

float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
uint16_t bf16_to_f16_bits(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return __half_as_ushort(__float2half(f)); }

int load_safetensors(bz_device* dev, const std::string& path, bz_model** out, bz_model_config* cfg_out) {
  StLoader st;
  BZ_TRY(st.open(path));
  const std::string dir = is_dir(path) ? path : dirname_of(path);
  bz_model_config c; QuantInfo q;
  std::string text; JVal j;
  const std::string cfg_path = find_config_in_dir(dir);
  if (!cfg_path.empty() && ends_with(cfg_path, ".json") && read_text(cfg_path, &text) && json_parse(text.data(), text.size(), &j)) BZ_TRY(hf_config_to_pod(j, &c, &q));
  else BZ_TRY(detect_from_tensors(st, &c));
  detect_quant(dir, st, &q);
  if (q.method) c.act_dtype = BZ_F16;   // awq.rs:69-71 / gptq.rs: AWQ and GPTQ models always run in f16
  if (!c.tie_embeddings && c.arch != BZ_ARCH_MAMBA2 && !st.find("lm_head.weight") && !st.find("lm_head.qweight")) c.tie_embeddings = 1;
  bz_model* m = nullptr;
  BZ_TRY(bz_model_create(dev, &c, &m));
  int rc = BZ_OK;
  std::vector<uint16_t> tmp16; std::vector<float> f32a, f32b;
  for (auto& kv : st.tensors) {
    const std::string& name = kv.first; const StTensor& t = kv.second;
    if (rc != BZ_OK) break;
    if (ends_with(name, ".qzeros") || ends_with(name, ".scales") || ends_with(name, ".g_idx")) continue;   // awq.rs:150-157 / gptq.rs:150-168
    if (ends_with(name, "rotary_emb.inv_freq")) continue;
    if (ends_with(name, ".qweight")) {
      const std::string base = name.substr(0, name.size() - 8);
      const StTensor* sc = st.find(base + ".scales"); const StTensor* qz = st.find(base + ".qzeros");
      if (!sc || !qz || t.shape.size() != 2 || sc->shape.size() != 2 || qz->shape.size() != 2 || sc->dtype != "F16")
        { rc = BZ_E_INVALID; bz_set_error("quantised layer '%s': qweight / scales (F16) / qzeros triplet incomplete", base.c_str()); break; }
      {   // every byte count the add_* calls will read, checked against the file (a lying triplet must fail here, not read out of bounds)
        const bool gptq = q.method == 2;
        const int64_t K = gptq ? t.shape[0] * 8 : t.shape[0], N = gptq ? t.shape[1] : t.shape[1] * 8;
        const int gs = q.group_size;
        const StTensor* gi = st.find(base + ".g_idx"); const StTensor* bi = st.find(base + ".bias");
        bool bad = t.dtype != "I32" && t.dtype != "U32";
        bad = bad || gs <= 0 || K <= 0 || N <= 0 || K % gs != 0 || N % 8 != 0 || K > (1 << 24) || N > (1 << 24);
        const int64_t G = bad ? 0 : K / gs;
        bad = bad || sc->shape[0] != G || sc->shape[1] != N || sc->bytes != (size_t)G * N * 2;
        bad = bad || (qz->dtype != "I32" && qz->dtype != "U32") || qz->shape[0] != G || qz->shape[1] != N / 8 || qz->bytes != (size_t)G * (N / 8) * 4;
        bad = bad || t.bytes != (size_t)K * N / 2;
        if (gptq && gi) bad = bad || gi->dtype != "I32" || gi->shape.size() != 1 || gi->shape[0] != K || gi->bytes != (size_t)K * 4;
        if (gptq && bi) bad = bad || bi->shape.size() != 1 || bi->shape[0] != N || (bi->dtype != "F32" && bi->dtype != "F16") || bi->bytes != (size_t)N * (bi->dtype == "F32" ? 4 : 2);
        if (bad) { rc = BZ_E_INVALID; bz_set_error("quantised layer '%s': qweight / scales / qzeros / g_idx / bias shapes or byte counts are inconsistent (group size %d)", base.c_str(), gs); break; }
      }
      f32a.resize((size_t)sc->shape[0] * sc->shape[1]);
      for (size_t i = 0; i < f32a.size(); i++) f32a[i] = f16_to_f32(((const uint16_t*)sc->data)[i]);   // awq.rs:202-206 cast_f16_bytes_to_f32
      if (q.method == 2) {
        const int64_t K = t.shape[0] * 8, N = t.shape[1];          // gptq.rs:198-247
        const StTensor* gi = st.find(base + ".g_idx"); const StTensor* bi = st.find(base + ".bias");
        std::vector<float> bias;
        if (bi) { bias.resize((size_t)N); for (int64_t i = 0; i < N; i++) bias[i] = bi->dtype == "F32" ? ((const float*)bi->data)[i] : f16_to_f32(((const uint16_t*)bi->data)[i]); }
        rc = bz_model_add_gptq(m, (base + ".weight").c_str(), N, K, (const uint32_t*)t.data, f32a.data(), (const uint32_t*)qz->data,
                               gi ? (const int32_t*)gi->data : nullptr, bi ? bias.data() : nullptr, q.group_size);
      } else {
        const int64_t K = t.shape[0], N = t.shape[1] * 8, G = qz->shape[0];   // awq.rs:190-225
        static const int SH[8] = {0, 16, 4, 20, 8, 24, 12, 28};                // awq.rs:32
        f32b.resize((size_t)G * N);
        const uint32_t* pk = (const uint32_t*)qz->data;
        for (int64_t g = 0; g < G; g++)
          for (int64_t jx = 0; jx < N / 8; jx++) { const uint32_t v = pk[g * (N / 8) + jx]; for (int k = 0; k < 8; k++) f32b[g * N + jx * 8 + k] = (float)((v >> SH[k]) & 0xF); }
        rc = bz_model_add_awq(m, (base + ".weight").c_str(), N, K, (const uint32_t*)t.data, f32a.data(), f32b.data(), q.group_size);
      }
      continue;
    }
    if (ends_with(name, ".bias") && st.find(name.substr(0, name.size() - 5) + ".qweight")) continue;   // gptq.rs:158-166: handled with its layer
    const int dt = st_dtype(t.dtype);
    if (dt != BZ_F32 && dt != BZ_F16 && dt != BZ_BF16) { rc = BZ_E_UNSUPPORTED; bz_set_error("tensor '%s': dtype %s is not supported", name.c_str(), t.dtype.c_str()); break; }
    if (t.shape.empty() || t.shape.size() > 3) { rc = BZ_E_UNSUPPORTED; bz_set_error("tensor '%s': rank %zu is not supported", name.c_str(), t.shape.size()); break; }
    const void* data = t.data; int use_dt = dt;
    if (q.method && dt == BZ_BF16) {     // awq.rs:93-103: BF16 tensors of an AWQ / GPTQ checkpoint are cast to F16
      const size_t ne = t.bytes / 2;
      tmp16.resize(ne);
      for (size_t i = 0; i < ne; i++) tmp16[i] = bf16_to_f16_bits(((const uint16_t*)t.data)[i]);
      data = tmp16.data(); use_dt = BZ_F16;
    }
    rc = bz_model_add_dense(m, name.c_str(), use_dt, t.shape.data(), (int)t.shape.size(), data);
  }
  if (rc == BZ_OK) rc = bz_model_finalize(m);
  if (rc != BZ_OK) { bz_model_free(m); return rc; }
  if (cfg_out) bz_model_get_config(m, cfg_out);
  *out = m;
  return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// GGUF container (v2 / v3): header, metadata key-values, tensor infos, aligned data section
// ---------------------------------------------------------------------------------------------------------
struct GgufVal { int type = -1; uint64_t u = 0; double f = 0; std::string s; size_t arr_n = 0; };
struct GgufTensor { std::string name; std::vector<int64_t> ne; int type = 0; uint64_t offset = 0; };
struct Gguf {
  Mapped file; std::map<std::string, GgufVal> kv; std::vector<GgufTensor> tensors; size_t data_off = 0; uint32_t version = 0;
  const unsigned char* p = nullptr; const unsigned char* e = nullptr; bool ok = true;
  template <class T> T rd() { T v{}; if ((size_t)(e - p) < sizeof(T)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
  std::string rstr() { uint64_t n = rd<uint64_t>(); if (!ok || (uint64_t)(e - p) < n) { ok = false; return ""; } std::string s((const char*)p, (size_t)n); p += n; return s; }
  bool scalar(int ty, GgufVal* v) {
    switch (ty) {
      case 0: v->u = rd<uint8_t>(); v->f = (double)v->u; break; case 1: { int8_t x = rd<int8_t>(); v->u = (uint64_t)(int64_t)x; v->f = x; break; }
      case 2: v->u = rd<uint16_t>(); v->f = (double)v->u; break; case 3: { int16_t x = rd<int16_t>(); v->u = (uint64_t)(int64_t)x; v->f = x; break; }
      case 4: v->u = rd<uint32_t>(); v->f = (double)v->u; break; case 5: { int32_t x = rd<int32_t>(); v->u = (uint64_t)(int64_t)x; v->f = x; break; }
      case 6: { float x = rd<float>(); v->f = x; v->u = (uint64_t)x; break; }
      case 7: v->u = rd<uint8_t>() != 0; v->f = (double)v->u; break;
      case 8: v->s = rstr(); break;
      case 10: v->u = rd<uint64_t>(); v->f = (double)v->u; break; case 11: { int64_t x = rd<int64_t>(); v->u = (uint64_t)x; v->f = (double)x; break; }
      case 12: v->f = rd<double>(); v->u = (uint64_t)v->f; break;
      default: return false;
    }
    return ok;
  }
  int open(const std::string& path) {
    BZ_TRY(file.open(path));
    p = (const unsigned char*)file.p; e = p + file.n;
    if (file.n < 24 || memcmp(p, "GGUF", 4)) BZ_FAIL(BZ_E_INVALID, "'%s' is not a GGUF file", path.c_str());
    p += 4;
    version = rd<uint32_t>();
    if (version < 2 || version > 3) BZ_FAIL(BZ_E_UNSUPPORTED, "GGUF version %u is not supported", version);
    const uint64_t nt = rd<uint64_t>(), nkv = rd<uint64_t>();
    for (uint64_t i = 0; i < nkv && ok; i++) {
      std::string key = rstr();
      GgufVal v; v.type = (int)rd<uint32_t>();
      if (v.type == 9) {
        const int et = (int)rd<uint32_t>(); v.arr_n = (size_t)rd<uint64_t>();
        for (size_t k = 0; k < v.arr_n && ok; k++) { GgufVal tmp; if (et == 9 || !scalar(et, &tmp)) ok = false; }
      } else if (!scalar(v.type, &v)) ok = false;
      kv[key] = v;
    }
    for (uint64_t i = 0; i < nt && ok; i++) {
      GgufTensor t; t.name = rstr();
      const uint32_t nd = rd<uint32_t>();
      if (nd > 4) { ok = false; break; }
      for (uint32_t d = 0; d < nd; d++) {
        const uint64_t x = rd<uint64_t>();
        if (x == 0 || x > (1ull << 40)) { ok = false; break; }      // negative (as i64) / absurd extents
        t.ne.push_back((int64_t)x);
      }
      t.type = (int)rd<uint32_t>(); t.offset = rd<uint64_t>();
      tensors.push_back(t);
    }
    if (!ok) BZ_FAIL(BZ_E_INVALID, "'%s': truncated or malformed GGUF header", path.c_str());
    size_t align = 32;
    auto it = kv.find("general.alignment");
    if (it != kv.end()) {
      const uint64_t a = it->second.u;
      if (a == 0 || a > 4096 || (a & (a - 1))) BZ_FAIL(BZ_E_INVALID, "'%s': general.alignment %llu is not a power of two <= 4096", path.c_str(), (unsigned long long)a);
      align = (size_t)a;
    }
    data_off = (size_t)(p - (const unsigned char*)file.p);
    data_off = (data_off + align - 1) / align * align;
    if (!tensors.empty() && data_off > file.n) BZ_FAIL(BZ_E_INVALID, "'%s': truncated GGUF file (no data section)", path.c_str());
    return BZ_OK;
  }
  bool u32(const std::string& k, long long* out) const { auto it = kv.find(k); if (it == kv.end() || it->second.type == 8 || it->second.type == 9) return false; *out = (long long)it->second.u; return true; }
  bool f32(const std::string& k, double* out) const { auto it = kv.find(k); if (it == kv.end() || it->second.type == 8 || it->second.type == 9) return false; *out = it->second.f; return true; }
  std::string str(const std::string& k, const char* d) const { auto it = kv.find(k); return (it != kv.end() && it->second.type == 8) ? it->second.s : std::string(d); }
};

// loader/gguf.rs:101-306 config_from_gguf_metadata
int gguf_config(const Gguf& g, bz_model_config* c, std::string* arch_out) {
  config_defaults(c);
  const std::string arch = g.str("general.architecture", "llama");
  *arch_out = arch;
  long long v; double f;
  if (g.u32("general.vocab_size", &v)) c->vocab = (int)v;
  else { auto it = g.kv.find("tokenizer.ggml.tokens"); c->vocab = (it != g.kv.end() && it->second.type == 9) ? (int)it->second.arr_n : ((arch == "llama" || arch == "llama2" || arch == "llama3") ? 128256 : 32000); }
  if (!g.u32(arch + ".embedding_length", &v)) BZ_FAIL(BZ_E_INVALID, "GGUF missing %s.embedding_length", arch.c_str());
  c->hidden = (int)v;
  if (!g.u32(arch + ".block_count", &v)) BZ_FAIL(BZ_E_INVALID, "GGUF missing %s.block_count", arch.c_str());
  c->n_layers = (int)v;
  c->max_seq_len = g.u32(arch + ".context_length", &v) ? (int)v : 4096;
  c->inter = g.u32(arch + ".feed_forward_length", &v) ? (int)v : 0;
  c->rms_eps = g.f32(arch + ".attention.layer_norm_rms_epsilon", &f) ? (float)f : 1e-5f;
  c->act_dtype = BZ_F32;                     // gguf.rs:305
  c->rope_interleaved = 1;                   // GGML "NORM" rope: pairs (2i, 2i+1); llama.cpp's converter permutes q/k rows for it
  const bool is_ssm = arch == "mamba" || arch == "mamba2" || arch == "mamba3";
  if (is_ssm) {
    c->arch = BZ_ARCH_MAMBA2;
    c->ssm_d_state = g.u32(arch + ".ssm.state_size", &v) ? (int)v : 64;
    c->ssm_conv_kernel = g.u32(arch + ".ssm.conv_kernel", &v) ? (int)v : 4;
    c->ssm_d_inner = g.u32(arch + ".ssm.inner_size", &v) ? (int)v : c->hidden * 2;
    c->ssm_head_dim = g.u32(arch + ".ssm.head_dim", &v) ? (int)v : 64;
    c->ssm_n_heads = c->ssm_d_inner / c->ssm_head_dim;
    c->ssm_n_groups = g.u32(arch + ".ssm.group_count", &v) ? (int)v : 1;
    return BZ_OK;
  }
  c->n_heads = g.u32(arch + ".attention.head_count", &v) ? (int)v : 32;
  c->n_kv_heads = g.u32(arch + ".attention.head_count_kv", &v) ? (int)v : c->n_heads;
  c->head_dim = g.u32(arch + ".attention.key_length", &v) ? (int)v : (c->n_heads ? c->hidden / c->n_heads : 0);
  c->rope_theta = g.f32(arch + ".rope.freq_base", &f) ? (float)f : 10000.0f;
  if (g.u32(arch + ".attention.kv_lora_rank", &v)) {          // gguf.rs:188-196: MLA detection
    c->arch = BZ_ARCH_DEEPSEEK2; c->mla_kv_lora_rank = (int)v;
    c->mla_q_lora_rank = g.u32(arch + ".attention.q_lora_rank", &v) ? (int)v : 0;
    c->mla_rope_dim = g.u32(arch + ".attention.rope_dimension_count", &v) ? (int)v : 0;
  }
  if (g.u32(arch + ".expert_count", &v)) {                    // gguf.rs:271-283
    c->moe_n_experts = (int)v;
    c->moe_top_k = g.u32(arch + ".expert_used_count", &v) ? (int)v : 2;
  }
  return BZ_OK;
}

// llama.cpp tensor names -> the HF names bz_model_finalize expects
std::string gguf_to_hf_name(const std::string& n) {
  if (n == "token_embd.weight") return "model.embed_tokens.weight";
  if (n == "output_norm.weight") return "model.norm.weight";
  if (n == "output.weight") return "lm_head.weight";
  if (n.compare(0, 4, "blk.") != 0) return "";
  const size_t dot = n.find('.', 4);
  if (dot == std::string::npos) return "";
  const std::string idx = n.substr(4, dot - 4), rest = n.substr(dot + 1);
  static const std::pair<const char*, const char*> MAP[] = {
      {"attn_norm.weight", "input_layernorm.weight"}, {"ffn_norm.weight", "post_attention_layernorm.weight"}, {"attn_q.weight", "self_attn.q_proj.weight"},
      {"attn_k.weight", "self_attn.k_proj.weight"}, {"attn_v.weight", "self_attn.v_proj.weight"}, {"attn_output.weight", "self_attn.o_proj.weight"},
      {"ffn_gate.weight", "mlp.gate_proj.weight"}, {"ffn_up.weight", "mlp.up_proj.weight"}, {"ffn_down.weight", "mlp.down_proj.weight"}};
  for (auto& m : MAP) if (rest == m.first) return "model.layers." + idx + "." + m.second;
  return "";
}

// llama.cpp's mamba2 tensor names (the names the reference's loader maps live in the absent boostr crate; llama.cpp's converter is the convention GGUF files follow)
// -> the HF Mamba2 names bz_model_finalize expects.  ssm_a holds A = -exp(A_log) (the converter folds the exponential): it arrives as "mixer.A" and finalize
// turns it back into A_log = log(-A).
std::string gguf_to_hf_name_mamba2(const std::string& n) {
  if (n == "token_embd.weight") return "backbone.embeddings.weight";
  if (n == "output_norm.weight") return "backbone.norm_f.weight";
  if (n == "output.weight") return "lm_head.weight";
  if (n.compare(0, 4, "blk.") != 0) return "";
  const size_t dot = n.find('.', 4);
  if (dot == std::string::npos) return "";
  const std::string idx = n.substr(4, dot - 4), rest = n.substr(dot + 1);
  static const std::pair<const char*, const char*> MAP[] = {
      {"attn_norm.weight", "norm.weight"}, {"ssm_in.weight", "mixer.in_proj.weight"}, {"ssm_conv1d.weight", "mixer.conv1d.weight"}, {"ssm_conv1d.bias", "mixer.conv1d.bias"},
      {"ssm_dt.bias", "mixer.dt_bias"}, {"ssm_a", "mixer.A"}, {"ssm_d", "mixer.D"}, {"ssm_norm.weight", "mixer.norm.weight"}, {"ssm_out.weight", "mixer.out_proj.weight"}};
  for (auto& m : MAP) if (rest == m.first) return "backbone.layers." + idx + "." + m.second;
  return "";
}

size_t ggml_row_bytes(int type, int64_t K) {
  switch (type) { case 0: return (size_t)K * 4; case 1: case 30: return (size_t)K * 2; case 8: return (size_t)(K / 32) * 34; case 12: return (size_t)(K / 256) * 144;
                  case 14: return (size_t)(K / 256) * 210; default: return 0; }
}

int load_gguf(bz_device* dev, const std::string& path, bz_model** out, bz_model_config* cfg_out) {
  Gguf g;
  BZ_TRY(g.open(path));
  bz_model_config c; std::string arch;
  BZ_TRY(gguf_config(g, &c, &arch));
  if (c.arch != BZ_ARCH_LLAMA && c.arch != BZ_ARCH_MAMBA2)
    BZ_FAIL(BZ_E_UNSUPPORTED, "GGUF architecture '%s': the llama-family and mamba2 tensor namings are mapped in this build", arch.c_str());
  const bool mamba = c.arch == BZ_ARCH_MAMBA2;
  bool has_output = false;
  for (auto& t : g.tensors) if (t.name == "output.weight") has_output = true;
  c.tie_embeddings = has_output ? 0 : 1;
  bz_model* m = nullptr;
  BZ_TRY(bz_model_create(dev, &c, &m));
  int rc = BZ_OK;
  for (auto& t : g.tensors) {
    if (rc != BZ_OK) break;
    if (t.name == "rope_freqs.weight") continue;
    const std::string hf = mamba ? gguf_to_hf_name_mamba2(t.name) : gguf_to_hf_name(t.name);
    if (hf.empty()) { rc = BZ_E_UNSUPPORTED; bz_set_error("GGUF tensor '%s' has no mapping", t.name.c_str()); break; }
    if (t.ne.empty() || t.ne.size() > 2) { rc = BZ_E_UNSUPPORTED; bz_set_error("GGUF tensor '%s': rank %zu", t.name.c_str(), t.ne.size()); break; }
    const int64_t K = t.ne[0], N = t.ne.size() == 2 ? t.ne[1] : 1;      // ne[0] is the contiguous dimension
    const size_t rb = ggml_row_bytes(t.type, K);
    if (!rb) { rc = BZ_E_UNSUPPORTED; bz_set_error("GGUF tensor '%s': ggml type %d is not implemented (F32, F16, BF16, Q8_0, Q4_K, Q6_K)", t.name.c_str(), t.type); break; }
    {   // checked: offset / extents are file-controlled 64-bit values
      const size_t room = g.file.n - g.data_off;
      size_t need = 0;
      if (K <= 0 || N <= 0 || t.offset > room || !mul_ok(rb, (size_t)N, &need) || need > room - (size_t)t.offset)
        { rc = BZ_E_INVALID; bz_set_error("GGUF tensor '%s' runs past the end of the file", t.name.c_str()); break; }
    }
    const void* data = (const unsigned char*)g.file.p + g.data_off + t.offset;
    if (t.type == 0 || t.type == 1 || t.type == 30) {
      const int dt = t.type == 0 ? BZ_F32 : (t.type == 1 ? BZ_F16 : BZ_BF16);
      int64_t shape[2] = {N, K};
      // per-head / per-channel vectors that llama.cpp stores with a unit or group axis ([1, heads], [d_inner / groups, groups]) are vectors here
      const bool as_vec = mamba && t.ne.size() == 2 && hf.find("in_proj") == std::string::npos && hf.find("out_proj") == std::string::npos &&
                          hf.find("conv1d.weight") == std::string::npos && hf != "backbone.embeddings.weight" && hf != "lm_head.weight";
      const int64_t flat = N * K;
      rc = (t.ne.size() == 2 && !as_vec) ? bz_model_add_dense(m, hf.c_str(), dt, shape, 2, data) : bz_model_add_dense(m, hf.c_str(), dt, as_vec ? &flat : &K, 1, data);
    } else {
      rc = bz_model_add_gguf(m, hf.c_str(), t.type, N, K, data);
    }
  }
  if (rc == BZ_OK) rc = bz_model_finalize(m);
  if (rc != BZ_OK) { bz_model_free(m); return rc; }
  if (cfg_out) bz_model_get_config(m, cfg_out);
  *out = m;
  return BZ_OK;
}

std::string json_escape(const std::string& in) {
  std::string o;
  for (unsigned char ch : in) {
    if (ch == '"' || ch == '\\') { o += '\\'; o += (char)ch; }
    else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
    else o += (char)ch;
  }
  return o;
}

int copy_path(char* dst, size_t cap, const std::string& s) {
  if (s.size() + 1 > cap) BZ_FAIL(BZ_E_INVALID, "path too long");
  memcpy(dst, s.c_str(), s.size() + 1);
  return BZ_OK;
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int bz_detect_model_source(const char* path_c, bz_model_source* out) {
  BZ_API_BEGIN
  if (!path_c || !out) BZ_FAIL(BZ_E_INVALID, "detect_model_source: null argument");
  memset(out, 0, sizeof(*out));
  const std::string path = path_c;
  if (is_file(path)) {                                        // detect.rs:37-55
    const std::string ext = ext_of(path);
    if (ext == "safetensors") { out->format = BZ_FORMAT_SAFETENSORS; const std::string cfg = find_config_in_dir(dirname_of(path)); out->has_config = !cfg.empty(); BZ_TRY(copy_path(out->config_path, sizeof out->config_path, cfg)); }
    else if (ext == "gguf") { out->format = BZ_FORMAT_GGUF; }
    else BZ_FAIL(BZ_E_UNSUPPORTED, "Unsupported model file format: .%s", ext.c_str());
    return copy_path(out->weights_path, sizeof out->weights_path, path);
  }
  if (!is_dir(path)) BZ_FAIL(BZ_E_NOTFOUND, "Model path does not exist: %s", path.c_str());
  for (const char* nm : {"model.safetensors", "pytorch_model.safetensors"}) {   // detect.rs:64-77: SafeTensors preferred
    if (is_file(path + "/" + nm)) {
      out->format = BZ_FORMAT_SAFETENSORS;
      const std::string cfg = find_config_in_dir(path); out->has_config = !cfg.empty();
      BZ_TRY(copy_path(out->config_path, sizeof out->config_path, cfg));
      return copy_path(out->weights_path, sizeof out->weights_path, path + "/" + nm);
    }
  }
  std::vector<std::string> shards = glob_list(path + "/model-00001-of-*.safetensors");   // detect.rs:80-90
  if (!shards.empty()) {
    out->format = BZ_FORMAT_SAFETENSORS;
    const std::string cfg = find_config_in_dir(path); out->has_config = !cfg.empty();
    BZ_TRY(copy_path(out->config_path, sizeof out->config_path, cfg));
    return copy_path(out->weights_path, sizeof out->weights_path, shards[0]);
  }
  std::vector<std::string> ggufs = glob_list(path + "/*.gguf");                          // detect.rs:93-99
  if (!ggufs.empty()) { out->format = BZ_FORMAT_GGUF; return copy_path(out->weights_path, sizeof out->weights_path, ggufs[0]); }
  BZ_FAIL(BZ_E_NOTFOUND, "No supported model files found in directory: %s", path.c_str());
  BZ_API_END
}

extern "C" int bz_detect_architecture_from_names(const char* const* names, int n, bz_detected_arch* out) {
  BZ_API_BEGIN
  if (!names || n < 0 || !out) BZ_FAIL(BZ_E_INVALID, "detect_architecture_from_names: null argument");
  std::vector<std::string> v;
  for (int i = 0; i < n; i++) v.push_back(names[i] ? names[i] : "");
  return detect_from_names(v, out);
  BZ_API_END
}

extern "C" int bz_config_from_hf_json(const char* json_text, bz_model_config* cfg, bz_quant_info* qinfo) {
  BZ_API_BEGIN
  if (!json_text || !cfg) BZ_FAIL(BZ_E_INVALID, "config_from_hf_json: null argument");
  JVal j;
  if (!json_parse(json_text, strlen(json_text), &j)) BZ_FAIL(BZ_E_INVALID, "config.json: JSON syntax error");
  QuantInfo q;
  BZ_TRY(hf_config_to_pod(j, cfg, &q));
  if (q.method) cfg->act_dtype = BZ_F16;
  if (qinfo) { qinfo->quant_method = q.method; qinfo->group_size = q.group_size; qinfo->torch_dtype = q.has_dtype ? q.torch_dtype : -1; }
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_config_from_gguf(const char* path, bz_model_config* cfg, bz_gguf_info* info) {
  BZ_API_BEGIN
  if (!path || !cfg) BZ_FAIL(BZ_E_INVALID, "config_from_gguf: null argument");
  Gguf g;
  BZ_TRY(g.open(path));
  std::string arch;
  BZ_TRY(gguf_config(g, cfg, &arch));
  if (info) {                                                   // gguf.rs:309-346 get_gguf_info
    memset(info, 0, sizeof(*info));
    snprintf(info->architecture, sizeof info->architecture, "%s", arch.c_str());
    info->n_tensors = (int)g.tensors.size(); info->version = (int)g.version;
    std::map<int, int> counts;
    for (auto& t : g.tensors) counts[t.type]++;
    int best = -1, bc = 0;
    for (auto& kv : counts) if (kv.second > bc) { bc = kv.second; best = kv.first; }
    info->dominant_ggml_type = best;                            // detect_quantization_type: the most frequent tensor type
    info->is_mla = cfg->arch == BZ_ARCH_DEEPSEEK2; info->is_moe = cfg->moe_n_experts > 0; info->is_ssm = cfg->arch == BZ_ARCH_MAMBA2;
    info->file_size_bytes = (uint64_t)g.file.n;
  }
  return BZ_OK;
  BZ_API_END
}

// JSON listing of a SafeTensors checkpoint (single file, first shard or directory): names, dtypes, shapes -- SafeTensorsLoader::{tensor_names, tensor_info}
extern "C" int bz_safetensors_describe(const char* path, char* json_out, size_t cap, size_t* needed) {
  BZ_API_BEGIN
  if (!path) BZ_FAIL(BZ_E_INVALID, "safetensors_describe: null argument");
  StLoader st;
  BZ_TRY(st.open(path));
  std::string s = "{\"num_shards\": " + std::to_string(st.files.size()) + ", \"total_size\": " + std::to_string(st.total) + ", \"tensors\": {";
  bool first = true;
  for (auto& kv : st.tensors) {
    if (!first) s += ", ";
    first = false;
    s += "\"" + json_escape(kv.first) + "\": {\"dtype\": \"" + json_escape(kv.second.dtype) + "\", \"shape\": [";
    for (size_t i = 0; i < kv.second.shape.size(); i++) s += (i ? ", " : "") + std::to_string(kv.second.shape[i]);
    s += "], \"bytes\": " + std::to_string(kv.second.bytes) + "}";
  }
  s += "}}";
  if (needed) *needed = s.size() + 1;
  if (json_out && cap) { const size_t n = std::min(cap - 1, s.size()); memcpy(json_out, s.data(), n); json_out[n] = 0; }
  return BZ_OK;
  BZ_API_END
}

// loaders.rs load_model: detect the source, read the config, move every tensor through bz_model_add_*, finalize
extern "C" int bz_load_model(bz_device* dev, const char* path, bz_model** out, bz_model_config* cfg_out) {
  BZ_API_BEGIN
  if (!dev || !path || !out) BZ_FAIL(BZ_E_INVALID, "load_model: null argument");
  bz_model_source src;
  BZ_TRY(bz_detect_model_source(path, &src));
  if (src.format == BZ_FORMAT_GGUF) return load_gguf(dev, src.weights_path, out, cfg_out);
  return load_safetensors(dev, is_dir(path) ? std::string(path) : std::string(src.weights_path), out, cfg_out);
  BZ_API_END
}
