// bz_sampler_host.hip -- the host-side half of blazr's sampler, restated line for line from the reference's Rust:
//   compute_dynamic_temperature   /root/reference/src/engine/sampling.rs:41-86
//   compute_logprobs              /root/reference/src/engine/sampling.rs:197-256
//   apply_dry_penalty             /root/reference/src/engine/sampling.rs:270-320
//   apply_typical_filter          /root/reference/src/engine/sampling.rs:322-369
//   apply_logit_bias              /root/reference/src/engine/sampling.rs:464-480
//   MirostatState::{new, sample}  /root/reference/src/engine/mirostat.rs:19-110
// All of it is plain f32 host arithmetic over one row of logits, exactly as in the reference (which pulls the logits to the CPU for these
// options: sampling.rs:393-411, 118-131).  Mirostat's uniform draw comes from splitmix64 instead of rand::StdRng (not reproducible here).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "bz_internal.h"

extern "C" float bz_compute_dynamic_temperature(const float* logits, int64_t V, float base, float range, float exponent) {
  float max_logit = -INFINITY;
  for (int64_t i = 0; i < V; i++) max_logit = fmaxf(max_logit, logits[i]);
  std::vector<float> probs((size_t)V);
  float sum = 0.f;
  for (int64_t i = 0; i < V; i++) { probs[i] = expf(logits[i] - max_logit); sum += probs[i]; }
  float entropy = 0.f;
  for (int64_t i = 0; i < V; i++) { const float n = probs[i] / sum; if (n > 0.f) entropy -= n * logf(n); }
  const float max_entropy = logf((float)V);
  const float ne = max_entropy > 0.f ? fminf(fmaxf(entropy / max_entropy, 0.f), 1.f) : 0.5f;
  const float mapped = powf(ne, exponent);
  const float t = base - range + 2.0f * range * mapped;
  return fmaxf(t, 0.01f);
}

extern "C" int bz_apply_dry_penalty(float* logits, int64_t V, const uint32_t* recent, int64_t n_recent, float multiplier, int base, int allowed_length) {
  BZ_API_BEGIN
  if (!logits || (!recent && n_recent) || base < 1) BZ_FAIL(BZ_E_INVALID, "apply_dry_penalty: bad argument");
  const uint32_t* h = recent; int64_t hn = n_recent;
  if (allowed_length > 0 && allowed_length < n_recent) { h = recent + (n_recent - allowed_length); hn = allowed_length; }
  if (hn < base) return BZ_OK;
  const int64_t suffix_start = hn - (base - 1) > 0 ? hn - (base - 1) : 0;       // saturating_sub
  const int64_t slen = hn - suffix_start;
  for (int64_t start = 0; start < suffix_start; start++) {
    const int64_t end = start + slen;
    if (end >= hn) break;
    if (memcmp(h + start, h + suffix_start, (size_t)slen * 4) == 0) {
      int64_t match_len = slen, s = start, e = suffix_start;
      while (s > 0 && e > 0 && h[s - 1] == h[e - 1]) { match_len++; s--; e--; }
      const uint32_t next = h[end];
      if ((int64_t)next < V) logits[next] -= multiplier * (float)match_len;
    }
  }
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_apply_typical_filter(float* logits, int64_t V, float typical_p) {
  BZ_API_BEGIN
  if (!logits || V <= 0) BZ_FAIL(BZ_E_INVALID, "apply_typical_filter: bad argument");
  float max_logit = -INFINITY;
  for (int64_t i = 0; i < V; i++) max_logit = fmaxf(max_logit, logits[i]);
  std::vector<float> probs((size_t)V);
  float sum = 0.f;
  for (int64_t i = 0; i < V; i++) { probs[i] = expf(logits[i] - max_logit); sum += probs[i]; }
  float entropy = 0.f;
  for (int64_t i = 0; i < V; i++) { const float n = probs[i] / sum; if (n > 0.f) entropy -= n * logf(n); }
  struct Dev { int64_t idx; float dev, p; };
  std::vector<Dev> d((size_t)V);
  for (int64_t i = 0; i < V; i++) {
    const float n = probs[i] / sum;
    const float info = n > 0.f ? -logf(n) : INFINITY;
    d[i] = Dev{i, fabsf(info - entropy), n};
  }
  std::stable_sort(d.begin(), d.end(), [](const Dev& a, const Dev& b) { return a.dev < b.dev; });   // Rust's sort_by is stable
  std::vector<char> keep((size_t)V, 0);
  float cumsum = 0.f;
  for (auto& x : d) {
    if (cumsum >= typical_p && cumsum > 0.f) break;
    keep[x.idx] = 1;
    cumsum += x.p;
  }
  for (int64_t i = 0; i < V; i++) if (!keep[i]) logits[i] = -INFINITY;
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_apply_logit_bias(float* logits, int64_t V, const uint32_t* ids, const float* bias, int n) {
  BZ_API_BEGIN
  if (!logits || (n && (!ids || !bias))) BZ_FAIL(BZ_E_INVALID, "apply_logit_bias: bad argument");
  std::vector<float> b((size_t)V, 0.f);           // sampling.rs:470-476: a dense bias vector, later entries of the map overwrite earlier ones
  for (int i = 0; i < n; i++) if ((int64_t)ids[i] < V) b[ids[i]] = bias[i];
  for (int64_t i = 0; i < V; i++) logits[i] += b[i];
  return BZ_OK;
  BZ_API_END
}

extern "C" int bz_compute_logprobs(const float* logits, int64_t V, uint32_t chosen, int top_n, float* chosen_logprob, uint32_t* top_ids, float* top_lps, int* n_top) {
  BZ_API_BEGIN
  if (!logits || V <= 0 || !chosen_logprob || !n_top) BZ_FAIL(BZ_E_INVALID, "compute_logprobs: bad argument");
  float max_logit = -INFINITY;
  for (int64_t i = 0; i < V; i++) max_logit = fmaxf(max_logit, logits[i]);
  float s = 0.f;
  for (int64_t i = 0; i < V; i++) s += expf(logits[i] - max_logit);
  const float lse = logf(s) + max_logit;
  const int n = (int)std::min<int64_t>(std::min<int64_t>(top_n, V), 20);
  struct E { int64_t id; float lp; };
  std::vector<E> top;
  auto sort_desc = [&]() { std::stable_sort(top.begin(), top.end(), [](const E& a, const E& b) { return a.lp > b.lp; }); };
  for (int64_t i = 0; i < V; i++) {
    const float lp = logits[i] - lse;
    if ((int)top.size() < n) { top.push_back(E{i, lp}); if ((int)top.size() == n) sort_desc(); }
    else if (n > 0 && lp > top[n - 1].lp) { top[n - 1] = E{i, lp}; sort_desc(); }
  }
  sort_desc();
  *chosen_logprob = (int64_t)chosen < V ? logits[chosen] - lse : -INFINITY;
  *n_top = (int)top.size();
  for (size_t i = 0; i < top.size(); i++) { if (top_ids) top_ids[i] = (uint32_t)top[i].id; if (top_lps) top_lps[i] = top[i].lp; }
  return BZ_OK;
  BZ_API_END
}

struct bz_mirostat { float mu, tau, eta; uint64_t rng; };
extern "C" int bz_mirostat_create(float tau, float eta, uint64_t seed, bz_mirostat** out) {
  BZ_API_BEGIN
  if (!out) BZ_FAIL(BZ_E_INVALID, "mirostat_create: null argument");
  *out = new bz_mirostat{2.0f * tau, tau, eta, seed};     // mirostat.rs:27-33: mu = 2 tau
  return BZ_OK;
  BZ_API_END
}
extern "C" int bz_mirostat_free(bz_mirostat* s) { delete s; return BZ_OK; }
extern "C" float bz_mirostat_mu(const bz_mirostat* s) { return s ? s->mu : 0.f; }
extern "C" int bz_mirostat_sample(bz_mirostat* st, const float* logits, int64_t V, float temperature, uint32_t* token, float* logprob) {
  BZ_API_BEGIN
  if (!st || !logits || V <= 0 || !token) BZ_FAIL(BZ_E_INVALID, "mirostat_sample: bad argument");
  std::vector<float> scaled(logits, logits + V);
  if (temperature != 1.0f && temperature > 0.0f) { const float inv_t = 1.0f / temperature; for (auto& l : scaled) l *= inv_t; }
  float max_logit = -INFINITY;
  for (float l : scaled) max_logit = fmaxf(max_logit, l);
  struct P { int64_t id; float p; };
  std::vector<P> probs((size_t)V);
  float sum = 0.f;
  for (int64_t i = 0; i < V; i++) { probs[i] = P{i, expf(scaled[i] - max_logit)}; sum += probs[i].p; }
  for (auto& x : probs) x.p /= sum;
  std::stable_sort(probs.begin(), probs.end(), [](const P& a, const P& b) { return a.p > b.p; });
  std::vector<P> cand;
  for (auto& x : probs) if (x.p > 0.f && -log2f(x.p) <= st->mu) cand.push_back(x);
  if (cand.empty()) cand.push_back(probs[0]);
  float total = 0.f;
  for (auto& x : cand) total += x.p;
  uint64_t z = (st->rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  const float r = (float)(z >> 40) * (1.0f / 16777216.0f);     // 24 random bits -> [0, 1), as rand's Standard f32
  float cumsum = 0.f;
  int64_t chosen = cand[0].id; float chosen_prob = cand[0].p;
  for (auto& x : cand) { cumsum += x.p / total; if (cumsum > r) { chosen = x.id; chosen_prob = x.p; break; } }
  const float surprise = chosen_prob > 0.f ? -log2f(chosen_prob) : st->tau;
  st->mu -= st->eta * (surprise - st->tau);
  *token = (uint32_t)chosen;
  if (logprob) *logprob = logf(chosen_prob);
  return BZ_OK;
  BZ_API_END
}
