// bz-run -- the smallest C++ caller of libblazr_hip.so: what blazr's `run` command does around the hot path
// (/root/reference/src/cli/run.rs:60-160 GPU branch: open device -> detect + load the model -> warm up -> Executor::generate), with token ids
// in and out instead of a tokenizer (the tokenizers are out of scope).  It exists to show the C ABI driven from compiled code
// with no Python in the process; tests/test_gpu_loader.py runs it against a checkpoint on disk.
//
//   bz-run <model dir | .safetensors | .gguf> --prompt 1,2,3 [--max-tokens N] [--temperature T] [--top-k K] [--top-p P] [--min-p P]
//          [--repeat-penalty R] [--seed S] [--graphs] [--paged-attention] [--device D] [--stats]
// prints the generated ids, comma separated, on stdout.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/blazr_hip.h"

static int fail(const char* what) { fprintf(stderr, "bz-run: %s: %s\n", what, bz_last_error()); return 1; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: bz-run <model path> --prompt id,id,... [--max-tokens N] [--graphs] [--paged-attention] ...\n"); return 2; }
  std::string model = argv[1];
  std::vector<int64_t> prompt;
  bz_gen_config gc;
  memset(&gc, 0, sizeof gc);
  gc.max_tokens = 32; gc.temperature = 0.0f; gc.repeat_penalty = 1.0f; gc.repeat_last_n = 64; gc.top_p = 1.0f; gc.eos_id = -1; gc.block_size = 16;
  gc.dry_base = 2; gc.dynatemp_exponent = 1.0f;
  int device_id = 0; bool stats_on = false;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "bz-run: %s needs a value\n", name); exit(2); } return argv[++i]; };
    if (a == "--prompt") { const char* s = next("--prompt"); char* end; while (*s) { prompt.push_back(strtoll(s, &end, 10)); if (end == s) break; s = *end == ',' ? end + 1 : end; } }
    else if (a == "--max-tokens") gc.max_tokens = atoi(next("--max-tokens"));
    else if (a == "--temperature") gc.temperature = (float)atof(next("--temperature"));
    else if (a == "--top-k") gc.top_k = atoi(next("--top-k"));
    else if (a == "--top-p") gc.top_p = (float)atof(next("--top-p"));
    else if (a == "--min-p") gc.min_p = (float)atof(next("--min-p"));
    else if (a == "--repeat-penalty") gc.repeat_penalty = (float)atof(next("--repeat-penalty"));
    else if (a == "--seed") gc.seed = strtoull(next("--seed"), nullptr, 10);
    else if (a == "--eos") gc.eos_id = strtoll(next("--eos"), nullptr, 10);
    else if (a == "--graphs") gc.use_graph = 1;                    // cli/run.rs:144-157
    else if (a == "--paged-attention") gc.paged = 1;
    else if (a == "--device") device_id = atoi(next("--device"));
    else if (a == "--stats") stats_on = true;
    else { fprintf(stderr, "bz-run: unknown option %s\n", a.c_str()); return 2; }
  }
  if (prompt.empty()) { fprintf(stderr, "bz-run: --prompt id,id,... is required\n"); return 2; }
  bz_device* dev = nullptr;
  if (bz_device_open(device_id, &dev) != BZ_OK) return fail("device");                    // CudaDevice::new + CudaClient::new (run.rs:70-81)
  bz_model* m = nullptr; bz_model_config cfg;
  if (bz_load_model(dev, model.c_str(), &m, &cfg) != BZ_OK) return fail("load");          // detect_model_source + load_model (run.rs:88-118)
  std::vector<int64_t> out((size_t)(gc.max_tokens > 0 ? gc.max_tokens : 1));
  bz_gen_stats st;
  if (bz_generate(m, prompt.data(), (int)prompt.size(), &gc, out.data(), &st) != BZ_OK) return fail("generate");
  for (int i = 0; i < st.n_generated; i++) printf(i ? ",%lld" : "%lld", (long long)out[i]);
  printf("\n");
  if (stats_on)   // cli/bench.rs:299-306 decode tok/s = (tokens - 1) / (total - TTFT)
    fprintf(stderr, "prefill %.2f ms, decode %.2f ms, %d tokens, %.1f tok/s decode, finish=%s\n", st.prefill_ms, st.decode_ms, st.n_generated,
            st.n_generated > 1 ? (st.n_generated - 1) / (st.decode_ms / 1e3) : 0.0, st.finish_reason ? "eos" : "length");
  bz_model_free(m);
  bz_device_close(dev);
  return 0;
}
