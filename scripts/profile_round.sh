#!/bin/bash
# Round-end profiling of the headline bench on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-latency --steps 32 --warmup 4 --reps 1`
#   3. rocprofv3 --kernel-trace --stats of the four secondary presets
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing domains besides the kernel trace) reduced by scripts/pmc_traffic.py
# Outputs under gpurun_out/prof_round/ ; copy the summaries into profiles/ afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_round
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-latency --steps 32 --warmup 4 --reps 1 > $O/bench_trace.json 2> $O/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 2 --reps 1 > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 2 --reps 1 > $O/bench_write.json 2> $O/write.err
python3 scripts/pmc_traffic.py $O/fetch $O/write > $O/pmc_traffic.json
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/trace/*/*domain_stats.csv | head -1) $O/domain_stats.csv
rm -rf $O/fetch $O/write $O/trace
# secondary presets: rocprofv3 per-kernel summary of a short bench run each (no CPU leg)
for p in mistral-7b-q4km llama3.2-1b-bf16 mamba2-2.7b deepseek-v2-lite; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$p -- python3 bench.py --preset $p --no-cpu-baseline --no-latency --steps 32 --warmup 4 --reps 1 > $O/bench_trace_$p.json 2> $O/trace_$p.err
  cp $(ls $O/trace_$p/*/*kernel_stats.csv | head -1) $O/kernel_stats_$p.csv
  rm -rf $O/trace_$p
done
head -12 $O/kernel_stats.csv | cut -c1-170
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/prof_round/pmc_traffic.json"))
for k, v in d["kernels"].items():
    if v["hbm_bytes_per_launch"] > 5e6:
        print(k[:60], v)
PY
