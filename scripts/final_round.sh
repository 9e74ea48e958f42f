#!/bin/bash
# Round-end measurements on the GPU box (run through gpurun from the repo root, after scripts/profile_round.sh):
#   1. the bench line of every BASELINE config (defaults: 1 warm-up + 3 timed repetitions, CPU baseline leg included)
#   2. prompt-phase timings (scripts/bench_prefill*.py), batched decode (scripts/bench_batch.py --graph)
#   3. rocprofv3 per-kernel summary of a 2048-token prompt on 8 layers of the 8B AWQ shape, and one --pmc pass (MFMA busy cycles; kernel trace only)
# Outputs under gpurun_out/final/ ; copy the summaries into profiles/ afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for p in mistral-7b-q4km llama3.2-1b-bf16 mamba2-2.7b deepseek-v2-lite; do
  python3 bench.py --preset $p > $O/bench_$p.json 2> $O/bench_$p.err
  echo "$p done"
done
python3 scripts/bench_prefill.py --preset llama3-8b-awq > $O/prefill_llama3-8b-awq.json 2> $O/pf1.err
python3 scripts/bench_prefill.py --preset llama3.2-1b-bf16 > $O/prefill_llama3.2-1b-bf16.json 2> $O/pf2.err
python3 scripts/bench_prefill_mamba2.py > $O/prefill_mamba2-2.7b.json 2> $O/pf3.err
python3 scripts/bench_batch.py --graph --batches 2,4,8,16,32,64 > $O/batched_decode_graph.json 2> $O/bb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf_trace -- python3 scripts/prof_prefill.py llama3-8b-awq 2048 2 8 > $O/pf_trace.log 2>&1
cp $(ls $O/pf_trace/*/*kernel_stats.csv | head -1) $O/prefill_kernel_stats_llama3-8b-awq_2048.csv
rm -rf $O/pf_trace
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pf_pmc -- python3 scripts/prof_prefill.py llama3-8b-awq 2048 2 8 > $O/pf_pmc.log 2>&1 || echo "pmc pass failed"
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/final/pf_pmc/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    for k, v in agg.items():
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            d = n[k] or 1
            out[k] = {"dispatches": d, "mfma_busy_cycles_per_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"] / d, "busy_cu_cycles_per_launch": v.get("SQ_BUSY_CU_CYCLES", 0) / d,
                      "gui_active_per_launch": v.get("GRBM_GUI_ACTIVE", 0) / d}
json.dump(out, open("gpurun_out/final/prefill_mfma_pmc_llama3-8b-awq_2048.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $O/pf_pmc
for f in $O/bench_*.json; do python3 scripts/bench_kernels.py $f | head -1 | cut -c1-120; done
cat $O/prefill_*.json | cut -c1-600
