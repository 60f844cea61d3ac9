#!/bin/bash
# After `gpurun -- bash tools/final_gpu_run.sh TAG`: copy what is judged from gpurun_out/TAG into profiles/ (newest file of each kind:
# gpurun merges runs into the same directories).  usage: bash tools/refresh_profiles.sh [TAG] [ROUND]
TAG=${1:-r05z}; R=${2:-r05}
O=gpurun_out/$TAG
for p in headline:stats_headline bench:stats msm:stats_msm msm24:stats_msm24; do
  n=${p%%:*}; d=${p##*:}
  cp "$(ls -t $O/$d/runc/*_kernel_stats.csv | head -1)" profiles/${R}_${n}_kernel_stats.csv
done
cp $O/bench.json profiles/${R}_bench.json
cp $O/pmc_summary.json profiles/${R}_pmc_summary.json
cp $O/pmc_prover_summary.json profiles/${R}_pmc_prover_summary.json
cp $O/pmc_big_summary.json profiles/${R}_pmc_big_summary.json
cp $O/small_proofs.txt profiles/${R}_small_proofs.txt
cp $O/proof_latency_trace.txt profiles/${R}_proof_latency_trace.txt
cp $O/pytest_gpu.txt profiles/${R}_pytest_gpu.txt
cp $O/sort_bench.txt profiles/${R}_sort_bench.txt
cp $O/poly_rows.txt profiles/${R}_poly_rows_2p22.txt
cp $O/prover20.txt profiles/${R}_prover_2p20.txt
python3 - <<PY
import json
b = json.load(open("profiles/${R}_bench.json"))
m, p = b["msm"], b["prover"]
print("ntt", "%.3e" % b["value"], b["roofline"]["all_kernels_us"], "| msm", round(m["ms_per_msm"], 3), "batch4", round(m["batch4"]["ms_per_msm"], 3),
      "witness", round(m["witness_like"]["ms_per_msm"], 3), "shard", round(m["shard_1_of_8"]["ms_per_msm"], 3), m["shard_1_of_8"]["tail_frac"],
      "| msm24", round(b["msm_large"]["ms_per_msm"], 2), "| proof", p["ms_per_proof"], p["two_contexts_ms_per_proof"])
PY
