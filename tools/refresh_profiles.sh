#!/bin/bash
# After `gpurun -- bash tools/final_gpu_run.sh TAG`: copy what is judged from gpurun_out/TAG into profiles/ (newest file of each kind:
# gpurun merges runs into the same directories).  An artefact whose producing command failed (FILE.failed beside it, or the file
# missing / empty) is SKIPPED and named on stderr.  usage: bash tools/refresh_profiles.sh [TAG] [ROUND]
TAG=${1:-r06z}; R=${2:-r06}
O=gpurun_out/$TAG
take() {   # take SRC DST
  if [ -e "$1.failed" ] || [ ! -s "$1" ]; then echo "SKIPPED $2: $1 is missing, empty or its command failed ($(cat "$1.failed" 2>/dev/null))" >&2; return; fi
  cp "$1" "$2"
}
for p in headline:stats_headline bench:stats msm:stats_msm msm24:stats_msm24; do
  n=${p%%:*}; d=${p##*:}
  f="$(ls -t $O/$d/runc/*_kernel_stats.csv 2>/dev/null | head -1)"
  take "${f:-$O/$d/none}" profiles/${R}_${n}_kernel_stats.csv
done
take $O/bench.json profiles/${R}_bench.json
take $O/pmc_summary.json profiles/${R}_pmc_summary.json
take $O/pmc_prover_summary.json profiles/${R}_pmc_prover_summary.json
take $O/pmc_big_summary.json profiles/${R}_pmc_big_summary.json
take $O/small_proofs.txt profiles/${R}_small_proofs.txt
take $O/proof_latency_trace.txt profiles/${R}_proof_latency_trace.txt
take $O/pytest_gpu.txt profiles/${R}_pytest_gpu.txt
take $O/poly_rows.txt profiles/${R}_poly_rows_2p22.txt
take $O/poly_rows_2p20.txt profiles/${R}_poly_rows_2p20.txt
take $O/binv_quads.txt profiles/${R}_binv_quads_final.txt
take $O/ruffini.txt profiles/${R}_ruffini.txt
take $O/prover20.txt profiles/${R}_prover_2p20.txt
take $O/prover24.txt profiles/${R}_prover_2p24_kernels.txt
python3 - <<PY
import json
b = json.load(open("profiles/${R}_bench.json"))
m, p = b["msm"], b["prover"]
print("ntt", "%.3e" % b["value"], b["roofline"]["all_kernels_us"], "| msm", round(m["ms_per_msm"], 3), "batch4", round(m["batch4"]["ms_per_msm"], 3),
      "witness", round(m["witness_like"]["ms_per_msm"], 3), "shard", round(m["shard_1_of_8"]["ms_per_msm"], 3), m["shard_1_of_8"]["tail_frac"],
      "| msm24", round(b["msm_large"]["ms_per_msm"], 2), "| proof", p["ms_per_proof"], p["two_contexts_ms_per_proof"],
      "| 2^24 proof", p["large"]["replicated"]["ms_per_proof"], p["large"]["distributed"]["ms_per_proof"])
PY
