#!/bin/bash
# GPU box (via gpurun): per-launch durations of the MSM kernels inside 2^20-gate proofs (rocprofv3 --kernel-trace of tools/prover_bench.py).
set -u
TAG=${1:-ptrace}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prover_bench.py ${2:-20} 2 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last proof: from the last msm_digits_hist launch group backwards -- print the last 40 MSM-stage launches
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("msm_digits_hist", "msm_digits_scatter", "msm_sort_local", "msm_accumulate_l1", "msm_bucket_wave"))]
for r in sel[-20:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{r["Kernel_Name"].split("(")[0][-34:]:36s} grid {r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"):>9s} {d:9.1f} us')
PY
