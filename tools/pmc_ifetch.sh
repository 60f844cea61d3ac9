#!/bin/bash
# GPU box (via gpurun): instruction-fetch / issue counters of one 2^20-point MSM (accumulate vs bucket reduction).
set -u
TAG=${1:-ifetch}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|SQC_[A-Z_0-9]*" $OUT/counters.txt | sort -u | tr '\n' ' ' > $OUT/sq_names.txt
ARGS="$GRAFT_REPO_ROOT/tools/msm_table_sweep.py 20 20 0"
for SET in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES"; do
  D=$OUT/$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 $ARGS > $D.log 2>&1
  echo "$SET rc=$?"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for (kn, cn), (n, v) in sorted(acc.items()):
        if "accumulate_l1" in kn or "bucket_wave" in kn or "level_kernel" in kn:
            print(f"{kn:42s} {cn:28s} launches {n:3d} mean {v / n:.4e}")
PY
