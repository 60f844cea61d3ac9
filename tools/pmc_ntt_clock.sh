#!/bin/bash
# GPU box: which clock do the 2^20 NTT passes run at, and where do a wave's cycles go?
# GRBM_GUI_ACTIVE / kernel duration = the engine clock during the kernel.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02n}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-msm --no-poly --no-prover --no-ntt-extra --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_clk -- python3 $ARGS > $OUT/pmc_clk.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_mix -- python3 $ARGS > $OUT/pmc_mix.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for sub in ("pmc_clk", "pmc_mix"):
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*kernel_trace.csv" % sub):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in cnt:
        if "ntt_pass" not in k:
            continue
        d = sum(dur[k]) / len(dur[k])
        print(sub, k, "avg_us=%.1f" % (d / 1e3), {c: round(sum(v) / len(v)) for c, v in cnt[k].items()})
PY
