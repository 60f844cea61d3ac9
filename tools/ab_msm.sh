#!/bin/bash
# A/B of library builds on the MSM shapes of a proof: for every library given (paths relative to the repo root), one 2^20-point MSM,
# a 2^17-point MSM (shard size) and the 2^20-gate prover.  usage (through gpurun): bash tools/ab_msm.sh TAG lib1.so lib2.so ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
for L in "$@"; do
  echo "== $L"
  PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/msm_table_sweep.py 20 20 8 2>&1 | grep "^c="
  PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/msm_table_sweep.py 17 16 2 2>&1 | grep "^c="
  PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/prover_bench.py 20 7 2>&1 | grep "prove 2\|bucket_chunk\|window_sum\|accumulate_l1 "
done > $R/gpurun_out/$TAG/ab.txt 2>&1
cat $R/gpurun_out/$TAG/ab.txt
