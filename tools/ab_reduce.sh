#!/bin/bash
# A/B of library builds on the bucket reduction (one 2^20-point MSM, c = 20): usage (through gpurun):
#   bash tools/ab_reduce.sh TAG "opts" "lbs" lib1.so lib2.so ...     (PM_MSM_STAMPS prints the waves' timeline in diagnostic builds)
TAG=$1; OPTS=$2; LBS=$3; shift 3
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
for L in "$@"; do
  echo "== $L  ($OPTS)"
  PM_OPTS=$OPTS PM_MSM_STAMPS=1 PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/msm_table_sweep.py 20 20 $LBS 2>&1 | grep "^c=\|stamps\]" | awk '/^\[stamps\] waves/{n++} n%4==1 || /^c=/'
done > $R/gpurun_out/$TAG/ab_reduce.txt 2>&1
cat $R/gpurun_out/$TAG/ab_reduce.txt
