#!/bin/bash
# window width of a 2^24-point MSM with the SRS table (VERDICT r02 #4e): c = 20, 22, 24 x buckets per lane pair
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03e}
mkdir -p $OUT
cd $R
timeout -k 10 900 python tools/msm_table_sweep.py 24 20,22,24 8,32,64 2>&1 | grep -v amdgpu.ids > $OUT/msm_sweep24.txt
echo rc=$? >> $OUT/msm_sweep24.txt
cat $OUT/msm_sweep24.txt
