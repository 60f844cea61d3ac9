#!/bin/bash
# quick look (through gpurun): MSM at 2^17 / 2^20 / 2^24 with kernel times, a 2^20-gate proof, rocprofv3 kernel stats of the proof
set -u
TAG=${1:-r03q}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 python tools/msm_table_sweep.py 20 20 8 skew > $OUT/msm20.txt 2>&1 && \
timeout -k 10 300 python tools/msm_table_sweep.py 17 16 2 > $OUT/msm17.txt 2>&1 && \
timeout -k 10 600 python tools/msm_table_sweep.py 24 20 8 > $OUT/msm24.txt 2>&1 && \
timeout -k 10 300 python tools/prover_bench.py 20 5 > $OUT/prover20.txt 2>&1 && \
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_prover -- python3 $R/tools/prover_bench.py 20 3 > $OUT/stats_prover.txt 2> $OUT/stats_prover.err)
echo "rc=$?"
cat $OUT/msm20.txt $OUT/msm17.txt $OUT/msm24.txt
tail -25 $OUT/prover20.txt
