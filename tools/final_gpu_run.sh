#!/bin/bash
# One consolidated GPU validation (run through gpurun): tests, bench, smoke, rocprof stats, PMC.
# Everything is written under gpurun_out/$1/; copy what should be judged into profiles/ (tools/refresh_profiles.sh).
# usage: bash tools/final_gpu_run.sh [TAG] [ROUND] [PART]   PART = 1: tests + PMC passes, 2: bench + tools + rocprof stats, all (default).
# gpurun allows 1200 s per call: run PART 1, copy the PMC summaries into profiles/ (refresh_profiles.sh does), then PART 2.
set -u
TAG=${1:-r06z}
ROUND=${2:-r06}
PART=${3:-all}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# an optional step: its failure does not stop the run, but leaves FILE.failed beside the output so that refresh_profiles.sh
# does not copy a shell error into profiles/ as if it were evidence (profiles/r05_sort_bench.txt was one)
opt() { local f=$1 t=$2; shift 2; rm -f $OUT/$f.failed; timeout -k 10 $t "$@" > $OUT/$f 2>&1 || echo "rc=$? cmd=$*" > $OUT/$f.failed; true; }
if [ $PART != 2 ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 && \
bash tools/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_prover_summary.json prv_ > $OUT/pmc_prover_summary.txt 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_big_summary.json big_ > $OUT/pmc_big_summary.txt 2>&1
echo "part 1 rc=$?"; tail -3 $OUT/pytest_gpu.txt
fi
if [ $PART != 1 ]; then
# bench.py quotes roofline.traffic from THIS round's committed PMC summaries: on the box they are the ones just made
# (refresh_profiles.sh copies the same files into profiles/ afterwards, so the committed line and its sources agree)
for k in pmc_summary pmc_prover_summary pmc_big_summary; do [ -s $OUT/$k.json ] && cp $OUT/$k.json profiles/${ROUND}_$k.json; done
( time timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time && \
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 && \
opt poly_rows.txt 300 python tools/poly_rows.py 22 10 && \
opt poly_rows_2p20.txt 300 python tools/poly_rows.py 20 10 && \
opt binv_quads.txt 300 python tools/binv_bench.py && \
opt ruffini.txt 300 python tools/ruffini_ab.py && \
opt prover20.txt 300 python tools/prover_bench.py 20 7 && \
opt prover24.txt 400 python tools/prover_bench.py 24 3 wide && \
opt small_proofs.txt 200 python tools/small_proof_c_sweep.py 10,11,12,13,14,15,16,17 0 && \
opt proof_latency_trace.txt 200 python tools/proof_latency_trace.py && \
(cd /tmp && export TMPDIR=/tmp && \
 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $R/bench.py --steps 200 --no-cpu-baseline --no-msm --no-poly --no-prover --no-ntt-extra > $OUT/stats_headline.json 2> $OUT/stats_headline.err && \
 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --no-cpu-baseline --msm-large-log-n 0 --prover-large-log-n 0 --no-poly --no-ntt-extra > $OUT/stats_bench.json 2> $OUT/stats.err && \
 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm -- python3 $R/tools/msm_table_sweep.py 20 20 0 > $OUT/stats_msm.txt 2> $OUT/stats_msm.err && \
 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm24 -- python3 $R/tools/msm_table_sweep.py 24 22 0 > $OUT/stats_msm24.txt 2> $OUT/stats_msm24.err)
echo "final rc=$?"
cat $OUT/smoke.txt
cat $OUT/bench.time
fi
find $OUT -name "*kernel_stats.csv" | head -4
