#!/bin/bash
# One consolidated GPU validation (run through gpurun): tests, bench, smoke, rocprof stats, PMC.
# Everything is written under gpurun_out/$1/; copy what should be judged into profiles/.
set -u
TAG=${1:-r05z}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 && \
bash tools/collect_pmc.sh $TAG > $OUT/collect_pmc.log 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_prover_summary.json prv_ > $OUT/pmc_prover_summary.txt 2>&1 && \
python tools/pmc_summary.py gpurun_out/$TAG $OUT/pmc_big_summary.json big_ > $OUT/pmc_big_summary.txt 2>&1 && \
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err && \
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 && \
(timeout -k 10 300 ./tools/sort_bench > $OUT/sort_bench.txt 2>&1; true) && \
(timeout -k 10 300 python tools/poly_rows.py 22 10 > $OUT/poly_rows.txt 2>&1; true) && \
(timeout -k 10 300 python tools/prover_bench.py 20 7 > $OUT/prover20.txt 2>&1; true) && \
(timeout -k 10 200 python tools/small_proof_c_sweep.py 10,11,12,13,14,15,16,17 0 > $OUT/small_proofs.txt 2>&1; true) && \
(timeout -k 10 200 python tools/proof_latency_trace.py > $OUT/proof_latency_trace.txt 2>&1; true) && \
(cd /tmp && export TMPDIR=/tmp && \
 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $R/bench.py --steps 200 --no-cpu-baseline --no-msm --no-poly --no-prover --no-ntt-extra > $OUT/stats_headline.json 2> $OUT/stats_headline.err && \
 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --no-cpu-baseline --msm-large-log-n 0 --no-poly --no-ntt-extra > $OUT/stats_bench.json 2> $OUT/stats.err && \
 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm -- python3 $R/tools/msm_table_sweep.py 20 20 0 > $OUT/stats_msm.txt 2> $OUT/stats_msm.err && \
 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm24 -- python3 $R/tools/msm_table_sweep.py 24 22 0 > $OUT/stats_msm24.txt 2> $OUT/stats_msm24.err)
echo "final rc=$?"
tail -3 $OUT/pytest_gpu.txt
cat $OUT/smoke.txt
find $OUT -name "*kernel_stats.csv" | head -4
