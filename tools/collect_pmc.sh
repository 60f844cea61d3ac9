#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 PMC passes for the bench's kernels.
# Counters go in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), and never
# together with sys/hip/hsa tracing.  Output: gpurun_out/$1/pmc_*/ (csv).
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --msm-steps 1 --msm-large-log-n 0 --no-poly --no-prover --no-ntt-extra --no-msm-extra --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
echo "pmc rc=$?"
# the prover rounds (quotient / permutation / lincomb kernels): same three passes, own directories
PARGS="$GRAFT_REPO_ROOT/tools/prover_bench.py 20 1"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prv_fetch -- python3 $PARGS > $OUT/prv_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prv_write -- python3 $PARGS > $OUT/prv_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prv_sq -- python3 $PARGS > $OUT/prv_sq.log 2>&1
echo "prover pmc rc=$?"
# the 2^24 shapes (ntt_extra.fwd_inv_2^24, msm_large): FETCH_SIZE and WRITE_SIZE only
BARGS="$GRAFT_REPO_ROOT/tools/pmc_big.py 24"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/big_fetch -- python3 $BARGS > $OUT/big_fetch.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/big_write -- python3 $BARGS > $OUT/big_write.log 2>&1
echo "big pmc rc=$?"
find $OUT -name "*counter_collection.csv" | head
