#!/bin/bash
# r05 A/B of library builds on the headline shapes: tools/ab_flags.py (NTT 2^20 / 2^24 fwd+inv, MSM 2^20 with per-kernel times)
# and the 2^20-gate prover, once per library.  usage (through gpurun): bash tools/ab_r05.sh TAG lib1.so lib2.so ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
for L in "$@"; do
  echo "== $L"
  PM_LIB_PATH=$R/$L timeout -k 10 300 python $R/tools/ab_flags.py 2>&1 | grep "NTT\|MSM"
  PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/prover_bench.py 20 7 2>&1 | grep "prove 2"
done > $R/gpurun_out/$TAG/ab.txt 2>&1
cat $R/gpurun_out/$TAG/ab.txt
