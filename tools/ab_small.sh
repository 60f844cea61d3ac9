#!/bin/bash
# A/B of library builds on small and witness-like MSM shapes and small proofs.  usage (through gpurun): bash tools/ab_small.sh TAG lib1.so lib2.so ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
for L in "$@"; do
  echo "== $L"
  PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/small_msm_chunk.py 2>&1 | grep "chunk= 0"
  for k in 10 12 14 16 20; do PM_LIB_PATH=$R/$L timeout -k 10 200 python $R/tools/prover_bench.py $k 9 2>&1 | grep "prove 2"; done
  PM_LIB_PATH=$R/$L timeout -k 10 300 python $R/bench.py --steps 5 --no-cpu-baseline --no-poly --no-ntt-extra --no-prover --msm-large-log-n 0 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); m=b['msm']; print({k:m[k] for k in m if 'ms' in k or 'witness' in k or 'batch' in k})"
done > $R/gpurun_out/$TAG/ab.txt 2>&1
cat $R/gpurun_out/$TAG/ab.txt
