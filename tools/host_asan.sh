#!/bin/bash
# Host-side AddressSanitizer build of the library (device code untouched: -Xarch_host) and the CPU tests that run its pure-host
# entry points under it (bucket-fill geometry, MSM sizing pass, transform plan, exchange fold, Keccak / transcript, ABI surface).
# CPU only -- GPU sanitizer runs are not available on this pool.   usage: bash tools/host_asan.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/plonk_asan_build
mkdir -p $B
cd $R/plonk-prototype_amd/csrc
for f in api ntt ntt4 msm poly plonk_rounds transcript prover comm; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function -Wno-pass-failed \
    -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -o $B/libplonk_asan.so $B/{api,ntt,ntt4,msm,poly,plonk_rounds,transcript,prover,comm}.o -ldl
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT PM_LIB_PATH=$B/libplonk_asan.so python -m pytest tests/test_msm_geometry.py tests/test_ntt_plan.py tests/test_abi.py tests/test_transcript.py -x -q
