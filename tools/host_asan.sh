#!/bin/bash
# Host-side AddressSanitizer and UndefinedBehaviorSanitizer builds of the library (device code untouched: -Xarch_host) and the CPU tests that run its pure-host
# entry points under it (bucket-fill geometry, MSM sizing pass, transform plan, exchange fold, Keccak / transcript, ABI surface).
# CPU only -- GPU sanitizer runs are not available on this pool.   usage: bash tools/host_asan.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TESTS="tests/test_msm_geometry.py tests/test_ntt_plan.py tests/test_abi.py tests/test_transcript.py tests/test_domain_helpers.py tests/test_comm_deadline.py"
for SAN in address undefined; do
  B=$R/build_ab/san_${SAN}
  mkdir -p $B
  cd $R/plonk-prototype_amd/csrc
  EXTRA="-Xarch_host -fno-omit-frame-pointer"
  [ $SAN = undefined ] && EXTRA="-Xarch_host -fno-sanitize-recover=undefined"
  for f in api ntt ntt4 msm poly plonk_rounds transcript prover comm; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function -Wno-pass-failed \
      -Wno-option-ignored -Xarch_host -fsanitize=$SAN $EXTRA -c $f.hip -o $B/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wno-option-ignored -fsanitize=$SAN -o $B/libplonk_$SAN.so $B/{api,ntt,ntt4,msm,poly,plonk_rounds,transcript,prover,comm}.o -ldl
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  [ $SAN = undefined ] && RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
  cd $R
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$RT PM_LIB_PATH=$B/libplonk_$SAN.so python -m pytest $TESTS -x -q -s -m "not gpu"
done
