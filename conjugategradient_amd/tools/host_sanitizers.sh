#!/bin/bash
# The library's HOST code under AddressSanitizer and ThreadSanitizer on the GPU box (csrc/Makefile: targets asan, tsan; GPU ASan is not available on the pool):
#   bash conjugategradient_amd/tools/host_sanitizers.sh OUTDIR ["FILES"]        (FILES default: the multi-rank and multigrid test files)
OUT=${1:-gpurun_out/sanitizers}
FILES=${2:-tests/test_gpu_parallel.py tests/test_gpu_mg.py}
mkdir -p "$OUT"
CLANG=/opt/rocm/lib/llvm/bin/clang
LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:quarantine_size_mb=4096 PYTHONMALLOC=malloc \
  MGCG_LIB_PATH=$PWD/conjugategradient_amd/tools/libMgcgGpu_asan.so timeout -k 10 500 python -m pytest $FILES -m gpu -q -p no:cacheprovider > "$OUT/asan.log" 2>&1
echo "asan rc=$?"
LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.tsan-x86_64.so) TSAN_OPTIONS=ignore_noninstrumented_modules=1:report_signal_unsafe=0 \
  MGCG_LIB_PATH=$PWD/conjugategradient_amd/tools/libMgcgGpu_tsan.so timeout -k 10 500 python -m pytest $FILES -m gpu -q -p no:cacheprovider > "$OUT/tsan.log" 2>&1
echo "tsan rc=$?"
tail -n 2 "$OUT/asan.log" "$OUT/tsan.log"
grep -c "ERROR: AddressSanitizer" "$OUT/asan.log"; grep -c "WARNING: ThreadSanitizer" "$OUT/tsan.log"
exit 0
