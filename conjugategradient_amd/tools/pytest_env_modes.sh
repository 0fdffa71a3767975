#!/bin/bash
# The GPU suite under EVERY knob the library keeps (include/MgcgGpu.h: the tuning list): each must leave every result unchanged.
#   bash conjugategradient_amd/tools/pytest_env_modes.sh OUT.log [a|b|c|d|e] ["FILES"]
# FILES: the test files to run instead of the whole suite (e.g. "tests/test_gpu_parallel.py tests/test_gpu_mg.py": all sixteen settings then fit one call).
# (verbose, virtual_devices, force_multirank and fail_comm_init are set by the tests themselves.)
OUT=$1
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
SELECT=${3:-tests}
run() { echo "== $* ($SELECT)" >> "$OUT"; env "$@" timeout -k 10 1100 python -m pytest $SELECT -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3 >> "$OUT"; }
# (parts of at most four runs, so that each fits one gpurun call of 20 minutes -- the suite with every sum in the reference's order, part e,
#  takes a call of its own: its full-size solves run serial sums; no second argument = everything)
PART=${2:-abcde}
if [[ $PART == *a* ]]; then
run MGCG_DEFAULT=1
run MGCG_COMPRESSION=1
run MGCG_COMPRESSION=2
run MGCG_OVERLAP=2
fi
if [[ $PART == *b* ]]; then
run MGCG_OVERLAP=2 MGCG_HALO_STREAM=1
run MGCG_NO_FOLD=1
run MGCG_FOLD_UP=0
run MGCG_FOLD_UP=1
fi
if [[ $PART == *c* ]]; then
run MGCG_TILE_PACK=0
run MGCG_CHECK_EVERY=1
run MGCG_DEEP_HALO=0
run MGCG_OVERLAP=0
fi
if [[ $PART == *d* ]]; then
run MGCG_PLACEMENT=0
run MGCG_AUTO_TILES=0
run MGCG_TILE_SHIFT=19
fi
if [[ $PART == *e* ]]; then
# every sum in the reference's order: one serial sum of 1.3e8 terms takes half a second, so the full-size SOLVES (hundreds of iterations at
# 512^3: tests/test_gpu_fullsize.py, which has its own legs in this mode) are left out of this pass
if [[ -z "$3" ]]; then SELECT="tests --ignore=tests/test_gpu_fullsize.py"; fi
run MGCG_DOT_ORDER=1
fi
cat "$OUT"
