#!/bin/bash
# The GPU suite under EVERY knob the library keeps (include/MgcgGpu.h: the tuning list): each must leave every result unchanged.
#   bash conjugategradient_amd/tools/pytest_env_modes.sh OUT.log [a|b|c]
# (verbose, virtual_devices, force_multirank and fail_comm_init are set by the tests themselves; check_every, dot_order and the rest below.)
OUT=$1
: > "$OUT"
run() { echo "== $*" >> "$OUT"; env "$@" timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3 >> "$OUT"; }
# (three parts, so that each fits one gpurun call; no second argument = everything)
HALF=${2:-abc}
if [[ $HALF == *a* ]]; then
run MGCG_DEFAULT=1
run MGCG_COMPRESSION=1
run MGCG_COMPRESSION=2
run MGCG_OVERLAP=2
run MGCG_OVERLAP=2 MGCG_HALO_STREAM=1
fi
if [[ $HALF == *b* ]]; then
run MGCG_NO_FOLD=1
run MGCG_FOLD_UP=0
run MGCG_FOLD_UP=1
run MGCG_TILE_PACK=0
run MGCG_CHECK_EVERY=1
fi
if [[ $HALF == *c* ]]; then
run MGCG_OVERLAP=0
run MGCG_PLACEMENT=0
run MGCG_AUTO_TILES=0
run MGCG_TILE_SHIFT=19
run MGCG_DOT_ORDER=1
fi
cat "$OUT"
