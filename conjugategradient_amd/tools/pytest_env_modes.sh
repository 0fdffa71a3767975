#!/bin/bash
# The GPU suite under the library's schedule / form switches: every one of them must leave every result unchanged.
#   bash conjugategradient_amd/tools/pytest_env_modes.sh OUT.log
OUT=$1
: > "$OUT"
run() { echo "== $*" >> "$OUT"; env "$@" timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3 >> "$OUT"; }
# (two halves, so that each fits one gpurun call: pytest_env_modes.sh OUT.log a|b; no second argument = everything)
HALF=${2:-abc}
if [[ $HALF == *a* ]]; then
run MGCG_DEFAULT=1
run MGCG_COMPRESSION=1
run MGCG_COMPRESSION=2
run MGCG_OVERLAP=2
run MGCG_OVERLAP=2 MGCG_HALO_STREAM=1
run MGCG_NO_FOLD=1
fi
if [[ $HALF == *b* ]]; then
run MGCG_FOLD_UP=0
run MGCG_FOLD_UP=1
run MGCG_NO_ZSWEEP=1
run MGCG_TILE_PACK=0
run MGCG_NO_FOLDED_FINALIZE=1
run MGCG_LAZY_CODE_OBJECTS=1
fi
if [[ $HALF == *c* ]]; then            # round 4's knobs
run MGCG_OVERLAP=0
run MGCG_PLACEMENT=0
run MGCG_AUTO_TILES=0
run MGCG_TILE_SHIFT=19
fi
cat "$OUT"
