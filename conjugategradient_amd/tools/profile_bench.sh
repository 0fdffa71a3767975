#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the bench command  -> per-kernel average durations
#   2. two --pmc passes on the same command          -> HBM-side read / write request counts of the SpMV kernel
# Usage: bash conjugategradient_amd/tools/profile_bench.sh OUTDIR [bench args...]
set -u
OUT=$1; shift
ARGS="$*"
# one rank only: bench.py --gpus N > 1 starts its ranks from a parent that must be GPU-free, and under rocprofv3 the parent is not
# (profile a single rank of the several-ranks path with tools/forced_path_run.py instead)
case " $ARGS " in *" --gpus "[2-9]*|*" --gpus="[2-9]*|*" --gpus "1[0-9]*) echo "profile_bench.sh: --gpus > 1 is refused under the profiler (use tools/forced_path_run.py)" >&2; exit 2;; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras $ARGS"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/stats" -- $BENCH) > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
(cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_rd" -- $BENCH) > "$OUT/pmc_rd.log" 2>&1
echo "pmc_rd rc=$?"
(cd /tmp && rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_wr" -- $BENCH) > "$OUT/pmc_wr.log" 2>&1
echo "pmc_wr rc=$?"
STEPS=$(echo " $ARGS " | sed -n 's/.* --steps \([0-9][0-9]*\) .*/\1/p'); STEPS=${STEPS:-100}
python3 conjugategradient_amd/tools/profile_summarize.py "$OUT" "$STEPS" > "$OUT/summary.json"
cat "$OUT/summary.json"
