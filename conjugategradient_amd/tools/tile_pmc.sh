#!/bin/bash
# Counter passes over the column-tile product of BASELINE config 5's shape (tools/tile_lab): the production pass (tile-major, y read and
# rewritten per tile) and the row-block-persistent variant V3 (y written once).  What bounds the pass -- fabric traffic, L2 -> L1 line fills of
# the gathers, or latency?  Run on the GPU box from the repo root:   bash conjugategradient_amd/tools/tile_pmc.sh OUTDIR [T] 
# Each counter group is its own rocprofv3 run (--pmc with --kernel-trace only).
set -u
OUT=$1; T=${2:-27}
mkdir -p "$OUT"
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/conjugategradient_amd/tools/tile_lab
ROWS=10000000
W=$(( (ROWS + T - 1) / T ))
MEAN=$(python3 -c "print(31.0/$T)")
run() { local variant=$1 name=$2; shift 2
  local envv="TILE_LAB_QUICK=1"; [ "$variant" = v3 ] && envv="TILE_LAB_V3_ONE=1"
  (cd /tmp && env $envv rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/${variant}_$name" -- $LAB $ROWS $T 19 $MEAN $W) > "$OUT/${variant}_$name.log" 2>&1
  echo "pass $variant $name rc=$?"; }
for v in prod v3; do
  run $v rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  run $v wr   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
  run $v l2   TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum
  run $v tcp  TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run $v tcp2 TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum
  run $v sq   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
out = {}
for variant in ("prod", "v3"):
    agg = defaultdict(float)
    launches = defaultdict(int)
    dur = defaultdict(float)
    for d in sorted(glob.glob(os.path.join(root, variant + "_*"))):
        if not os.path.isdir(d):
            continue
        # counters: sum over the dispatches of the pass kernels, divided by the number of products (6: one warm-up + 5 timed)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "pass_v" not in k:
                    continue
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
                launches[row["Counter_Name"]] += 1
    out[variant] = {c: {"sum_over_launches": v, "launches": launches[c]} for c, v in sorted(agg.items())}
json.dump(out, open(os.path.join(root, "counters_raw.json"), "w"), indent=1)
print(json.dumps({v: {c: d["sum_over_launches"] for c, d in cs.items()} for v, cs in out.items()}, indent=1))
PY
