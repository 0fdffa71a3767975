#!/bin/bash
# Column-tile pass (12-byte entries) against the width of the x window: time per product for several tile counts, and the L2 <-> fabric
# traffic (two rocprofv3 --pmc passes) for three of them.  Run on the GPU box from the repo root:
#   bash conjugategradient_amd/tools/tile_width_sweep.sh OUTDIR
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/conjugategradient_amd/tools/tile_lab
ROWS=10000000
for T in ${TIME_TS:-20 24 27 32 40 54}; do
  W=$(( (ROWS + T - 1) / T ))
  MEAN=$(python3 -c "print(31.0/$T)")
  echo "== T=$T width=$W" | tee -a "$OUT/times.log"
  TILE_LAB_QUICK=1 $LAB $ROWS $T 19 $MEAN $W >> "$OUT/times.log" 2>&1
done
for T in ${PMC_TS:-20 32 40}; do
  W=$(( (ROWS + T - 1) / T ))
  MEAN=$(python3 -c "print(31.0/$T)")
  (cd /tmp && TILE_LAB_QUICK=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_rd_T$T" -- $LAB $ROWS $T 19 $MEAN $W) > "$OUT/pmc_rd_T$T.log" 2>&1
  echo "pmc rd T=$T rc=$?"
  (cd /tmp && TILE_LAB_QUICK=1 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_wr_T$T" -- $LAB $ROWS $T 19 $MEAN $W) > "$OUT/pmc_wr_T$T.log" 2>&1
  echo "pmc wr T=$T rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*_T*"))):
    if not os.path.isdir(d): continue
    T = d.rsplit("_T", 1)[1]
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pass_v1b" not in row["Kernel_Name"]: continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    o = out.setdefault(T, {})
    for k, v in acc.items():
        o[k + "_per_pass"] = sum(v) / len(v); o["launches"] = len(v)
for T, o in out.items():
    rd = 128 * o.get("TCC_EA0_RDREQ_128B_sum_per_pass", 0) + 64 * o.get("TCC_EA0_RDREQ_64B_sum_per_pass", 0) + 32 * o.get("TCC_EA0_RDREQ_32B_sum_per_pass", 0)
    wr = 64 * o.get("TCC_EA0_WRREQ_64B_sum_per_pass", 0) + 32 * (o.get("TCC_EA0_WRREQ_sum_per_pass", 0) - o.get("TCC_EA0_WRREQ_64B_sum_per_pass", 0))
    o["read_bytes_per_pass"] = rd; o["write_bytes_per_pass"] = wr
    o["read_bytes_per_product"] = rd * int(T); o["write_bytes_per_product"] = wr * int(T)
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
