#!/bin/bash
# counter passes over placement_pmc.py (run on the GPU box from the repo root): bash conjugategradient_amd/tools/placement_pmc.sh OUTDIR
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
S=$GRAFT_REPO_ROOT/conjugategradient_amd/tools/placement_pmc.py
run() { local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$name" -- python3 $S) > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; grep "x in" "$OUT/$name.log" | tr '\n' ';'; echo; }
run rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run hit  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
run lvl  TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
run tcp  TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
for name in ("rd", "hit", "lvl", "tcp"):
    rows = []
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "spmv_rowtile" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for c in sorted({r["Counter_Name"] for r in rows}):
        vals = {int(r["Dispatch_Id"]): float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c}
        print(f"{c:42s}", " ".join(f"{vals.get(i, 0):.4g}" for i in ids))
PY
