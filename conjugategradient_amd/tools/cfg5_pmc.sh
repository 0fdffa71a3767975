#!/bin/bash
# BASELINE config 5 (random SPD, 10 M rows) evidence: per-kernel stats and PMC passes of tests/perf/cfg5_run.py --no-solve
# (run on the GPU box from the repo root):  bash conjugategradient_amd/tools/cfg5_pmc.sh OUTDIR [rows]
set -u
OUT=$1; ROWS=${2:-10000000}
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
RUN="python3 $GRAFT_REPO_ROOT/tests/perf/cfg5_run.py --no-solve --reps 5 --rows $ROWS"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/stats" -- $RUN) > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
pass() { local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$name" -- $RUN) > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; }
pass rd  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass wr  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_SECTORS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
t = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "spmv" not in k and "tile" not in k: continue
        t[k[-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
stats = {}
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        stats[row["Name"].split("(")[0][-60:]] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
out = {}
for k, cs in sorted(t.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rd = 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0))
    out[k] = {"counters_mean_per_launch": m, "l2_miss_read_bytes": rd, "l2_write_bytes": wr, "stats": stats.get(k)}
print(json.dumps(out, indent=1))
PY
