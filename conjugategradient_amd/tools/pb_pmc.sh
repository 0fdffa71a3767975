#!/bin/bash
# HBM-side traffic of the two kernels of the propagation-blocking product (tools/pb_lab, PB_LAB_ONE=1): two rocprofv3 --pmc passes
# (read requests by size; write requests), run on the GPU box from the repo root:   bash conjugategradient_amd/tools/pb_pmc.sh OUTDIR
# (PB_LAB_ROUNDS_ONLY=1, the default: pass 2 in rounds over the tiles; PB_LAB_ROUNDS_ONLY= : the one-round gather form of 256-row blocks)
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp PB_LAB_ONE=1 PB_LAB_SKIP_RUNS=1 PB_LAB_GATHER_ONLY=1 PB_LAB_ROUNDS_ONLY=${PB_LAB_ROUNDS_ONLY:-1}
LAB=$GRAFT_REPO_ROOT/conjugategradient_amd/tools/pb_lab
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_rd" -- $LAB 10000000 2) > "$OUT/pmc_rd.log" 2>&1
echo "pmc_rd rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_wr" -- $LAB 10000000 2) > "$OUT/pmc_wr.log" 2>&1
echo "pmc_wr rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if "pb_pass" not in k: continue
    m = lambda n: sum(c.get(n, [0])) / max(1, len(c.get(n, [0])))
    rd = 128 * m("TCC_EA0_RDREQ_128B_sum") + 64 * m("TCC_EA0_RDREQ_64B_sum") + 32 * m("TCC_EA0_RDREQ_32B_sum")
    wr = 64 * m("TCC_EA0_WRREQ_64B_sum") + 32 * (m("TCC_EA0_WRREQ_sum") - m("TCC_EA0_WRREQ_64B_sum"))
    res[k] = {"launches": len(c.get("TCC_EA0_RDREQ_sum", [])), "read_GB": rd / 1e9, "written_GB": wr / 1e9, "tcc_hit": m("TCC_HIT_sum"), "tcc_miss": m("TCC_MISS_sum")}
print(json.dumps(res, indent=1))
PY
