#!/bin/bash
# rocprofv3 counter passes over "spmv_lab N 1 pmc" (run on the GPU box from the repo root): bash conjugategradient_amd/tools/lab_pmc.sh OUTDIR [n]
set -u
OUT=$1; N=${2:-512}
mkdir -p "$OUT"
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/conjugategradient_amd/tools/spmv_lab
run() { local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$name" -- $LAB $N 1 pmc) > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"; }
run rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run hit  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
run wr   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum
run lvl  TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
run tcp  TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run sq   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
t = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        if "v2_rows" not in k: continue
        key = (k, int(row["Grid_Size"]), int(row["Workgroup_Size"]))
        t[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in sorted(t):
    print(key)
    for c, v in sorted(t[key].items()):
        print(f"    {c:40s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
