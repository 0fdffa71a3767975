#!/bin/bash
# rocprofv3 counter passes for the SpMV kernel (run on the GPU box from the repo root).
# Usage: bash conjugategradient_amd/tools/pmc_passes.sh OUTDIR "--grid 512 --variants 1:128:0,1:128:4"
set -u
OUT=$1; shift
ARGS="$*"
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$name" -- python3 -m conjugategradient_amd.tools.spmv_profile_run $ARGS) > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
export PYTHONPATH=$GRAFT_REPO_ROOT
run rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run hit  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
run wr   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum
run sq   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
run tcp  TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
