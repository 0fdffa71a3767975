set -u
OUT=gpurun_out/pmc_pattern
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() { local name=$1; shift; (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$name" -- python3 -m conjugategradient_amd.tools.spmv_profile_run --grid 512 --variants 0:64:0 --compression 1) > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum && \
run wr   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum && \
run sq   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD && \
run tcp  TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
