#!/bin/bash
# One session, one box: every number DESIGN.md quotes for round 2 (run on the GPU box from the repo root).
#   bash conjugategradient_amd/tools/evidence_r2.sh gpurun_out/r2/final
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
T="timeout -k 10"
echo "== bench line (plain CSR headline + extras)";            $T 600 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"; echo "rc=$?"
echo "== bench line, MGCG as the timed loop (CSR)";             $T 300 python bench.py --solver mgcg --steps 30 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_line_mgcg_csr.json" 2> "$OUT/bench_line_mgcg.err"; echo "rc=$?"
echo "== rocprof stats + PMC of the CG loop (CSR)";             bash conjugategradient_amd/tools/profile_bench.sh "$OUT/prof_cg_csr" --steps 30 --warmup 3 > "$OUT/prof_cg_csr_summary.json" 2> "$OUT/prof_cg_csr.err"; echo "rc=$?"
echo "== rocprof stats + PMC of the MGCG loop (CSR)";           bash conjugategradient_amd/tools/profile_bench.sh "$OUT/prof_mgcg_csr" --solver mgcg --steps 20 --warmup 2 > "$OUT/prof_mgcg_csr_summary.json" 2> "$OUT/prof_mgcg_csr.err"; echo "rc=$?"
echo "== SpMV kernels side by side (old rows kernel 9 vs row-tile 10)"
$T 300 python conjugategradient_amd/tools/spmv_sweep.py --grid 512 --variants "rows,DOT rows,rowtile,DOT rowtile,rowtile no sweep" --rounds 7 > "$OUT/spmv_sweep_rows_vs_rowtile.log" 2>&1; echo "rc=$?"
echo "== what the box streams";                                 $T 200 conjugategradient_amd/tools/bw_probe 4 > "$OUT/bw_probe.log" 2>&1; echo "rc=$?"
echo "== slab latency";                                         $T 600 python conjugategradient_amd/tools/slab_latency.py > "$OUT/slab_latency.json" 2> "$OUT/slab_latency.err"; echo "rc=$?"
echo "== MGCG hierarchy next to CG (time to solution)";         $T 600 python conjugategradient_amd/tools/bench_mgcg.py > "$OUT/mgcg_vs_cg_512_csr.json" 2> "$OUT/mgcg_vs_cg.err"; echo "rc=$?"
ls -la "$OUT"
