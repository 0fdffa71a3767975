#!/bin/bash
# One session, one box: every number DESIGN.md quotes for round 5 (run on the GPU box from the repo root).
#   bash conjugategradient_amd/tools/evidence_r5.sh gpurun_out/r5/final
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
echo "== what the box streams";                                 $T 200 conjugategradient_amd/tools/bw_probe 4 > "$OUT/bw_probe.log" 2>&1; echo "rc=$?"
echo "== one rank's share of config 4 on the several-ranks path of a one-rank RCCL communicator: plain / per-pass exchanges / deep-halo cycle"
bash conjugategradient_amd/tools/deep_halo_ab.sh "$OUT/deep_halo_ab.log" 3 > "$OUT/deep_halo_ab_summary.json" 2> "$OUT/deep_halo_ab.err"; echo "rc=$?"
echo "== the same for plain CG (several-ranks path against the plain loop)"
for leg in "--force 0" ""; do $T 200 python conjugategradient_amd/tools/forced_path_run.py --solver cg --overlap 1 --steps 200 --repeats 3 $leg 2>/dev/null | tail -n 1 >> "$OUT/forced_path_cg.log"; done; echo "rc=$?"
echo "== the reference's own driver";                           $T 300 python tests/perf/reference_driver_run.py > "$OUT/reference_driver_mgcgmain_207402.json" 2> "$OUT/reference_driver.err"; echo "rc=$?"
echo "== MGCG hierarchy next to CG (time to solution)";         $T 600 python conjugategradient_amd/tools/bench_mgcg.py > "$OUT/mgcg_vs_cg_512_csr.json" 2> "$OUT/mgcg_vs_cg.err"; echo "rc=$?"
echo "== config 5";                                             $T 600 python tests/perf/cfg5_run.py --reps 20 > "$OUT/config5_random_spd_10M_line.json" 2> "$OUT/config5.err"; echo "rc=$?"
echo "== N = 2 rehearsal of the driver's scaling command (two ranks share the one GPU over the host-staged transport)"
$T 400 python bench.py --gpus 2 --grid 256 --steps 20 --warmup 3 --allow-fallback > "$OUT/bench_gpus2_grid256_one_gpu_box.json" 2> "$OUT/bench_gpus2_grid256.err"; echo "rc=$?"
ls -la "$OUT"
