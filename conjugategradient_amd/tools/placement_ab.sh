#!/bin/bash
# Best-of-k placement of p against the plain draw: the default bench command in fresh processes, alternating MGCG_PLACEMENT = 0 / 3 / 7
# (run on the GPU box from the repo root).  Usage: bash conjugategradient_amd/tools/placement_ab.sh OUT.log [rounds]
OUT=${1:-gpurun_out/placement_ab.log}
ROUNDS=${2:-4}
: > "$OUT"
for i in $(seq 1 "$ROUNDS"); do
  for k in ${KS:-0 3}; do
    MGCG_PLACEMENT=$k python3 bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-60} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
p = d.get('placement_draw_rank0') or {}
print(json.dumps({'placement': $k, 'round': $i, 'it_per_s': round(d['value'], 2), 'spmv_in_loop_ms': round(d['roofline']['avg_launch_ms'], 4), 'frac': round(d['roofline']['frac'], 4),
                  'export_frac': round((d.get('roofline_csr_spmv') or {}).get('frac', 0), 4), 'ms_per_step': round(d['ms_per_step'], 4), 'draw': {k[:2]: [[round(t, 3) for t in v['candidates_spmv_ms']], v['chosen']] for k, v in p.items() if isinstance(v, dict)}}))" >> "$OUT"
    tail -1 "$OUT"
  done
done
