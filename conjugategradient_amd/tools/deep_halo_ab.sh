#!/bin/bash
# One rank's slab of BASELINE config 4 (512 x 512 x 64) on the several-ranks code path of a real one-rank RCCL communicator
# (MGCG_FORCE_MULTIRANK), fresh process per leg, alternating: the plain single-rank loop / per-sweep exchanges (MGCG_DEEP_HALO=0) /
# the deep-halo cycle (default).   bash conjugategradient_amd/tools/deep_halo_ab.sh OUT.log [rounds]
OUT=$1; ROUNDS=${2:-3}
: > "$OUT"
for r in $(seq 1 $ROUNDS); do
  for leg in "plain --force 0" "per_sweep" "deep"; do
    set -- $leg
    name=$1; shift
    env=""
    [ "$name" = per_sweep ] && env="MGCG_DEEP_HALO=0"
    echo "== $name (round $r)" >> "$OUT"
    env $env timeout -k 10 300 python3 conjugategradient_amd/tools/forced_path_run.py --solver mgcg --overlap 1 --steps 100 --repeats 3 "$@" 2>/dev/null | tail -n 1 >> "$OUT"
  done
done
python3 - "$OUT" <<'PY'
import json, sys
legs = {}
name = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        name = line.split()[1]
    elif line.startswith("{"):
        legs.setdefault(name, []).append(json.loads(line)["ms_per_iteration"])
med = {k: sorted(v)[len(v) // 2] for k, v in legs.items()}
print(json.dumps({"ms_per_iteration_median_of_processes": med, "all": legs,
                  "added_us_vs_plain": {k: 1e3 * (v - med.get("plain", 0.0)) for k, v in med.items() if k != "plain"}}))
PY
