#!/bin/bash
exec < /dev/null
# A/B of what a lane runs (round 6): the k_wgrad_adam path (default) against the slab path (MAMDR_FUSED=0: tower -> k_wgrad ->
# k_update) and the lane count, Taobao-10 bs 1,024: tools/r06_lanes_paths.sh <tag>  ->  gpurun_out/<tag>/lanes_paths.txt
TAG=${1:-r06_lanes}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; RES=$OUT/lanes_paths.txt; : > "$RES"
run() {   # <label> <workload> <lanes> [env...]
    local label=$1 wl=$2 lanes=$3; shift 3
    env "$@" timeout 300 python bench.py --workload $wl --no-targets --cpu-budget 0 --no-profile --steps 10 --warmup 3 --lanes $lanes \
        > "$OUT/b.json" 2> "$OUT/b.err"
    python - "$label" "$wl" "$lanes" "$OUT/b.json" >> "$RES" <<'PY'
import json, sys
label, wl, lanes, path = sys.argv[1:5]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    l = d.get("lanes") or {}
    print("%-34s %-9s lanes %s: single chain %8.0f | lanes %8.0f domain-steps/s (%.2fx)%s" % (
        label, wl, lanes, d["value"], l.get("value", float("nan")), l.get("over_single_chain", float("nan")),
        "  ERROR " + l["error"] if "error" in l else ""))
except Exception as e:
    print("%-34s %-9s lanes %s: FAILED %s" % (label, wl, lanes, e))
PY
    tail -1 "$RES"
}
for rep in 1 2; do
  run "k_wgrad_adam path (default)"       taobao10 4 MAMDR_FUSED=1
  run "slab path"                         taobao10 4 MAMDR_FUSED=0
  run "slab path, 4-row tower"            taobao10 4 MAMDR_FUSED=0 MAMDR_TOWER_TILE=4
  run "slab path"                         taobao10 6 MAMDR_FUSED=0
  run "k_wgrad_adam path"                 taobao10 6 MAMDR_FUSED=1
  run "slab path"                         taobao10 8 MAMDR_FUSED=0
  run "k_wgrad_adam path, 3 lanes"        taobao10 3 MAMDR_FUSED=1
  run "k_wgrad_adam path, 5 lanes"        taobao10 5 MAMDR_FUSED=1
done
cat "$RES"
