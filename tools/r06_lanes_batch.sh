#!/bin/bash
exec < /dev/null
# lanes with batched step launches (mamdr_group_*, MAMDR_LANES_BATCH=1) against lanes on streams of their own:
# tools/r06_lanes_batch.sh <tag>  ->  gpurun_out/<tag>/lanes_batch.txt
TAG=${1:-r06_lanes}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; RES=$OUT/lanes_batch.txt; : > "$RES"
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
    sl = l.get("step_launches") or {}
    print("%-34s %-9s lanes %s: single chain %8.0f | lanes %8.0f domain-steps/s (%.2fx)%s%s" % (
        label, wl, lanes, d["value"], l.get("value", float("nan")), l.get("over_single_chain", float("nan")),
        "  %.2f steps per launch" % sl["lanes_per_launch"] if sl else "", "  ERROR " + l["error"] if "error" in l else ""))
except Exception as e:
    print("%-34s %-9s lanes %s: FAILED %s" % (label, wl, lanes, e))
PY
    tail -1 "$RES"
}
for rep in 1 2; do
  run "streams of their own (r05)"    taobao10 4 MAMDR_LANES_BATCH=0
  run "batched + held"                taobao10 4 MAMDR_LANES_BATCH=1 MAMDR_LANES_HOLD=1
  run "batched + held, wait 10 us"    taobao10 4 MAMDR_LANES_BATCH=1 MAMDR_LANES_HOLD=1 MAMDR_GROUP_WAIT_US=10
  run "batched + held, wait 25 us"    taobao10 4 MAMDR_LANES_BATCH=1 MAMDR_LANES_HOLD=1 MAMDR_GROUP_WAIT_US=25
  run "batched + held, wait 60 us"    taobao10 4 MAMDR_LANES_BATCH=1 MAMDR_LANES_HOLD=1 MAMDR_GROUP_WAIT_US=60
  run "batched, wait 10 us (no hold)" taobao10 4 MAMDR_LANES_BATCH=1 MAMDR_GROUP_WAIT_US=10
done
cat "$RES"
