#!/bin/bash
exec < /dev/null
# A/B of CU-partitioned lane streams (MAMDR_LANE_CUS, mamdr_stream_create_masked): tools/r06_lanes_cus.sh <tag>
# every lane's stream confined to n CUs (consecutive CU numbers, or "<n>i" interleaved) against the free-for-all of round 5;
# Taobao-10 bs 1,024 and Taobao-30 bs 4,096, 4 lanes (and 2 lanes on halves of the device).  Output: gpurun_out/<tag>/lanes_cus.txt
TAG=${1:-r06_lanes}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; RES=$OUT/lanes_cus.txt; : > "$RES"
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
    print("%-28s %-9s lanes %s: single chain %8.0f | lanes %8.0f domain-steps/s (%.2fx)%s" % (
        label, wl, lanes, d["value"], l.get("value", float("nan")), l.get("over_single_chain", float("nan")),
        "  ERROR " + l["error"] if "error" in l else ""))
except Exception as e:
    print("%-28s %-9s lanes %s: FAILED %s" % (label, wl, lanes, e))
PY
    tail -1 "$RES"
}
for rep in 1 2; do
  run "free-for-all (r05)"     taobao10 4 MAMDR_LANE_CUS=0
  run "64 consecutive CUs"     taobao10 4 MAMDR_LANE_CUS=64
  run "64 interleaved CUs"     taobao10 4 MAMDR_LANE_CUS=64i
  run "128 consecutive (ovl)"  taobao10 4 MAMDR_LANE_CUS=128
  run "2 lanes free-for-all"   taobao10 2 MAMDR_LANE_CUS=0
  run "2 lanes x 128 consec."  taobao10 2 MAMDR_LANE_CUS=128
  run "2 lanes x 128 interl."  taobao10 2 MAMDR_LANE_CUS=128i
  run "2 x 128i, 16-row tower"  taobao10 2 MAMDR_LANE_CUS=128i MAMDR_TOWER_TILE=16
done
run "free-for-all (r05)"     taobao30 4 MAMDR_LANE_CUS=0
run "64 consecutive CUs"     taobao30 4 MAMDR_LANE_CUS=64
run "64 interleaved CUs"     taobao30 4 MAMDR_LANE_CUS=64i
run "2 lanes x 128 interl."  taobao30 2 MAMDR_LANE_CUS=128i
cat "$RES"
