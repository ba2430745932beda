#!/bin/bash
# environment-switch sweeps on the BASELINE workloads: tools/r03_sweeps.sh <tag>
TAG=${1:-r03s}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
run() {   # name, workload, steps, env...
    local name=$1 wl=$2 steps=$3; shift 3
    env "$@" python bench.py --workload $wl --steps $steps --warmup 1 --cpu-budget 0 --no-targets --no-profile > "$OUT/$name.json" 2> "$OUT/$name.err"
    python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-28s %9.0f steps/s  %7.2f us/step" % (sys.argv[2], d["value"], d["us_per_domain_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run t30_default taobao30 6 A=1
run t30_rpg512 taobao30 6 MAMDR_RPG=512
run t30_rpg128 taobao30 6 MAMDR_RPG=128
run t30_fused2 taobao30 6 MAMDR_FUSED=2
run t30_tile4 taobao30 6 MAMDR_TOWER_TILE=4
run t10_default taobao10 10 A=1
run t10_fused0 taobao10 10 MAMDR_FUSED=0
run a13_default amazon13 1 A=1
run a13_flush16 amazon13 1 MAMDR_LAZY_FLUSH_EVERY=16
run a13_flush64 amazon13 1 MAMDR_LAZY_FLUSH_EVERY=64
run a13_flush128 amazon13 1 MAMDR_LAZY_FLUSH_EVERY=128
run a6_default amazon6 1 A=1
run a6_flush16 amazon6 1 MAMDR_LAZY_FLUSH_EVERY=16
run a6_flush64 amazon6 1 MAMDR_LAZY_FLUSH_EVERY=64
