#!/bin/bash
exec < /dev/null
# round-4 A/B of the headline step's per-call duties (through gpurun): tools/r04_ab.sh <tag>
# each variant = the default bench line's timed region only (no CPU baseline, no targets), 3 repeats interleaved
TAG=${1:-r04c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
run() {   # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step')" >> "$OUT/ab.txt"
}
for rep in 1 2 3; do
    run default A=1
    run no_pregather MAMDR_NO_PREGATHER=1
    run dm_each MAMDR_DM_EACH=1
    run no_pregather_dm_each MAMDR_NO_PREGATHER=1 MAMDR_DM_EACH=1
done
cat "$OUT/ab.txt"
