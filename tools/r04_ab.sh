#!/bin/bash
exec < /dev/null
# round-4 A/B of the headline step's per-call duties (through gpurun): tools/r04_ab.sh <tag> [variants...]
# each variant = the default bench line's timed region (no CPU baseline, no targets) + its per-kernel pass, 2 repeats interleaved
TAG=${1:-r04c}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
run() {   # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step', {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})" >> "$OUT/ab.txt"
}
for rep in 1 2; do
    run default A=1
    run no_w2_direct MAMDR_NO_W2_DIRECT=1
    run dm_call MAMDR_DM_CALL=1
    run round3 MAMDR_NO_W2_DIRECT=1 MAMDR_DM_CALL=1 MAMDR_NO_PASS_WINDOW=1
    run no_window MAMDR_NO_PASS_WINDOW=1
done
cat "$OUT/ab.txt"
