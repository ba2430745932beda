#!/bin/bash
exec < /dev/null
# round-5 A/B (through gpurun): prefetch riders in k_wgrad_adam's launch touching the next tower launch's pre-gathered rows
# (MAMDR_FUSED_PF=1 = on; default off since this A/B), headline workload, 3 repeats interleaved
TAG=${1:-r05j}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
run() {
    local name=$1; shift
    env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step', {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})" >> "$OUT/ab_fused_pf.txt"
}
for rep in 1 2 3; do
    run riders_on MAMDR_FUSED_PF=1
    run riders_off A=1
done
cat "$OUT/ab_fused_pf.txt"
