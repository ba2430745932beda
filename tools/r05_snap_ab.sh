#!/bin/bash
exec < /dev/null
# round-5 A/B (through gpurun): the W0[256:384] snapshot of k_tower4 taken at the END of the kernel (default since round 5) vs at
# its start (build variant -DMAMDR_T4_SNAP_EARLY = rounds 2 - 4: tools/build_variant.sh snap_early -DMAMDR_T4_SNAP_EARLY), headline
# workload, 3 repeats interleaved
TAG=${1:-r05p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
run() {
    local name=$1; shift
    env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step', {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})" >> "$OUT/ab_snapshot.txt"
}
for rep in 1 2 3; do
    run snapshot_at_end A=1
    run snapshot_at_start MAMDR_LIB_PATH=$PWD/mamdr_amd/build/variants/libsnap_early.so
done
cat "$OUT/ab_snapshot.txt"
