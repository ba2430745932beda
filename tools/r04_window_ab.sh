#!/bin/bash
exec < /dev/null
# A/B of the pass window's row budget on the headline workload (through gpurun): tools/r04_window_ab.sh <tag>
TAG=${1:-r04w}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for rep in 1 2; do
  for rows in 24576 49152 98304 196608 393216; do
    MAMDR_PASS_WINDOW_ROWS=$rows timeout 300 python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('window_rows $rows', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step', {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})" >> "$OUT/window_ab.txt"
  done
done
cat "$OUT/window_ab.txt"
