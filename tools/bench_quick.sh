#!/bin/bash
# quick timing of one bench workload without the CPU baseline: tools/bench_quick.sh <workload> [steps]
python bench.py --workload $1 --steps ${2:-2} --warmup 1 --cpu-budget 0 > gpurun_out/q_$1.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/q_$1.json')); print('$1', round(d['value'],1), round(d['us_per_domain_step'],1), d['roofline']['kernel'], round(d['roofline']['achieved'],1), {k:round(v['avg_us'],1) for k,v in d['kernels_avg_us'].items()})"
