#!/bin/bash
# same-box A/B of k_emb_flush: the library vs the diagnostic build with the alphas read from global memory
D=gpurun_out/${OUTDIR:-r04n}
mkdir -p $D
for rep in 1 2; do
for V in lds base; do
  L=""; [ $V = base ] && L=$PWD/mamdr_amd/build/variants/libflushbase.so
for W in amazon6 amazon13; do
  MAMDR_LIB_PATH=$L python bench.py --workload $W --steps 1 --warmup 1 --cpu-budget 0 --no-targets 2>/dev/null | tail -1 > $D/bench_${V}_${W}_$rep.json
  python - $D/bench_${V}_${W}_$rep.json $V <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
k=j.get('kernels_avg_us') or {}
print(sys.argv[2], j['config']['workload'][:30], 'steps/s %.0f' % j['value'], 'us/step %.2f' % (1e6/j['value']), ['%s %.1f' % (n[:20], v['avg_us']) for n,v in k.items() if 'flush' in n])
PY
done; done; done
