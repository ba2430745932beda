#!/bin/bash
# the table flush (k_emb_flush) after a change: the lazy == dense bitwise tests, then Amazon-6 / Amazon-13 full-row epochs
D=gpurun_out/${OUTDIR:-r04l}
mkdir -p $D
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -q -x -k "lazy or trainable or star or amazon or flush" 2>&1 | tail -3
for W in amazon6 amazon13; do
  python bench.py --workload $W --steps 1 --warmup 1 --cpu-budget 0 --no-targets 2>/dev/null | tail -1 > $D/bench_$W.json
  python - $D/bench_$W.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
k=j.get('kernels_avg_us') or {}
print(j['config']['workload'][:40], 'steps/s %.0f' % j['value'], 'us/step %.2f' % (1e6/j['value']), {n[:30]: (round(v,1) if isinstance(v,(int,float)) else v) for n,v in k.items() if 'flush' in n})
PY
done
