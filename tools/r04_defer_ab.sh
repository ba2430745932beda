#!/bin/bash
# A/B of the queued weight gradients of the generic-layer engine (one pair of launches per step instead of one per layer)
mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_mtl.py tests/test_gpu_fmnets.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r04d/pytest.txt
for rep in 1 2; do
MAMDR_GRAPH_NO_DEFER=1 python tools/graph_bench.py 3 shared_bottom,mmoe,ple,ccpm,autoint > gpurun_out/r04d/nodefer_$rep.jsonl 2>/dev/null
python tools/graph_bench.py 3 shared_bottom,mmoe,ple,ccpm,autoint > gpurun_out/r04d/defer_$rep.jsonl 2>/dev/null
done
MAMDR_GRAPH_WQ_BLOCKS=256 python tools/graph_bench.py 3 shared_bottom,mmoe,ple > gpurun_out/r04d/defer_b256.jsonl 2>/dev/null
MAMDR_GRAPH_WQ_BLOCKS=1024 python tools/graph_bench.py 3 shared_bottom,mmoe,ple > gpurun_out/r04d/defer_b1024.jsonl 2>/dev/null
cat gpurun_out/r04d/pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04d/*.jsonl')):
    print(f.split('/')[-1], ' '.join('%s %.1f us (%s)' % (j['tower'], j['us_per_domain_step'], j['roofline']['launches_per_step']) for j in map(json.loads, open(f))))
PY
