#!/bin/bash
# A/B of the 32 x 32 split-K tiles of the generic-layer engine's forward / d-input contractions
mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_mtl.py tests/test_gpu_fmnets.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r04e/pytest.txt
cat gpurun_out/r04e/pytest.txt
for T in 0 256 1024 100000; do
MAMDR_GRAPH_TILE32_BELOW=$T python tools/graph_bench.py 3 shared_bottom,mmoe,ple,ccpm,autoint 2>/dev/null | grep tower > gpurun_out/r04e/below_$T.jsonl
done
MAMDR_PNN_ENGINE=graph MAMDR_NFM_ENGINE=graph python tools/graph_bench.py 3 nfm,pnn 2>/dev/null | grep tower > gpurun_out/r04e/twins.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04e/*.jsonl')):
    out=[]
    for l in open(f):
        try: j=json.loads(l)
        except Exception: continue
        out.append('%s %.1f (%s)' % (j['tower'], j['us_per_domain_step'], j['roofline']['launches_per_step']))
    print('%-20s' % f.split('/')[-1], ' | '.join(out))
PY
