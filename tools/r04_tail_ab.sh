#!/bin/bash
mkdir -p gpurun_out/${OUTDIR:-r04f}
python -m pytest tests/test_gpu_mtl.py tests/test_gpu_fmnets.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/${OUTDIR:-r04f}/pytest.txt
cat gpurun_out/${OUTDIR:-r04f}/pytest.txt
MAMDR_GRAPH_NO_DEFER=1 python -m pytest tests/test_gpu_mtl.py tests/test_gpu_fmnets.py -m gpu -q -x -k "one_step or uncertainty" 2>&1 | tail -2
python tools/graph_bench.py 3 shared_bottom,mmoe,ple,ccpm,autoint 2>/dev/null | grep tower > gpurun_out/${OUTDIR:-r04f}/tail.jsonl
MAMDR_PNN_ENGINE=graph MAMDR_NFM_ENGINE=graph python tools/graph_bench.py 3 nfm,pnn 2>/dev/null | grep tower > gpurun_out/${OUTDIR:-r04f}/twins.jsonl
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/${OUTDIR:-r04f}/*.jsonl')):
    out=[]
    for l in open(f):
        try: j=json.loads(l)
        except Exception: continue
        out.append('%s %.1f (%s)' % (j['tower'], j['us_per_domain_step'], j['roofline']['launches_per_step']))
    print('%-20s' % f.split('/')[-1], ' | '.join(out))
PY
