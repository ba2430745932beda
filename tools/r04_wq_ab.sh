#!/bin/bash
D=gpurun_out/${OUTDIR:-r04i}
mkdir -p $D
python -m pytest tests/test_gpu_mtl.py tests/test_gpu_fmnets.py -m gpu -q -x 2>&1 | tail -2
for B in 256 512 1024; do
MAMDR_GRAPH_WQ_BLOCKS=$B python tools/graph_bench.py 3 shared_bottom,mmoe,ple,ccpm,autoint 2>/dev/null | grep tower > $D/wq_$B.jsonl
done
python - $D <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.jsonl')):
    out=[]
    for l in open(f):
        try: j=json.loads(l)
        except Exception: continue
        out.append('%s %.1f (%s)' % (j['tower'], j['us_per_domain_step'], j['roofline']['launches_per_step']))
    print('%-20s' % f.split('/')[-1], ' | '.join(out))
PY
