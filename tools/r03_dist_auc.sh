#!/bin/bash
# per-domain test AUC of mlp_meta_mamdr on Taobao-10 (6 epochs) at N = 1, 2, 4, 8 ranks SHARING the one GPU over gloo,
# for both DN modes, and at N = 1 for four more run seeds (same generated logs): tools/r03_dist_auc.sh <tag>
TAG=${1:-r03d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export MAMDR_SHARE_GPU=1
CFG=config/Taobao-10/deepctr_DN+DR.json
for S in 123 124 125 126 127; do
    python tools/dist_auc.py $CFG 6 sharded $S 2>/dev/null | grep DISTAUC >> "$OUT/dist_auc.jsonl"
done
P=29610
for MODE in sharded replicated; do
    for N in 2 4 8; do
        P=$((P + 1))
        python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P tools/dist_auc.py $CFG 6 $MODE 123 2>"$OUT/err_${MODE}_$N.log" | grep DISTAUC >> "$OUT/dist_auc.jsonl"
    done
done
python - "$OUT/dist_auc.jsonl" <<'PY'
import json, sys
import numpy as np
rows = [json.loads(l.split("DISTAUC ", 1)[1]) for l in open(sys.argv[1])]
base = next(r for r in rows if r["world"] == 1 and r["seed"] == 123)
b = np.array(base["domain_auc"])
for r in rows:
    d = np.array(r["domain_auc"]) - b
    print("world %d %-10s seed %d: avg AUC %.5f  max |per-domain shift vs N=1 seed 123| %.4f  mean shift %+.4f" % (
        r["world"], r["dn_mode"], r["seed"], r["avg_auc"], np.abs(d).max(), d.mean()))
PY
