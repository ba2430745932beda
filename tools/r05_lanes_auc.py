"""round 5: what the lanes' semantics (the L-rank sharded epoch: per-lane Adam slots and shuffle streams, DN = sum of the lanes'
displacements) do to the trained model -- config/Taobao-10/deepctr_DN+DR.json AS CONFIGURED (full rows, bs 1,024, patience 3;
epoch capped at 30) through run.py's whole pipeline (train -> early stop -> best state -> test -> finetune) on 1 / 2 / 4 lanes,
three dataset seeds.  python tools/r05_lanes_auc.py"""
import contextlib
import copy
import io
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamdr_amd import cli        # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = json.load(open(os.path.join(root, "config", "Taobao-10", "deepctr_DN+DR.json")))
for seed in (123, 7, 2024):
    ref = None
    for lanes in (1, 2, 4):
        cfg = copy.deepcopy(base)
        tmp = tempfile.mkdtemp()
        cfg["train"].update(epoch=30, lanes=lanes, result_save_path=tmp + "/r", checkpoint_path=tmp + "/c")
        cfg["dataset"]["seed"] = seed
        built = []
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            res = cli.main(cfg, on_model=built.append)
        m = built[0]
        epochs = len([t for t in m.trace if t[0] == "dn"])
        if lanes == 1:
            ref = res[3]
        print("seed %4d lanes %d: avg test AUC after finetune %.4f (per domain min %.4f max %.4f), avg loss %.4f, largest per-domain "
              "difference from the single chain %.4f, %.1f s" % (
                  seed, lanes, res[1], min(res[3].values()), max(res[3].values()), res[0],
                  max(abs(res[3][d] - ref[d]) for d in ref), time.time() - t0), flush=True)
