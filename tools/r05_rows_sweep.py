"""Round 5: what an 8-row tower tile could buy on Taobao-30 bs 4,096.  Times single inner steps of R rows
on the slab path (the engine of the bs-4,096 workload) with the tower the library picks, and with MAMDR_TOWER_TILE=4 / 16
forced (one process per setting: the switch is read at mamdr_create).  Usage: python tools/r05_rows_sweep.py [tile]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] in ("4", "16"):
    os.environ["MAMDR_TOWER_TILE"] = sys.argv[1]
import numpy as np      # noqa: E402
import torch            # noqa: E402

import bench            # noqa: E402
from mamdr_amd import _lib as L, engine, synthetic      # noqa: E402

bs = 4096
g = synthetic.generate("taobao30", batch_size=bs, seed=123, splits=("train",))
params = bench.init_params(g, 0)
eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
eng.bind_table("user_emb", g["tables"]["user_emb"])
eng.bind_table("item_emb", g["tables"]["item_emb"])
d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
c = g["data"]["train"][d]
eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
eng.set_weights(eng.pack(params))
n = eng.n_rows(d, "train")
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
sizes = sorted(-(-v["uid"].shape[0] % bs) or bs for v in g["data"]["train"].values())
last = [v["uid"].shape[0] % bs or bs for v in g["data"]["train"].values()]
print("tile %s; last-batch rows of the 30 domains: %s" % (os.environ.get("MAMDR_TOWER_TILE", "auto"), sorted(last)))
for rows in (512, 1024, 1280, 1536, 2048, 2560, 3072, 4096):
    # a pass of exactly `rows` rows = one step of that size (steps of a call share one size except the last)
    for _ in range(20):
        eng.train_steps(d, perm=perm, n_steps=1, batch_size=rows, pass_rows=rows)
    torch.cuda.synchronize()
    eng.profile(True)
    eng.profile_reset()
    t = time.time()
    reps = 200
    for _ in range(reps):
        eng.train_steps(d, perm=perm, n_steps=1, batch_size=rows, pass_rows=rows)
    torch.cuda.synchronize()
    dt = time.time() - t
    ks = []
    for k in range(3):
        ms, cnt = eng.profile_read(k)
        ks.append("%s %.2f" % (L.KERNEL_NAMES[k], ms / max(cnt, 1) * 1e3))
    eng.profile(False)
    print("rows %5d: tower %s | %s" % (rows, eng.step_kernel_names(rows)[L.KERNEL_FWD_BWD], " | ".join(ks)), flush=True)
