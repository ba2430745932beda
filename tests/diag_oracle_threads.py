"""probe: how the numpy oracle's step time depends on the BLAS thread count on this host (round 5: the GPU box has 256
hardware threads and the oracle's small GEMMs ran 3-4x slower there than on 8 cores).  Prints ms / step."""
import os
import sys
import time

import numpy as np
from threadpoolctl import threadpool_info, threadpool_limits

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamdr_amd import synthetic      # noqa: E402
from oracle import tower as ot       # noqa: E402

print("cpus", os.cpu_count(), [(i["internal_api"], i["num_threads"]) for i in threadpool_info()])
g = synthetic.generate("taobao10", batch_size=1024, seed=123)
p = ot.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], g["n_domain"])
p["user_emb"], p["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
m = ot.OracleModel(p, dropout=0.5, lr=1e-3)
d = g["data"]["train"][0]
perm = np.arange(d["uid"].shape[0])
for lim in (None, 1, 4, 8, 16, 32, 64):
    for B in (1024, 4096):
        n = 12 if B == 1024 else 4
        if lim is None:
            m.train_pass(d, np.tile(perm, 4), B, max_steps=2)
            t = time.time(); m.train_pass(d, np.tile(perm, 4), B, max_steps=n); dt = time.time() - t
        else:
            with threadpool_limits(limits=lim):
                m.train_pass(d, np.tile(perm, 4), B, max_steps=2)
                t = time.time(); m.train_pass(d, np.tile(perm, 4), B, max_steps=n); dt = time.time() - t
        print("threads %s bs %d: %.1f ms/step" % (lim, B, dt / n * 1e3), flush=True)
