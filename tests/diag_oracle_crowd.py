"""diagnostic (GPU box, run by hand): how the numpy oracle's step time scales with the number of concurrently running,
core-pinned worker processes -- the oracle job pool of tests/oracle_pool.py in miniature.
usage: python tests/diag_oracle_crowd.py [n_workers ...]"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = r'''
import os, sys, time
cores = [int(c) for c in sys.argv[1].split(",")]
os.sched_setaffinity(0, cores)
os.environ["OPENBLAS_NUM_THREADS"] = str(len(cores))
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from mamdr_amd import synthetic
from oracle import tower as ot
g = synthetic.generate("taobao10", batch_size=1024, seed=123, splits=("train",))
p = ot.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], g["n_domain"])
p["user_emb"], p["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
m = ot.OracleModel(p, dropout=0.5, lr=1e-3)
d = g["data"]["train"][5]
perm = np.arange(d["uid"].shape[0])
m.train_pass(d, perm, 1024, max_steps=5)
t = time.time(); m.train_pass(d, perm, 1024, max_steps=40); dt = time.time() - t
print("%%.2f" %% (dt / 40 * 1e3))
''' % (os.path.dirname(HERE), HERE)

sys.path.insert(0, HERE)
import oracle_pool      # noqa: E402
cores = oracle_pool.physical_cores()
free = [g[0] for g in cores[16:]]
MALLOC = {"MALLOC_MMAP_THRESHOLD_": str(32 << 20), "MALLOC_TRIM_THRESHOLD_": str(1 << 30), "MALLOC_TOP_PAD_": str(256 << 20)}
for n, tuned in [(int(a), t) for a in (sys.argv[1:] or ["1", "8", "16", "28"]) for t in (False, True)]:
    per = 4
    procs = []
    t0 = time.time()
    for i in range(n):
        blk = free[(i * per) % (len(free) - per + 1):][:per]
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ",".join(map(str, blk))], stdout=subprocess.PIPE,
                                      universal_newlines=True, env=dict(os.environ, **(MALLOC if tuned else {}))))
    outs = [float(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    print("%2d workers x %d cores, glibc malloc %s: oracle step %.1f .. %.1f ms (median %.1f); wall %.1f s" % (
        n, per, "keeps its heap (no mmap / munmap per temporary)" if tuned else "default", min(outs), max(outs), sorted(outs)[len(outs) // 2], time.time() - t0), flush=True)
