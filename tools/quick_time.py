"""Exploratory timing of the step kernels on one GPU (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mamdr_amd import engine, synthetic, _lib as L
import bench

shape = sys.argv[1] if len(sys.argv) > 1 else "taobao10"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = synthetic.generate(shape, batch_size=bs, seed=123)
params = bench.init_params(g, 0)
eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
for split in ("train", "val"):
    for d in range(g["n_domain"]):
        c = g["data"][split][d]; eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
eng.set_weights(eng.pack(params))
d = max(range(g["n_domain"]), key=lambda k: eng.n_rows(k, "train"))
n = eng.n_rows(d, "train"); steps = -(-n // bs)
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
for _ in range(3): eng.train_steps(d, perm=perm)
torch.cuda.synchronize()
t = time.time(); reps = 20
for _ in range(reps): eng.train_steps(d, perm=perm)
torch.cuda.synchronize(); dt = time.time() - t
print("domain %d rows %d steps/pass %d: %.1f us/step (%.0f steps/s)" % (d, n, steps, dt / (reps * steps) * 1e6, reps * steps / dt))
eng.profile(True); eng.profile_reset()
for _ in range(5): eng.train_steps(d, perm=perm)
for k in range(3):
    ms, cnt = eng.profile_read(k); print("  %-16s %8.2f us avg over %d" % (L.KERNEL_NAMES[k], ms / max(cnt, 1) * 1e3, cnt))
eng.profile(False)
t = time.time()
for dd in range(g["n_domain"]): eng.evaluate(dd, "val")
print("eval all val: %.2f ms" % ((time.time() - t) * 1e3))
