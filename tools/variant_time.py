"""Exploratory: time the step kernels with an alternative build of the library.
usage: python tools/variant_time.py <libvariant.so> [shape] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mamdr_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from mamdr_amd import engine, synthetic, _lib as L
shape = sys.argv[2] if len(sys.argv) > 2 else "taobao30"
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
g = synthetic.generate(shape, batch_size=bs, seed=123)
eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
rs = np.random.RandomState(0)
eng.set_weights(torch.from_numpy((rs.standard_normal(eng.n_params) * 0.05).astype(np.float32)).to(eng.device))
n = eng.n_rows(d, "train"); steps = -(-n // bs)
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
for _ in range(3): eng.train_steps(d, perm=perm)
torch.cuda.synchronize()
t = time.time(); reps = 20
for _ in range(reps): eng.train_steps(d, perm=perm)
torch.cuda.synchronize(); dt = time.time() - t
eng.profile(True); eng.profile_reset()
for _ in range(5): eng.train_steps(d, perm=perm)
ks = []
for k in range(3):
    ms, cnt = eng.profile_read(k); ks.append("%s %.1f" % (L.KERNEL_NAMES[k], ms / max(cnt, 1) * 1e3))
print("%s %s bs %d: %.1f us/step | %s" % (os.path.basename(sys.argv[1]), shape, bs, dt / (reps * steps) * 1e6, ", ".join(ks)))
