"""Diagnostic: per-phase cycles of k_tower<train, 384> (Star) from s_memtime stamps (-DMAMDR_STAMPS build).
usage: python tools/stamp_star.py [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mamdr_amd import build as B
so = os.path.join(ROOT, "mamdr_amd", "build", "libmamdr_hip_stamps.so")
srcs = [os.path.join(B.CSRC, s) for s, _ in B.SOURCES]
extra = os.environ.get("MAMDR_DIAG_FLAGS", "").split()
subprocess.check_call([B._hipcc()] + B.COMMON + ["-DMAMDR_STAMPS"] + extra + ["-shared", "-o", so] + srcs,
                      stderr=subprocess.DEVNULL)
from mamdr_amd import _lib
_lib.LIB_PATH = so
from mamdr_amd import engine, synthetic
from mamdr_amd.model_zoo.star import initial_tensors
import ctypes as C
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
trainable = os.environ.get("STAR_FROZEN") != "1"
g = synthetic.generate("amazon13", batch_size=bs, seed=123, row_scale=0.1)
D = g["n_domain"]
eng = engine.TowerEngine(g["n_user"], g["n_item"], D, bs, dropout=0.5, emb_trainable=trainable, tower="star")
if not trainable:
    eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
d = max(range(D), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
p = initial_tensors(np.random.RandomState(1), g["n_user"], g["n_item"], D, 128, (256, 128, 64),
                    None if trainable else g["tables"]["user_emb"], None if trainable else g["tables"]["item_emb"])
eng.set_weights(eng.pack(p))
n = eng.n_rows(d, "train")
tiles = bs // 16
stamps = torch.zeros(65536 + 8192, dtype=torch.int64, device=eng.device)
eng.lib.mamdr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
eng.lib.mamdr_debug_set_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
for _ in range(3):
    eng.train_steps(d, perm=perm, first_step=0, n_steps=2)
torch.cuda.synchronize()
st = stamps.cpu().numpy()[:tiles * 16].reshape(tiles, 16)[:, :10].astype(np.float64)
names = ["w0 prefetch+gather(+PN affine)", "L0 fwd", "L1 fwd (incl. barrier)", "L2 fwd", "barrier+out/loss",
         "bw2 prefetch+barrier+dz3", "bwd2 (dz2)", "bwd1 (dz1)", "bwd0 (dx, 384 wide)"]
dif = np.diff(st, axis=1)
tot = st[:, 9] - st[:, 0]
print("star tower, %d rows, %d tiles, trainable=%s; total cycles median %.0f (min %.0f max %.0f)" %
      (bs, tiles, trainable, np.median(tot), tot.min(), tot.max()))
for i, nme in enumerate(names):
    print("  %-32s %8.0f  (%4.1f%%)" % (nme, np.median(dif[:, i]), 100 * np.median(dif[:, i]) / np.median(tot)))
print("first start -> last end: %.0f cycles" % (st[:, 9].max() - st[:, 0].min()))
