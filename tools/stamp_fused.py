"""Diagnostic: per-phase cycles of k_wgrad_adam's workgroups from s_memtime stamps (build -DMAMDR_STAMPS, never shipped).
usage: python tools/stamp_fused.py <libstamps.so> [shape] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from mamdr_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mamdr_amd import engine, synthetic
shape = sys.argv[2] if len(sys.argv) > 2 else "taobao10"
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
g = synthetic.generate(shape, batch_size=bs, seed=123)
eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
rs = np.random.RandomState(0)
eng.set_weights(torch.from_numpy((rs.standard_normal(eng.n_params) * 0.05).astype(np.float32)).to(eng.device))
n = eng.n_rows(d, "train")
stamps = torch.zeros(65536 + 8192, dtype=torch.int64, device=eng.device)
eng.lib.mamdr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
eng.lib.mamdr_debug_set_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
for _ in range(5):
    eng.train_steps(d, perm=perm, first_step=0, n_steps=3)
torch.cuda.synchronize()
ws = stamps.cpu().numpy()[65536:65536 + 8 * 242].reshape(242, 8)[:, :5].astype(np.float64)
t0 = ws[:, 0].min()
print("k_wgrad_adam stamps (cycles relative to the first workgroup's start): start | contraction begins | ends | barrier passed | done")
for name, sl in (("S workgroups", slice(0, 32)), ("tiles", slice(32, 240)), ("output unit", slice(240, 242))):
    w = np.diff(ws[sl], axis=1)
    print("  %-13s n=%3d  phase cycles: median %s   max %s   lifetime median %d max %d" % (
        name, len(w), np.round(np.median(w, axis=0)).astype(int).tolist(), np.round(w.max(axis=0)).astype(int).tolist(),
        np.median(ws[sl][:, 4] - ws[sl][:, 0]), (ws[sl][:, 4] - ws[sl][:, 0]).max()))
print("  (phases: prologue | contraction | wait at the barrier | reduce + optimiser step)")
