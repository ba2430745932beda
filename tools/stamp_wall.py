"""Diagnostic: wall-clock (s_memrealtime, 100 MHz, device-wide) start / end of every workgroup of the tower kernel and of
k_wgrad_adam in the LAST step of a short run -> gaps between the kernels and the spread inside them.
usage: python tools/stamp_wall.py <libstamps.so> [shape] [batch]   (MAMDR_FUSED / MAMDR_DM_EACH honoured)"""
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
for _ in range(4):
    eng.train_steps(d, perm=perm, first_step=0, n_steps=6)
torch.cuda.synchronize()
st = stamps.cpu().numpy()
tiles = bs // 4 if bs <= 2048 else bs // 16
ev = []
for par in (0, 1):
    tw = st[par * 16384:par * 16384 + tiles * 16].reshape(tiles, 16)[:, 10:12].astype(np.float64) * 10.0          # ns
    fz = st[65536 + par * 4096:65536 + par * 4096 + 8 * 242].reshape(242, 8)[:, 5:7].astype(np.float64) * 10.0
    fz = fz[fz[:, 0] > 0]
    if tw[:, 0].min() > 0:
        ev.append(("tower", tw[:, 0].min(), tw[:, 0].max(), tw[:, 1].min(), tw[:, 1].max()))
    if len(fz):
        ev.append(("k_wgrad_adam", fz[:, 0].min(), fz[:, 0].max(), fz[:, 1].min(), fz[:, 1].max()))
ev.sort(key=lambda e: e[1])
t0 = ev[0][1]
print("last two steps, wall clock (ns after the first workgroup start): first start | last start | first end | last end")
prev_end = None
for name, a, b, c_, d_ in ev:
    gap = "" if prev_end is None else "   gap after the previous kernel's last end: %.0f" % (a - prev_end)
    print("  %-13s %7.0f %7.0f %7.0f %7.0f%s" % (name, a - t0, b - t0, c_ - t0, d_ - t0, gap))
    prev_end = d_
