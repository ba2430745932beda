"""Diagnostic: per-phase cycle shares of k_tower<train> from s_memtime stamps.
Builds a separate library with -DMAMDR_STAMPS (never shipped) and prints the median
per-phase cycles over workgroups.  Usage: python tools/stamp_tower.py [shape] [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mamdr_amd import build as B
so = os.path.join(ROOT, "mamdr_amd", "build", "libmamdr_hip_stamps%s.so" % os.environ.get("STAMP_SO_SUFFIX", ""))
srcs = [os.path.join(B.CSRC, s) for s, _ in B.SOURCES]
extra = os.environ.get("MAMDR_DIAG_FLAGS", "").split()
if not (os.environ.get("MAMDR_STAMPS_PREBUILT") and os.path.exists(so)):       # (prebuilt in the build container: `--build-only`)
    subprocess.check_call([B._hipcc()] + B.COMMON + ["-DMAMDR_STAMPS"] + extra + ["-shared", "-o", so] + srcs)
if "--build-only" in sys.argv:
    sys.exit(0)
from mamdr_amd import _lib
_lib.LIB_PATH = so
from mamdr_amd import engine, synthetic
import ctypes as C
shape = sys.argv[1] if len(sys.argv) > 1 else "taobao10"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = synthetic.generate(shape, batch_size=bs, seed=123)
eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
rs = np.random.RandomState(0)
w = (rs.standard_normal(eng.n_params) * 0.05).astype(np.float32); eng.set_weights(torch.from_numpy(w).to(eng.device))
# the stamp build writes stamps through TowerArgs.pred_out, which only eval sets; train leaves it null.
# So use the eval kernel?  No: train path -- patch: engine passes no pred_out in train.  We therefore
# stamp the eval launch (forward phases) and, for train, rely on a debug env hook.
n = eng.n_rows(d, "train")
tiles = bs // (4 if os.environ.get('MAMDR_TOWER_TILE') != '16' else 16)
stamps = torch.zeros(65536 + 8192 + 4096, dtype=torch.int64, device=eng.device)
eng.lib.mamdr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
eng.lib.mamdr_debug_set_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
# STAMP_OPT=accumulate (round 5): the meta pass's steps -- gradients added to an accumulator, the weights NOT rewritten, so
# the next tower finds them where the previous one left them (its own XCD's L2) instead of fetching what k_wgrad_adam wrote
# on other XCDs a moment ago: the difference of the two timelines is the price of reading freshly updated weights
opt = os.environ.get("STAMP_OPT", "adam")
if opt == "accumulate":
    acc = eng.new_vector()
    eng.bind_accumulator(acc)
for _ in range(5):
    eng.train_steps(d, perm=perm, first_step=0, n_steps=3, optimizer=opt)
print("optimizer of the stamped steps: %s" % opt)
torch.cuda.synchronize()
allst = stamps.cpu().numpy()
st = allst[:tiles * 16].reshape(tiles, 16)[:, :10].astype(np.float64)
if os.environ.get('MAMDR_TOWER_TILE') != '16':
    names4 = ["prefetch+gather", "L0 contract", "x store+barrier+L0 epilogue", "L1 contract", "barrier+L1 epilogue",
              "L2 contract", "barrier+L2 epi+out/loss/dz3", "bwd2 contract+epilogue", "bwd1 contract+epilogue"]
names = ["w0 prefetch+gather", "L0 fwd", "L1 fwd (incl. barrier)", "L2 fwd", "barrier+out/loss", "bw2 prefetch+barrier+dz3",
         "bwd2 (dz2)", "bwd1 (dz1)", "bwd0 (dxe)"]
full = allst[:tiles * 16].reshape(tiles, 16).astype(np.float64)
if full[:, 12].min() > 0:
    g0 = full[:, 0]
    print("  gather detail (cycles after kernel start, median): bookkeeping barrier passed %.0f | pair sums staged %.0f | "
          "pending barrier passed %.0f | domain rows finished %.0f | gather done %.0f" % tuple(
              np.median(full[:, k] - g0) for k in (12, 13, 14, 15, 1)))
w4 = allst[256 * 16:(256 + tiles) * 16].reshape(tiles, 16).astype(np.float64)
if w4[:, 0].min() > 0:
    print("  wave 4 (cycles after kernel start, median): after the bookkeeping barrier %.0f | partials requested %.0f | "
          "at the gather-end barrier %.0f" % tuple(np.median(w4[:, k] - full[:, 0]) for k in (0, 1, 2)))
dif = np.diff(st, axis=1)
tot = st[:, 9] - st[:, 0]
print("tiles %d; total cycles median %.0f (min %.0f max %.0f); s_memtime ticks = shader cycles" % (tiles, np.median(tot), tot.min(), tot.max()))
for i, nme in enumerate(names4 if os.environ.get('MAMDR_TOWER_TILE') != '16' else names):
    print("  %-28s %8.0f  (%4.1f%%)" % (nme, np.median(dif[:, i]), 100 * np.median(dif[:, i]) / np.median(tot)))
span = st[:, 9].max() - st[:, 0].min()
print("first start -> last end: %.0f cycles" % span)

ws = allst[65536:65536 + 8 * 544].reshape(544, 8)[:, :5].astype(np.float64)
ws = ws[ws[:, 0] > 0]
wd = np.diff(ws, axis=1)
print("k_wgrad: %d tile workgroups stamped; lifetime median %.0f cycles" % (len(ws), np.median(ws[:, 4] - ws[:, 0])))
for i, nme in enumerate(["tile descriptor load", "row loads + MFMAs (wave 0)", "LDS write + barrier", "reduce + slab store"]):
    print("  %-28s %8.0f" % (nme, np.median(wd[:, i])))
print("  first start -> last end: %.0f cycles; start spread %.0f" % (ws[:, 4].max() - ws[:, 0].min(), ws[:, 0].max() - ws[:, 0].min()))

# ---- k_update (slab path): [workgroups][4] stamps (entry, operands summed, exit); workgroup kinds by index: the first 32
#      step W0[256:384] by linearity (frozen-table mlp tower), then the float4 workgroups, then the domain table's (D x 8)
us = allst[65536 + 8192:65536 + 8192 + 4096].reshape(1024, 4)[:, :3].astype(np.float64)
live = us[:, 0] > 0
if live.any():
    idx = np.nonzero(live)[0]
    t0 = us[live, 0].min()
    n_lin = 32
    n_vec = (139777 + 3) // 4 // 256 + 1          # float4 workgroups behind the domain table's elements (update_blocks)
    kinds = [("W0[256:384] by linearity", idx[idx < n_lin]), ("float4 elements", idx[(idx >= n_lin) & (idx < n_lin + n_vec)]),
             ("domain table", idx[idx >= n_lin + n_vec])]
    print("k_update: %d workgroups stamped; first start -> last exit %.0f cycles; start spread %.0f" % (
        len(idx), us[live, 2].max() - t0, us[live, 0].max() - t0))
    for nme, ii in kinds:
        if len(ii) == 0:
            continue
        a = us[ii]
        ok = a[:, 2] > 0
        mid = np.where(a[:, 1] > 0, a[:, 1], a[:, 0])
        print("  %-26s %4d workgroups: start %6.0f .. %6.0f after the first; operands summed after %6.0f (median); lifetime median "
              "%6.0f max %6.0f; last exit %6.0f" % (nme, len(ii), (a[:, 0] - t0).min(), (a[:, 0] - t0).max(), np.median(mid - a[:, 0]),
                                                  np.median(a[ok, 2] - a[ok, 0]), (a[ok, 2] - a[ok, 0]).max(), (a[ok, 2] - t0).max()))
