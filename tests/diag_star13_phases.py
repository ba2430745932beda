"""diagnostic (GPU; run by hand: python tests/diag_star13_phases.py -- it uses the oracle, so it lives under tests/): the Star / Amazon-13 full-table MAMDR epoch of tests/test_gpu_fullsize.py, compared with the oracle
after EVERY phase (DN, then each DR query): per-domain val AUC of theta + phi_d on both sides and the relative L2
distance of theta / phi / the live tail.  Localises where a per-domain AUC difference comes from."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mamdr_amd import engine, meta, synthetic          # noqa: E402
from oracle import auc as oauc, loops as oloops, outer as oouter, star as ostar   # noqa: E402
from test_gpu_fullsize import _StarMeta, _bind_splits, _perm_stream   # noqa: E402

batch = 8192
meta_lr = float(os.environ.get("DIAG_META_LR", "0.5"))
shape = synthetic.SHAPES["amazon13"]
g = synthetic.generate("amazon13", batch_size=batch, seed=123, row_scale=90000 * 13 / shape["n_train"] / 3,
                       splits=("train", "val"), hot=dict(users=3000, items=1500, share=0.8))
D = g["n_domain"]
all_sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
doms = sorted(range(D), key=lambda d: -all_sizes[d])[:4]
params = ostar.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], D)
# PartitionedNorm's gamma / beta and the biases off their special initial values (1 / 0), as in
# tests/test_gpu_parity.py::make_star_problem: with beta = 0 the normalised domain columns (constant over a
# single-domain batch) are pure rounding residue, their kernel rows' gradients are noise that Adam normalises to
# steps of +- lr -- a random walk that differs between any two fp32 evaluations (measured with this test: the tensors
# outside theta / phi 0.8 % apart after one epoch, one domain's AUC 2e-3 off, tools/diag/star13_phases.py)
irs = np.random.RandomState(7)
for n_ in ("pn_gamma_shared", "pn_gamma_spec"):
    params[n_] = (params[n_] + irs.standard_normal(params[n_].shape) * 0.2).astype(np.float32)
for n_ in ("pn_beta_shared", "pn_beta_spec", "bs0", "bs1", "bs2", "bd0", "bd1", "bd2", "gb"):
    params[n_] = (irs.standard_normal(params[n_].shape) * 0.05).astype(np.float32)
eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.0, emb_trainable=True, tower="star")
_bind_splits(eng, g, doms)
eng.set_weights(eng.pack(params))
model = ostar.OracleStar(params, emb_trainable=True, lr=1e-3)
wrapped = _StarMeta(model)
theta_o = wrapped.get_flat().copy()
prs = np.random.RandomState(3)
plan = {"seq": [doms[i] for i in prs.permutation(4)], "dr": []}
for q in [doms[i] for i in prs.permutation(4)]:
    plan["dr"].append((q, [doms[i] for i in prs.permutation(4) if doms[i] != q][:2] + [q]))
print("plan", plan, flush=True)
phis_o = {d: np.zeros_like(theta_o) for d in doms}
theta_g = torch.from_numpy(theta_o).to(eng.device)
phis_g = {d: eng.new_vector(meta=True) for d in doms}
pf_o, pf_g = _perm_stream(all_sizes, 900), _perm_stream(all_sizes, 900)
merged_g = eng.new_vector(meta=True)


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def report(tag):
    live_o = model.get_flat()
    live_g = eng.get_weights().cpu().numpy()[:live_o.size]
    keep_o = wrapped.get_flat().copy()
    keep_g = eng.get_weights()
    line = []
    for d in doms:
        eng.merge(merged_g, theta_g, phis_g[d], "plus")
        eng.set_weights(merged_g)
        _, auc_g = eng.evaluate(d, "val")
        wrapped.set_flat(oouter.merge(theta_o, phis_o[d], "plus"))
        _, preds = model.evaluate(g["data"]["val"][d], batch)
        auc_o = float(oauc.auc500(g["data"]["val"][d]["label"], preds, batch))
        line.append("d%d %+.1e (%.4f)" % (d, auc_g - auc_o, auc_o))
    wrapped.set_flat(keep_o)
    eng.set_weights(keep_g)
    n_meta = theta_o.size
    print("%-14s AUC hip-oracle: %s | rel L2: theta %.1e live-meta %.1e live-tail %.1e phi %s" % (
        tag, "  ".join(line), rel(theta_g.cpu().numpy(), theta_o), rel(live_g[:n_meta], live_o[:n_meta]),
        rel(live_g[n_meta:], live_o[n_meta:]),
        " ".join("%.1e" % rel(phis_g[d].cpu().numpy(), phis_o[d]) for d in doms)), flush=True)


# DN phase (mamdr.py:48-57)
tr_o, tr_g = [], []
wrapped.set_flat(theta_o)
for d in plan["seq"]:
    oloops._pass(wrapped, g["data"]["train"], pf_o, d, batch, tr_o, "dn")
oouter.mamdr_update(theta_o, wrapped.get_flat(), theta_o, meta_lr)
eng.set_weights(theta_g)
for d in plan["seq"]:
    meta.run_pass(eng, d, pf_g, batch, 1e-3, tr_g, "dn")
eng.interp(theta_g, eng.meta_weights, theta_g, meta_lr)
report("after DN")
for query, support in plan["dr"]:
    merged = oouter.merge(theta_o, phis_o[query], "plus")
    for j in support:
        wrapped.set_flat(merged)
        oloops._pass(wrapped, g["data"]["train"], pf_o, j, batch, tr_o, "dr_support")
        oloops._pass(wrapped, g["data"]["train"], pf_o, query, batch, tr_o, "dr_query")
        oouter.mamdr_update(phis_o[query], wrapped.get_flat(), merged, meta_lr)
        merged = oouter.merge(theta_o, phis_o[query], "plus")
    meta.dr_query(eng, theta_g, phis_g[query], query, support, pf_g, batch, 1e-3, meta_lr, tr_g, merged_g)
    assert tr_o == tr_g
    report("after DR q=%d" % query)
eng.close()
