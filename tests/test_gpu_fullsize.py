"""GPU parity at the metric's OWN workloads: one whole MAMDR meta-epoch, per-domain AUC within 1e-3.

BASELINE.json `metric` is quoted on Taobao-10 bs 1024 (configs[1]) and north_star's target on Taobao-30 bs 4096
(configs[3]).  tests/test_gpu_parity.py::test_mamdr_epoch_auc_parity shows the 1e-3 bar in miniature (4 domains,
256 rows); here the HIP path runs the workloads themselves -- every domain, full tables, full splits, the config's
sample_num / add_query_domain, the kernels those batch sizes select (bs 1024: k_pass_prep + k_tower4<.., W1L, PRE> +
k_wgrad_adam + the pending domain-table step; bs 4096: k_tower + k_wgrad + k_update) -- against the numpy oracle
(oracle/loops.mamdr_epoch, the restatement of model_zoo/mamdr.py:41-108 + specific_base_model.py:64-97) on the same
plan, shuffles and dropout masks.  Asserted: the traces are equal and |AUC_hip - AUC_oracle| <= 1e-3 on EVERY
domain's validation split with the merged weights theta + phi_d (north_star: "per-domain AUC matches the reference
within 1e-3").
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import auc as oauc          # noqa: E402
from oracle import loops as oloops      # noqa: E402
from oracle import outer as oouter      # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32


def build(shape, batch, seed=123):
    from mamdr_amd import engine, synthetic
    g = synthetic.generate(shape, batch_size=batch, seed=seed)
    rs = np.random.RandomState(1024)
    params = otower.init_params(rs, g["n_user"], g["n_item"], g["n_domain"])
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
    return g, params, engine


def oracle_epochs(g, params, batch, plans, perm_seeds, meta_lr, phis0, dropout_seed=1024):
    """the reference loop on the numpy oracle: -> (theta, phis, trace, seconds)."""
    from mamdr_amd import engine
    model = otower.OracleModel({k: (v if k in ("user_emb", "item_emb") else v.copy()) for k, v in params.items()},
                               dropout=0.5, lr=1e-3, dropout_seed=dropout_seed)
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(g["n_domain"])]
    theta = model.get_flat().copy()
    phis = [p.copy() for p in phis0]
    it = iter(perm_seeds)
    t0 = time.time()
    trace = []
    for plan in plans:
        trace += oloops.mamdr_epoch(model, theta, phis, g["data"]["train"], plan,
                                    lambda d: engine.shuffle_perm(sizes[d], 10000, next(it)), batch, meta_lr)
    return model, theta, phis, trace, time.time() - t0


def run_case(shape, batch, meta_lr, epochs=1, max_auc_diff=1e-3):
    from mamdr_amd import meta, plan as mplan
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    g, params, engine = build(shape, batch)
    D = g["n_domain"]
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5)
    eng.bind_table("user_emb", params["user_emb"])
    eng.bind_table("item_emb", params["item_emb"])
    for split in ("train", "val"):
        for d in range(D):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    sizes = [eng.n_rows(d, "train") for d in range(D)]
    # config/Taobao-10/deepctr_DN+DR.json: sample_num 5, add_query_domain, shuffled sequence
    planner = mplan.EpochPlanner(range(D), 5, True, True, seed=123)
    plans = [planner.next_epoch() for _ in range(epochs)]
    n_pass = sum(len(mplan.epoch_passes(p)) for p in plans)
    perm_seeds = [0x5eed0000 + k for k in range(n_pass)]
    names = otower.param_names(False)
    # phi_d starts as a second random initialisation of the whole model (mamdr.py:31-33)
    phis0 = []
    for d in range(D):
        p2 = otower.init_params(np.random.RandomState(2000 + d), 8, 8, D)
        phis0.append(otower.flatten(p2, names))
    model, theta_o, phis_o, trace_o, secs = oracle_epochs(g, params, batch, plans, perm_seeds, meta_lr, phis0,
                                                          eng.dropout_seed)

    def to_dev(flat):
        named, o = {}, 0
        for nme in names:
            sz = params[nme].size
            named[nme] = flat[o:o + sz]
            o += sz
        return eng.pack(named)

    theta_g = to_dev(otower.flatten(params, names))
    phis_g = [to_dev(p) for p in phis0]
    it = iter(perm_seeds)
    trace_g = []
    t0 = time.time()
    for plan in plans:
        trace_g += meta.mamdr_epoch(eng, theta_g, phis_g, plan,
                                    lambda d: engine.shuffle_perm(sizes[d], 10000, next(it)), batch, lr=1e-3,
                                    meta_lr=meta_lr)
    torch.cuda.synchronize()
    gsecs = time.time() - t0
    assert trace_g == trace_o
    n_steps = sum(t[2] for t in trace_g)
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == (1 if batch <= 1024 else 0)
    print("%s bs %d: %d domain-steps in %d passes; oracle %.1f s, hip %.2f s" % (shape, batch, n_steps, len(trace_g),
                                                                                 secs, gsecs))
    merged = eng.new_vector()
    worst, aucs = 0.0, []
    for d in range(D):
        eng.merge(merged, theta_g, phis_g[d], "plus")
        eng.set_weights(merged)
        _, auc_g = eng.evaluate(d, "val")
        model.set_flat(oouter.merge(theta_o, phis_o[d], "plus"))
        _, preds = model.evaluate(g["data"]["val"][d], batch)
        auc_o = float(oauc.auc500(g["data"]["val"][d]["label"], preds, batch))
        print("  domain %2d: val rows %6d  AUC hip %.5f oracle %.5f  diff %+.1e" % (d, len(preds), auc_g, auc_o,
                                                                                  auc_g - auc_o))
        worst = max(worst, abs(auc_g - auc_o))
        aucs.append(auc_o)
        assert abs(auc_g - auc_o) <= max_auc_diff, (d, auc_g, auc_o)
    # the comparison is made on a model that has learnt (predictions spread over the threshold grid)
    assert float(np.mean(aucs)) > 0.6, aucs
    print("  worst |dAUC| %.2e, mean oracle AUC %.4f" % (worst, float(np.mean(aucs))))
    eng.close()


def test_taobao10_bs1024_two_epochs_config_meta_lr():
    """BASELINE.json configs[1] as configured (config/Taobao-10/deepctr_DN+DR.json: meta_learning_rate 0.1): 10
    domains, 92,137 train rows, bs 1024 -> 2 x 1,2xx inner steps on the fused path (oracle AUC ~0.6-0.75 by then)."""
    run_case("taobao10", 1024, meta_lr=0.1, epochs=2)


def test_taobao10_bs1024_full_epoch_auc_parity():
    """the same workload with meta lr 0.5, so that ONE epoch already gives a trained model (oracle AUC 0.76-0.82:
    predictions spread over the 500 thresholds) -- the setting of the miniature test in tests/test_gpu_parity.py."""
    run_case("taobao10", 1024, meta_lr=0.5)


def test_taobao30_bs4096_full_epoch_auc_parity():
    """BASELINE.json configs[3] / north_star target: 30 domains, 394,805 train rows, bs 4096 (slab path), 1,501
    inner steps; meta lr 0.5 as above (at the config's 0.1 one epoch leaves the oracle at AUC ~0.57)."""
    run_case("taobao30", 4096, meta_lr=0.5)
