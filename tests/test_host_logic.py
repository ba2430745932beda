"""Host logic on CPU: registry, wrappers, meta loops, planner, dataset formats, result files.
The tower is replaced by tests/fake_engine.FakeEngine (oracle-backed)."""
import copy
import json
import os
import re

import numpy as np
import pytest
import torch

from fake_engine import FakeEngine, MetaSubset
from mamdr_amd import cli, meta, plan as mplan, synthetic
from mamdr_amd.utils import dataset as mds
from oracle import loops as oloops
from oracle import rng as orng
from oracle import tower as otower

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def tiny_config(tmp_path, name="mlp_meta_mamdr_finetune", epochs=2):
    with open(os.path.join(ROOT, "config", "Taobao-10", "deepctr_DN+DR.json")) as f:
        cfg = json.load(f)
    cfg = copy.deepcopy(cfg)
    cfg["model"]["name"] = name
    cfg["model"]["hidden_dim"] = [16, 8, 4]
    for k in ("user_dim", "item_dim", "domain_dim"):
        cfg["model"][k] = 8
    cfg["train"].update(epoch=epochs, patience=1, sample_num=2, result_save_path=str(tmp_path / "result"),
                        checkpoint_path=str(tmp_path / "checkpoint"))
    cfg["dataset"].update(batch_size=64, synthetic={"name": "Taobao", "split": "s", "n_domain": 3, "n_user": 300,
                                                    "n_item": 200, "n_train": 900, "n_val": 300, "n_test": 300,
                                                    "pretrained": True})
    return cfg


def small_gen(emb_dim=8):
    return synthetic.generate({"name": "Taobao", "split": "s", "n_domain": 3, "n_user": 300, "n_item": 200,
                               "n_train": 900, "n_val": 300, "n_test": 300, "pretrained": True},
                              batch_size=64, seed=5, emb_dim=emb_dim)


def patch_emb_dim(monkeypatch):
    """tiny tables for CPU speed: synthetic.generate(emb_dim=8) through the dataset layer."""
    real = synthetic.generate
    monkeypatch.setattr(synthetic, "generate", lambda *a, **k: real(*a, **dict(k, emb_dim=8)))


# ------------------------------------------------------------------ registry (run.py:37-85)
def test_registry_dispatch_and_errors(tmp_path, monkeypatch):
    patch_emb_dim(monkeypatch)
    from mamdr_amd.model_zoo import MAML, MAMDR, MLDG, DeepCTR, DomainNegotiation, Reptile
    ds = mds.MultiDomainDataset(tiny_config(tmp_path)["dataset"])
    for name, cls in (("mlp", DeepCTR), ("mlp_meta_mamdr_finetune", MAMDR), ("mlp_meta_reptile", Reptile),
                      ("mlp_meta_mldg", MLDG),
                      ("mlp_meta_domain_negotiation_finetune", DomainNegotiation), ("mlp_meta", MAML)):
        cfg = tiny_config(tmp_path, name)
        assert type(cli.build_model(cfg, ds, FakeEngine)) is cls
    # substring order: 'domain_negotiation' wins over 'mamdr' (run.py:55-58)
    assert type(cli.build_model(tiny_config(tmp_path, "mlp_meta_domain_negotiation_mamdr"), ds, FakeEngine)) \
        is DomainNegotiation
    from mamdr_amd.model_zoo import PCGrad, UncertaintyWeight
    assert type(cli.build_model(tiny_config(tmp_path, "mlp_uncertainty_weight"), ds, FakeEngine)) is UncertaintyWeight
    assert type(cli.build_model(tiny_config(tmp_path, "mlp_pcgrad"), ds, FakeEngine)) is PCGrad
    assert type(cli.build_model(tiny_config(tmp_path, "wdl"), ds, FakeEngine)) is DeepCTR
    for bad in ("autoint", "ccpm"):             # the CPU stand-in has no such towers (the HIP engine does)
        with pytest.raises(NotImplementedError):
            cli.build_model(tiny_config(tmp_path, bad), ds, FakeEngine)
    for name in ("nfm", "pnn"):                 # deepctr.py:33-35,44-46: the generic-layer engine (factory.graph)
        m = cli.build_model(tiny_config(tmp_path, name), ds, FakeEngine)
        assert type(m) is DeepCTR and m.model.kind == name
    # 'star' with the deepctr configs' own `norm: none` / `dense: dense` keys = Star's plain-DNN form (star.py:74-87) on the mlp
    # engine; the PartitionedNorm + StarFCN form needs the Star tower, which this CPU stand-in does not have
    from mamdr_amd.model_zoo import MAMDR as _M
    assert type(cli.build_model(tiny_config(tmp_path, "star_meta_mamdr"), ds, FakeEngine)) is _M
    star_cfg = tiny_config(tmp_path, "star_meta_mamdr")
    star_cfg["model"].update(norm="pn", dense="star", auxiliary_net=False)
    with pytest.raises(NotImplementedError):
        cli.build_model(star_cfg, ds, FakeEngine)
    with pytest.raises(ValueError):
        cli.build_model(tiny_config(tmp_path, "nonsense"), ds, FakeEngine)


# ------------------------------------------------------------------ meta loops vs oracle loops
def test_meta_epochs_follow_oracle_loops():
    g = small_gen()
    sizes = {d: g["data"]["train"][d]["uid"].shape[0] for d in range(3)}

    def fresh():
        eng = FakeEngine(g["n_user"], g["n_item"], 3, 64, emb_dim=8, hidden=(16, 8, 4))
        eng.bind_table("user_emb", g["tables"]["user_emb"])
        eng.bind_table("item_emb", g["tables"]["item_emb"])
        for d in range(3):
            c = g["data"]["train"][d]
            eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
        return eng

    def perm_fn_factory():
        return mplan.PassShuffler(sizes, 10000, 9, shuffle_fn=orng.shuffle_perm)

    plan = {"seq": [1, 2, 0], "dr": [(1, [0, 1]), (2, [1, 2]), (0, [2, 0])]}
    rs = np.random.RandomState(1)
    # MAMDR
    e1, e2 = fresh(), fresh()
    theta0 = e1.oracle.get_flat().copy()
    phis0 = [(rs.standard_normal(theta0.size) * 0.01).astype(np.float32) for _ in range(3)]
    th_o, ph_o = theta0.copy(), [p.copy() for p in phis0]
    tr_o = oloops.mamdr_epoch(e1.oracle, th_o, ph_o, g["data"]["train"], plan, perm_fn_factory(), 64, 0.1)
    th_g, ph_g = torch.from_numpy(theta0.copy()), [torch.from_numpy(p.copy()) for p in phis0]
    tr_g = meta.mamdr_epoch(e2, th_g, ph_g, plan, perm_fn_factory(), 64, lr=1e-3, meta_lr=0.1)
    assert tr_g == tr_o
    assert np.array_equal(th_g.numpy(), th_o)
    for a, b in zip(ph_g, ph_o):
        assert np.array_equal(a.numpy(), b)
    # steps accounting
    spd = [-(-sizes[d] // 64) for d in range(3)]
    assert sum(t[2] for t in tr_g) == mplan.plan_steps(plan, spd)
    # train.finetune_every_epoch (mamdr.py:110-143): one more pass of the merged model per query, phi := theta~ - merged
    e1, e2 = fresh(), fresh()
    th_o, ph_o = theta0.copy(), [p.copy() for p in phis0]
    tr_o = oloops.mamdr_epoch(e1.oracle, th_o, ph_o, g["data"]["train"], plan, perm_fn_factory(), 64, 0.1,
                              finetune_every_epoch=True)
    th_g, ph_g = torch.from_numpy(theta0.copy()), [torch.from_numpy(p.copy()) for p in phis0]
    tr_g = meta.mamdr_epoch(e2, th_g, ph_g, plan, perm_fn_factory(), 64, lr=1e-3, meta_lr=0.1, finetune_every_epoch=True)
    assert tr_g == tr_o and [t for t in tr_g if t[0] == "dr_finetune"] == [("dr_finetune", q, spd[q]) for q, _ in plan["dr"]]
    assert np.array_equal(th_g.numpy(), th_o)
    for a, b, p0 in zip(ph_g, ph_o, phis0):
        assert np.array_equal(a.numpy(), b) and np.abs(b - p0).max() > 1e-3
    # MAML (per-domain and batch outer steps)
    for bv in (False, True):
        e1, e2 = fresh(), fresh()
        th_o = theta0.copy()
        acc_o = np.zeros_like(th_o)
        tr_o = oloops.maml_epoch(e1.oracle, th_o, otower.OuterAdam(th_o.size), acc_o, g["data"]["train"], [1, 0, 2],
                                 perm_fn_factory(), 64, 0.1, batch_variant=bv)
        th_g = torch.from_numpy(theta0.copy())
        acc_g = torch.zeros(th_g.numel())
        e2.bind_accumulator(acc_g)
        tr_g = meta.maml_epoch(e2, th_g, meta.OuterAdamState(e2), acc_g, [1, 0, 2], perm_fn_factory(), 64, 1e-3, 0.1,
                               batch_variant=bv)
        assert tr_g == tr_o and [t[0] for t in tr_g[:2]] == ["maml_train", "maml_meta"]
        assert np.array_equal(th_g.numpy(), th_o) and not acc_g.numpy().any()
        assert np.abs(th_o - theta0).max() > 1e-3                # the outer Adam moved theta by ~meta_lr
    # MLDG (mldg.py:62-125) and MAML over the exclusive meta-train / meta-val split (take / skip windows)
    windows = {d: ((0, int(sizes[d] * 0.8)), (int(sizes[d] * 0.8), sizes[d])) for d in range(3)}
    for fn_g, fn_o, first in ((meta.mldg_epoch, oloops.mldg_epoch, "mldg_train"),
                              (meta.maml_epoch, oloops.maml_epoch, "maml_train")):
        for bv in (False, True):
            e1, e2 = fresh(), fresh()
            th_o = theta0.copy()
            acc_o = np.zeros_like(th_o)
            tr_o = fn_o(e1.oracle, th_o, otower.OuterAdam(th_o.size), acc_o, g["data"]["train"], [1, 0, 2],
                        perm_fn_factory(), 64, 0.1, batch_variant=bv, windows=windows)
            th_g = torch.from_numpy(theta0.copy())
            acc_g = torch.zeros(th_g.numel())
            e2.bind_accumulator(acc_g)
            tr_g = fn_g(e2, th_g, meta.OuterAdamState(e2), acc_g, [1, 0, 2], perm_fn_factory(), 64, 1e-3, 0.1,
                        batch_variant=bv, windows=windows)
            assert tr_g == tr_o and tr_g[0][0] == first
            # the two passes of a domain cover the 80 % / 20 % slices, final partial batches kept
            assert tr_g[0][2] == -(-windows[1][0][1] // 64) and tr_g[1][2] == -(-(sizes[1] - windows[1][0][1]) // 64)
            assert np.array_equal(th_g.numpy(), th_o) and not acc_g.numpy().any()
            assert np.abs(th_o - theta0).max() > 1e-3
        # train.target_domain: every meta pass runs over the target domain's whole train split (maml.py:336-338)
        e1, e2 = fresh(), fresh()
        th_o = theta0.copy()
        acc_o = np.zeros_like(th_o)
        tr_o = fn_o(e1.oracle, th_o, otower.OuterAdam(th_o.size), acc_o, g["data"]["train"], [1, 0],
                    perm_fn_factory(), 64, 0.1, windows=windows, meta_domain=2)
        th_g = torch.from_numpy(theta0.copy())
        acc_g = torch.zeros(th_g.numel())
        e2.bind_accumulator(acc_g)
        tr_g = fn_g(e2, th_g, meta.OuterAdamState(e2), acc_g, [1, 0], perm_fn_factory(), 64, 1e-3, 0.1,
                    windows=windows, meta_domain=2)
        assert tr_g == tr_o and [t[1:] for t in tr_g if t[0].endswith("_meta")] == [(2, -(-sizes[2] // 64))] * 2
        assert np.array_equal(th_g.numpy(), th_o)
    # PCGrad (pcgrad.py:62-124): query gradient + projected auxiliary gradients, outer Adam on the live model
    e1, e2 = fresh(), fresh()
    aux_plan = {1: [0, 2], 0: [2], 2: [1, 0]}
    tr_o = oloops.pcgrad_epoch(e1.oracle, otower.OuterAdam(theta0.size), g["data"]["train"], [1, 0, 2], aux_plan,
                               perm_fn_factory(), 64, 0.01)
    cur_g, aux_g = torch.zeros(theta0.size), torch.zeros(theta0.size)
    tr_g = meta.pcgrad_epoch(e2, meta.OuterAdamState(e2), cur_g, aux_g, [1, 0, 2], aux_plan, perm_fn_factory(), 64,
                             1e-3, 0.01)
    assert tr_g == tr_o and [t[0] for t in tr_g[:3]] == ["pcgrad_query", "pcgrad_aux", "pcgrad_aux"]
    assert np.array_equal(e2.oracle.get_flat(), e1.oracle.get_flat())
    assert np.abs(e1.oracle.get_flat() - theta0).max() > 1e-3
    # DN and Reptile (both variants)
    for fn_g, fn_o, kw in ((meta.dn_epoch, oloops.dn_epoch, {}), (meta.reptile_epoch, oloops.reptile_epoch, {}),
                           (meta.reptile_epoch, oloops.reptile_epoch, {"batch_variant": True})):
        e1, e2 = fresh(), fresh()
        th_o = theta0.copy()
        tr_o = fn_o(e1.oracle, th_o, g["data"]["train"], [2, 0, 1], perm_fn_factory(), 64, 0.1, **kw)
        th_g = torch.from_numpy(theta0.copy())
        tr_g = fn_g(e2, th_g, [2, 0, 1], perm_fn_factory(), 64, 1e-3, 0.1, **kw)
        assert tr_g == tr_o and np.array_equal(th_g.numpy(), th_o)
        assert np.array_equal(e2.oracle.get_flat(), th_o)       # model left at theta
        # train.target_domain (domain_negotiation.py:44-45,67,89-93; reptile.py:47-48,82-85,98-102)
        e1, e2 = fresh(), fresh()
        th_o = theta0.copy()
        seq = [2, 0] if fn_g is meta.dn_epoch else [2, 0, 1]     # DN's meta sequence excludes the target, Reptile skips it
        tr_o = fn_o(e1.oracle, th_o, g["data"]["train"], seq, perm_fn_factory(), 64, 0.1, meta_train_step=2, target=1, **kw)
        th_g = torch.from_numpy(theta0.copy())
        tr_g = fn_g(e2, th_g, seq, perm_fn_factory(), 64, 1e-3, 0.1, meta_train_step=2, target=1, **kw)
        assert tr_g == tr_o and np.array_equal(th_g.numpy(), th_o)
        assert tr_g[-1] == ("target", 1, spd[1])                # the epoch ends with a full pass over the target
        assert np.array_equal(e2.oracle.get_flat(), e1.oracle.get_flat())
        assert np.abs(e2.oracle.get_flat() - th_o).max() > 1e-4  # ... of the model, not of theta
        if fn_g is meta.dn_epoch:
            assert tr_g[:3] == [("dn", 2, 2), ("dn", 0, 2), ("dn", 1, spd[1])]   # the target's pass is not capped
        else:
            assert tr_g[:4] == [("reptile", 2, 2), ("target_step", 1, 1), ("reptile", 0, 2), ("target_step", 1, 1)]


def test_moving_average_update_known_answers():
    """TF 1.12's zero-debiased moving average behind K.moving_average_update (oracle/outer.py): a constant gradient
    is reproduced from the first update on (that is what the debiasing is for), the hidden `biased` follows
    b_t = g (1 - 0.999^t), and a cleared accumulator is overwritten, not averaged with."""
    from oracle import outer as oouter
    g = np.array([1.0, -2.5, 0.0, 3e-3], np.float32)
    u, b, step = np.zeros(4, np.float32), np.zeros(4, np.float32), 0
    for t in range(1, 6):
        step = oouter.moving_average_update(u, b, g, 0.999, step)
        assert step == t
        np.testing.assert_allclose(u, g, rtol=1e-3, atol=1e-9)       # fp32 cancellation in 1 - 0.999^t: ~1e-4 relative
        np.testing.assert_allclose(b, g * (1.0 - 0.999 ** t), rtol=1e-3, atol=1e-9)
        u[...] = 0                                                   # clear_grads (maml.py:203)
    # first update in exact arithmetic: b = 0.001 g, u = b / 0.001 -- in float32, bit for bit
    u, b = np.zeros(1, np.float32), np.zeros(1, np.float32)
    oouter.moving_average_update(u, b, np.array([0.7], np.float32), 0.999, 0)
    d = np.float32(1.0 - 0.999)
    b1 = np.float32(0) - (np.float32(0) - np.float32(0.7)) * d
    den = np.float32(1) - np.power(np.float32(1) - d, np.float32(1), dtype=np.float32)
    assert b[0] == b1 and u[0] == np.float32(0) - (np.float32(0) - b1 / den)
    # two different gradients: the debiased value is their weighted mean (weights 0.999 : 1)
    u, b, step = np.zeros(1, np.float32), np.zeros(1, np.float32), 0
    step = oouter.moving_average_update(u, b, np.array([1.0], np.float32), 0.999, step)
    step = oouter.moving_average_update(u, b, np.array([3.0], np.float32), 0.999, step)
    np.testing.assert_allclose(u[0], (0.999 * 1.0 + 3.0) / 1.999, rtol=1e-3)


@pytest.mark.parametrize("avg", ["moving_mean", "drop"])
def test_run_main_average_meta_grad(tmp_path, monkeypatch, avg):
    """train.average_meta_grad "moving_mean" / "drop" (maml.py:218-229) through run.py's entry: "drop" is the plain sum
    (Dropout in a K.function that never feeds the learning phase), "moving_mean" keeps the debiased moving average
    with its hidden state alive across the cleared accumulator."""
    patch_emb_dim(monkeypatch)
    results = {}
    for mode in ("none", avg):
        cfg = tiny_config(tmp_path, "mlp_meta_maml")
        cfg["train"]["average_meta_grad"] = mode
        holder = {}
        real_build = cli.build_model
        monkeypatch.setattr(cli, "build_model", lambda *a, real_build=real_build, holder=holder, **k:
                            holder.setdefault("m", real_build(*a, **k)))
        out = cli.main(cfg, FakeEngine)
        monkeypatch.setattr(cli, "build_model", real_build)
        results[mode] = (out, holder["m"])
    (o_none, m_none), (o_avg, m_avg) = results["none"], results[avg]
    assert m_avg.trace == m_none.trace and np.isfinite(o_avg[0])
    if avg == "drop":
        assert o_avg == o_none and m_avg.model._ema is None
    else:
        ema = m_avg.model._ema
        n_meta_batches = sum(n for p, _, n in m_avg.trace if p == "maml_meta")
        assert ema["step"] == n_meta_batches > 0 and np.abs(ema["biased"]).max() > 0
        assert o_avg != o_none                           # a different outer gradient


@pytest.mark.parametrize("name, extra", [("mlp_meta_domain_negotiation", {"target_domain": 1, "meta_train_step": 2}),
                                         ("mlp_meta_reptile", {"target_domain": 1}),
                                         ("mlp_meta_mamdr", {"target_domain": 2, "finetune_every_epoch": True}),
                                         ("mlp_meta_maml", {"target_domain": 0}), ("mlp_meta_mldg", {"target_domain": 0}),
                                         ("mlp_pcgrad", {"target_domain": 0}),
                                         ("mlp_meta_maml", {"target_domain": -1, "meta_finetune_step": 2})])
def test_run_main_target_domain_and_finetune_every_epoch(tmp_path, monkeypatch, name, extra):
    """train.target_domain (domain_negotiation.py:44-45,89-104; reptile.py:47-48,82-102; mamdr.py:153-154) and
    train.finetune_every_epoch (mamdr.py:110-143) through run.py's entry: the target domain closes DN's inner
    sequence and every epoch, Reptile steps on it after every domain, early stopping watches its val AUC."""
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name)
    cfg["train"].update(extra)
    seen = {}
    from mamdr_amd.model_zoo import base_model as bm, specific_base_model as sbm
    for cls in (bm.BaseModel, sbm.SpecificBase):
        if "early_stop_step" in cls.__dict__:
            real = cls.early_stop_step
            monkeypatch.setattr(cls, "early_stop_step",
                                lambda self, m, real=real: (seen.setdefault("metric", []).append(m), real(self, m))[1])
    model_holder = {}
    real_build = cli.build_model
    monkeypatch.setattr(cli, "build_model", lambda *a, **k: model_holder.setdefault("m", real_build(*a, **k)))
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg, FakeEngine)
    assert set(domain_auc) == {0, 1, 2} and np.isfinite(avg_loss)
    trace = model_holder["m"].trace
    t = extra["target_domain"]
    phases = [p for p, _, _ in trace]
    if "domain_negotiation" in name:
        assert all(d != t for p, d, _ in trace if p == "dn" and _ == 2)      # capped passes: never the target
        assert phases.count("target") >= 1 and all(d == t for p, d, _ in trace if p == "target")
    elif "reptile" in name:
        assert all(d != t for p, d, _ in trace if p == "reptile")
        assert phases.count("target_step") == 2 * phases.count("target") and phases.count("target") >= 1
    elif extra.get("meta_finetune_step", 0) > 0:
        # maml.py:245-287: every validation fine-tunes each domain for meta_finetune_step passes from the same weights
        n_val = len(seen["metric"])
        assert phases.count("meta_finetune") == n_val * 3 * 2
    elif "maml" in name or "mldg" in name:
        assert all(d != t for p, d, _ in trace if p.endswith("_train")) and phases.count("target") >= 1
        assert all(d == t for p, d, _ in trace if p == "target" or p.endswith("_meta"))   # maml.py:336-338
    elif "pcgrad" in name:
        assert all(d != t for p, d, _ in trace if p == "pcgrad_query") and "target" not in phases
    else:
        # the meta sequence skips the target domain (maml.py:305-308), so two query domains per epoch
        assert phases.count("dr_finetune") % 2 == 0 and phases.count("dr_finetune") >= 2
        assert all(d != t for p, d, _ in trace if p in ("dn", "dr_query", "dr_finetune"))
    assert len(seen["metric"]) >= 1 and all(0.0 <= m <= 1.0 for m in seen["metric"])


def test_non_exclusive_meta_split(tmp_path, monkeypatch):
    """meta_split "meta-train/val-no-exclusive" (maml.py:316-323): shuffle the whole split, then take / skip --
    the meta-train pass sees the first int(n * ratio) positions of one shuffle, the meta-val pass the remaining
    positions of ANOTHER shuffle, so the two may share rows (the exclusive variant never does)."""
    sizes = {0: 1000}
    sh = mplan.PassShuffler(sizes, 10000, 3)
    a = sh(0, (0, 800, "stream"))
    b = sh(0, (800, 1000, "stream"))
    assert a.shape == (800,) and b.shape == (200,) and a.dtype == np.int32
    assert len(np.unique(a)) == 800 and len(np.unique(b)) == 200 and a.max() < 1000 and b.max() < 1000
    assert len(np.intersect1d(a, b)) > 0 and a.max() >= 800          # not the file-order slices
    sh2 = mplan.PassShuffler(sizes, 10000, 3)
    c, e = sh2(0, (0, 800)), sh2(0, (800, 1000))
    assert c.max() < 800 and e.min() >= 800 and len(np.intersect1d(c, e)) == 0
    assert np.array_equal(mplan.PassShuffler(sizes, 10000, 3, shuffle=False)(0, (800, 1000, "stream")), np.arange(800, 1000))
    # through run.py's entry
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp_meta_maml")
    cfg["train"].update(meta_split="meta-train/val-no-exclusive", meta_split_ratio=0.8)
    holder = {}
    real_build = cli.build_model
    monkeypatch.setattr(cli, "build_model", lambda *a, **k: holder.setdefault("m", real_build(*a, **k)))
    cli.main(cfg, FakeEngine)
    m = holder["m"]
    n = {d: v["n_data"] for d, v in m.dataset.train_dataset.items()}
    for p, d, steps in m.trace:
        if p == "maml_train":
            assert steps == -(-int(n[d] * 0.8) // 64)
        elif p == "maml_meta":
            assert steps == -(-(n[d] - int(n[d] * 0.8)) // 64)


def test_epoch_planner_semantics():
    p = mplan.EpochPlanner(range(6), sample_num=3, add_query_domain=True, seed=1)
    plan = p.next_epoch()
    assert sorted(plan["seq"]) == list(range(6))
    assert [q for q, _ in plan["dr"]] == plan["seq"]            # DR visits queries in the DN order
    for q, support in plan["dr"]:
        assert support[-1] == q and len(support) == 4 and q not in support[:-1]
        assert len(set(support)) == 4
    plan2 = mplan.EpochPlanner(range(6), 3, True, seed=1).next_epoch()
    assert plan2 == plan                                         # seeded -> reproducible
    p3 = mplan.EpochPlanner(range(3), sample_num=5, add_query_domain=False, shuffle_sequence=False)
    plan3 = p3.next_epoch()
    assert plan3["seq"] == [0, 1, 2] and all(len(s) == 2 for _, s in plan3["dr"])


# ------------------------------------------------------------------ run.py end to end (CPU stand-in tower)
@pytest.mark.parametrize("name", ["mlp_meta_mamdr_finetune", "mlp_meta_domain_negotiation_finetune",
                                  "mlp_meta_reptile_batch", "mlp", "mlp_meta_maml_finetune"])
def test_run_main_end_to_end(tmp_path, monkeypatch, name):
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name)
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg, FakeEngine)
    assert set(domain_auc) == {0, 1, 2} and 0.0 <= avg_auc <= 1.0 and np.isfinite(avg_loss)
    assert abs(avg_auc - sum(domain_auc.values()) / 3) < 1e-12
    # result directory layout (base_model.py:183-200)
    rdir = os.path.join(cfg["train"]["result_save_path"], name, "Taobao", cfg["dataset"]["domain_split_path"])
    runs = os.listdir(rdir)
    assert len(runs) == 1 and re.match(r"loss_\d+\.\d{3}_auc_\d+\.\d{3}_", runs[0])
    files = set(os.listdir(os.path.join(rdir, runs[0])))
    assert {"dataset_info.json", "config.json.example", "result.json", "model_parameters.npz"} <= files
    with open(os.path.join(rdir, runs[0], "result.json")) as f:
        res = json.load(f)
    assert set(res) == {"avg_loss", "avg_auc", "domain_loss", "domain_auc"} and set(res["domain_auc"]) == {"0", "1", "2"}
    with open(os.path.join(rdir, runs[0], "dataset_info.json")) as f:
        info = json.load(f)
    assert info["total_train"] == sum(info[str(d)]["n_train"] for d in range(3))


@pytest.mark.parametrize("name,hidden", [("mlp_meta_mamdr_finetune", [16, 8]), ("deepfm_meta_domain_negotiation", [16, 8, 8, 4]),
                                         ("wdl", [8])])
def test_run_main_other_hidden_dims(tmp_path, monkeypatch, name, hidden):
    """deepctr.py:26-49 passes ANY hidden_dim list on as dnn_hidden_units: 1 .. 4 hidden layers run (round 4 raised for
    everything but three), theta / phi / checkpoints carry 2 n + 2 dense tensors in Keras order, five layers raise."""
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name)
    cfg["model"]["hidden_dim"] = hidden
    built = []
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg, FakeEngine, on_model=built.append)
    assert set(domain_auc) == {0, 1, 2} and np.isfinite(avg_loss)
    eng = built[0].model
    n = len(hidden)
    dense = [k for k in eng.segments if k[0] in "Wb" and k[1:].isdigit() or k in ("wo", "gb")]
    assert dense == ["W%d" % l for l in range(n)] + ["b%d" % l for l in range(n)] + ["wo", "gb"]
    dims = [24] + hidden
    assert [eng.segments["W%d" % l][1] for l in range(n)] == [dims[l] * dims[l + 1] for l in range(n)]
    cfg["model"]["hidden_dim"] = [8, 8, 8, 8, 8]
    with pytest.raises(ValueError):
        cli.main(cfg, FakeEngine)


def test_mamdr_wrapper_quirks(tmp_path, monkeypatch):
    """phi_d = fresh random init of the whole model; finetune = SGD lr 0.001 from best merged weights."""
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp_meta_mamdr_finetune", epochs=1)
    ds = mds.MultiDomainDataset(cfg["dataset"])
    model = cli.build_model(cfg, ds, FakeEngine)
    model.train()
    phi0 = model.domain_weights[0].numpy()
    assert np.abs(phi0).max() > 0.05                              # glorot-scale kernels, not zeros
    assert not np.array_equal(phi0, model.domain_weights[1].numpy())
    assert model.best_shared_weights is not None and set(model.best_domain_weights) == {0, 1, 2}
    model.model.calls.clear()
    model.separate_train_val_test(init_parms=False)
    assert model.model.calls and all(c[2] == "sgd" and c[3] == 0.001 for c in model.model.calls)
    # every DR pass pair: support pass then query pass; DN first
    phases = [t[0] for t in model.trace]
    n_dom = 3
    assert phases[:n_dom] == ["dn"] * n_dom
    assert phases[n_dom:] == ["dr_support", "dr_query"] * ((len(phases) - n_dom) // 2)


@pytest.mark.parametrize("name", ["mlp_meta_domain_negotiation_finetune", "mlp_meta_mamdr_finetune"])
def test_finetune_stage_follows_the_oracle(tmp_path, monkeypatch, name):
    """separate_train_val_test(init_parms=False) (base_model.py:41-109; MAMDR: specific_base_model.py:99-162) against
    oracle/loops.finetune_domains on the oracle-backed engine: per domain the same epochs run, the same kept
    checkpoint, the same val AUC sequence and test AUC (exact: same arithmetic on both sides), SGD with
    `learning_rate` / 0.001, every domain restarted from the same weights / from best theta + best phi_d."""
    import copy as _copy
    from oracle import auc as oauc
    from oracle import outer as oouter
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name, epochs=4)
    cfg["train"].update(patience=2, learning_rate=0.05)
    ds = mds.MultiDomainDataset(cfg["dataset"])
    model = cli.build_model(cfg, ds, FakeEngine)
    model.train()
    model.load_model(model.checkpoint_path)
    eng = model.model
    twin = _copy.deepcopy(eng.oracle)                    # the oracle model in the state the finetune stage starts from
    counter0 = model.shuffler.counter
    model.model.calls.clear()
    _, _, d_loss, d_auc = model.separate_train_val_test(init_parms=False)
    log = model.base_model.finetune_log if hasattr(model, "base_model") else model.finetune_log
    sizes = {d: v["n_data"] for d, v in ds.train_dataset.items()}
    shuf = mplan.PassShuffler(sizes, ds.shuffle_buffer_size, ds.seed, shuffle_fn=orng.shuffle_perm)
    shuf.counter = counter0
    data = {sp: {d: st[d]["data"] for d in st} for sp, st in (("train", ds.train_dataset), ("val", ds.val_dataset),
                                                              ("test", ds.test_dataset))}
    if "mamdr" in name:
        start = lambda d: oouter.merge(model.best_shared_weights.numpy(), model.best_domain_weights[d].numpy(), "plus")
        lr = 0.001
    else:
        w0 = twin.get_flat().copy()
        start = lambda d: w0
        lr = 0.05
    want, trace = oloops.finetune_domains(twin, data, start, shuf, 64, 4, 2, lr, oauc.auc500)
    assert sorted(want) == sorted(log) == [0, 1, 2]
    for d in want:
        assert log[d]["epochs"] == want[d]["epochs"] and log[d]["best_epoch"] == want[d]["best_epoch"], (d, log[d], want[d])
        assert log[d]["val_auc"] == want[d]["val_auc"]
        assert d_auc[d] == want[d]["test_auc"] and abs(d_loss[d] - want[d]["test_loss"]) < 1e-6
    assert all(c[2] == "sgd" and c[3] == lr for c in eng.calls)
    assert [t[2] for t in trace] == [c[1] for c in eng.calls]


def test_tables_without_pretraining_always_train(tmp_path, monkeypatch):
    """deepctr.py:104-116: `trainable=emb_trainable` is passed to SparseFeat only on the pretrained branch; with
    load_pretrain_emb false the tables take deepctr's default and train even when emb_trainable is false."""
    patch_emb_dim(monkeypatch)
    for pre, flag, want in ((True, False, False), (True, True, True), (False, False, True), (False, True, True)):
        cfg = tiny_config(tmp_path, "mlp")
        cfg["train"].update(load_pretrain_emb=pre, emb_trainable=flag)
        ds = mds.MultiDomainDataset(cfg["dataset"])
        m = cli.build_model(cfg, ds, FakeEngine)
        assert m.model.oracle.emb_trainable is want, (pre, flag)
        assert ("user_emb" in m.model.segments) is want


def test_early_stop_counts_ties(tmp_path, monkeypatch):
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp")
    cfg["train"]["patience"] = 2
    ds = mds.MultiDomainDataset(cfg["dataset"])
    m = cli.build_model(cfg, ds, FakeEngine)
    assert m.early_stop_step(0.7) is False
    assert m.early_stop_step(0.7) is False and m.counter == 1     # equal counts as no improvement
    assert m.early_stop_step(0.8) is False and m.counter == 0
    assert m.early_stop_step(0.75) is False
    assert m.early_stop_step(0.8) is True


# ------------------------------------------------------------------ dataset formats (utils/dataset.py)
def test_reference_layout_round_trip(tmp_path):
    g = small_gen()
    root = str(tmp_path / "dataset" / "Taobao")
    mds.write_reference_layout(g, root, "split_by_theme_3")
    conf = {"name": "Taobao", "dataset_path": root, "domain_split_path": "split_by_theme_3", "batch_size": 64,
            "shuffle_buffer_size": 10000, "num_parallel_reads": 8, "seed": 5}
    for attempt in range(2):                                     # second pass reads the .npz cache
        ds = mds.MultiDomainDataset(conf)
        assert (ds.n_uid, ds.n_pid, ds.n_domain) == (g["n_user"], g["n_item"], 3)
        for d in range(3):
            for split, store in (("train", ds.train_dataset), ("val", ds.val_dataset), ("test", ds.test_dataset)):
                for c in mds.COLUMNS:
                    assert np.array_equal(store[d]["data"][c], g["data"][split][d][c])
                n = g["data"][split][d]["uid"].shape[0]
                assert store[d]["n_data"] == n and store[d]["n_step"] == -(-n // 64)
            assert ds.ctr_ratio[d] == g["info"][d]["ctr_ratio"]
        assert np.array_equal(ds.user_emb, g["tables"]["user_emb"])
        assert np.array_equal(ds.item_emb, g["tables"]["item_emb"])
    info = ds.dataset_info
    assert info["n_user"] == g["n_user"] and info["total_val"] == sum(info[d]["n_val"] for d in range(3))


def test_dataset_cache_is_invalidated_by_source_changes(tmp_path):
    """the .npz cache carries (size, mtime) of every source file: an edited CSV is re-read, a torn cache file is
    ignored, and the cache name does not depend on the batch size."""
    g = small_gen()
    root = str(tmp_path / "dataset" / "Taobao")
    mds.write_reference_layout(g, root, "split_by_theme_3")
    conf = {"name": "Taobao", "dataset_path": root, "domain_split_path": "split_by_theme_3", "batch_size": 64,
            "shuffle_buffer_size": 10000, "num_parallel_reads": 8, "seed": 5}
    mds.MultiDomainDataset(conf)
    base = os.path.join(root, "split_by_theme_3")
    caches = [f for f in os.listdir(base) if f.endswith(".npz")]
    assert caches == ["mamdr_amd_cache.npz"]
    csv = os.path.join(base, "domain_1", "val.csv")
    with open(csv) as f:
        lines = f.read().splitlines()
    with open(csv, "w") as f:
        f.write("\n".join(lines[:-3]) + "\n")                    # three rows fewer
    ds = mds.MultiDomainDataset(dict(conf, batch_size=32))
    assert ds.val_dataset[1]["n_data"] == g["data"]["val"][1]["uid"].shape[0] - 3
    with open(os.path.join(base, caches[0]), "wb") as f:           # a torn cache file
        f.write(b"PK\x03\x04 not a zip")
    ds = mds.MultiDomainDataset(conf)
    assert ds.val_dataset[1]["n_data"] == g["data"]["val"][1]["uid"].shape[0] - 3
    assert np.array_equal(ds.train_dataset[0]["data"]["uid"], g["data"]["train"][0]["uid"])


def test_missing_dataset_is_a_clear_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        mds.MultiDomainDataset({"name": "Taobao", "dataset_path": str(tmp_path), "domain_split_path": "nope",
                                "batch_size": 64, "shuffle_buffer_size": 10, "num_parallel_reads": 1, "seed": 1})


def test_synthetic_shapes_match_table_one():
    g = synthetic.generate("taobao10", batch_size=1024)
    assert g["n_domain"] == 10 and g["n_user"] == 23778 and g["n_item"] == 6932
    assert g["info"]["total_train"] == 92137 and g["info"]["total_val"] == 37645 and g["info"]["total_test"] == 43502
    sizes = [g["info"][d]["n_train"] for d in range(10)]
    assert min(sizes) >= 1024 and max(sizes) > 5 * min(sizes)     # long tail, every domain >= one batch
    for d in range(10):
        c = g["data"]["train"][d]
        assert c["uid"].dtype == np.int32 and c["label"].dtype == np.float32
        assert set(np.unique(c["label"])) <= {0.0, 1.0} and np.all(c["domain"] == d)
        r = g["info"][d]["ctr_ratio"]
        assert 0.2 <= r <= 0.5


# ------------------------------------------------------------------ multi-task baselines (deep_mtl_ctr.py)
@pytest.mark.parametrize("name,extra", [
    ("shared_bottom", {}),
    ("mmoe", {"num_experts": 2, "gate_dnn_hidden_units": [4]}),
    ("ple", {"shared_expert_num": 1, "specific_expert_num": 2, "gate_dnn_hidden_units": [4], "num_levels": 1}),
])
def test_mtl_towers_run_entry(tmp_path, monkeypatch, name, extra):
    """run.py's entry for the multi-task names (run.py:44-45 -> DeepMTLCTR): alternate training through the
    per-domain models (deep_mtl_ctr.py:69-96), per-domain evaluation, early stopping, result files; the engine sees
    the domain with every pass and trains that task's variables only."""
    from fake_engine import FakeGraphEngine
    from mamdr_amd.model_zoo import DeepMTLCTR
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name, epochs=2)
    cfg["model"].update(hidden_dim=[16, 8], tower_hidden_dim=[4], **extra)
    ds = mds.MultiDomainDataset(cfg["dataset"])
    model = cli.build_model(cfg, ds, FakeGraphEngine)
    assert type(model) is DeepMTLCTR and model.model.kind == name
    # the initial tensors cover the engine's flat vector exactly, in its order
    assert [n for n, _ in model.plan] == list(model.model.segments)
    before = model.model.get_weights().numpy().copy()
    avg_loss, avg_auc, dl, da = cli.main(cfg, FakeGraphEngine)
    assert sorted(da) == [0, 1, 2] and np.isfinite(avg_loss) and 0.3 < avg_auc < 1.0
    res = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path / "result")) for f in fs if f == "result.json"]
    assert len(res) == 1 and json.load(open(res[0]))["avg_auc"] == avg_auc
    # every pass named its domain; each epoch covers every domain once
    model.train()
    doms = [c[0] for c in model.model.calls]
    assert sorted(doms[:3]) == [0, 1, 2]
    after = model.model.get_weights().numpy()
    assert not np.array_equal(before, after)


def test_mtl_ple_levels_and_trainable_tables_are_named(tmp_path, monkeypatch):
    from fake_engine import FakeGraphEngine
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "ple", epochs=1)
    cfg["model"].update(hidden_dim=[16], tower_hidden_dim=[4], shared_expert_num=1, specific_expert_num=1,
                        gate_dnn_hidden_units=[4], num_levels=2)
    ds = mds.MultiDomainDataset(cfg["dataset"])
    with pytest.raises(NotImplementedError, match="num_levels"):
        cli.build_model(cfg, ds, FakeGraphEngine)


@pytest.mark.parametrize("name", ["nfm", "pnn", "nfm_meta_domain_negotiation", "pnn_meta_maml", "nfm_meta_mldg", "pnn_pcgrad",
                                  "nfm_meta_mamdr_finetune"])
def test_nfm_pnn_run_entry(tmp_path, monkeypatch, name):
    """deepctr NFM / PNN through run.py's entry (plain alternate training; the meta wrappers on top: the outer updates
    are the same flat-vector operations whatever the tower, the accumulate passes of MAML / MLDG / PCGrad run the tower's
    step in learning phase 0)."""
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name, epochs=2)
    if "maml" in name or "mldg" in name or "pcgrad" in name:
        cfg["train"]["meta_learning_rate"] = 0.003
    if "mldg" in name:
        cfg["train"].update(meta_split="meta-train/val", meta_split_ratio=0.8)
    avg_loss, avg_auc, dl, da = cli.main(cfg, FakeEngine)
    assert sorted(da) == [0, 1, 2] and np.isfinite(avg_loss) and 0.3 < avg_auc < 1.0


# ------------------------------------------------------------------ meta parameter filters (maml.py:153-179)
def test_all_hidden_meta_range_follows_the_reference_semantics():
    """meta_parms ["all_hidden"] (maml.py:160-166): theta covers every variable whose name lacks "emb" -- here one contiguous
    range behind the domain table.  A Domain Negotiation epoch then resets / interpolates the hidden weights only; the
    domain table is trained by the inner steps and never reset.  Checked against the same loop written out on the oracle."""
    g = small_gen()
    sizes = {d: g["data"]["train"][d]["uid"].shape[0] for d in range(3)}

    def fresh():
        eng = FakeEngine(g["n_user"], g["n_item"], 3, 64, emb_dim=8, hidden=(16, 8, 4))
        eng.bind_table("user_emb", g["tables"]["user_emb"])
        eng.bind_table("item_emb", g["tables"]["item_emb"])
        for d in range(3):
            c = g["data"]["train"][d]
            eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
        return eng
    eng, ref = fresh(), fresh()
    off = eng.segments["W0"][0]
    assert off == 3 * 8                                        # the domain table sits in front
    eng.set_meta_range(off, eng.n_params - off)
    theta = eng.meta_weights.clone()
    assert theta.numel() == eng.n_params - 24
    sh = [mplan.PassShuffler(sizes, 10000, 9, shuffle_fn=orng.shuffle_perm) for _ in range(2)]
    seq = [2, 0, 1]
    meta.dn_epoch(eng, theta, seq, sh[0], 64, 1e-3, 0.5)
    # the same on the oracle-backed twin, by hand: model[hidden] := theta; passes; theta += (model[hidden] - theta) * 0.5
    full = ref.oracle.get_flat()
    th = full[off:].copy()
    for d in seq:
        meta.run_pass(ref, d, sh[1], 64, 1e-3, [], "dn")
    after = ref.oracle.get_flat()
    th = (th + (after[off:] - th) * np.float32(0.5)).astype(np.float32)
    np.testing.assert_array_equal(theta.numpy(), th)
    live = eng.oracle.get_flat()
    np.testing.assert_array_equal(live[off:], th)              # hidden weights := theta
    np.testing.assert_array_equal(live[:off], after[:off])     # the domain table keeps what the inner steps made of it
    assert not np.array_equal(after[:off], full[:off])


@pytest.mark.parametrize("name", ["mlp_meta_domain_negotiation", "mlp_meta", "mlp_meta_mamdr", "mlp_pcgrad", "mlp_meta_mldg"])
def test_all_hidden_through_the_run_entry(tmp_path, monkeypatch, name):
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, name, epochs=2)
    cfg["train"]["meta_parms"] = ["all_hidden"]
    avg_loss, avg_auc, dl, da = cli.main(cfg, FakeEngine)
    assert sorted(da) == [0, 1, 2] and np.isfinite(avg_loss) and 0.3 < avg_auc < 1.0


@pytest.mark.parametrize("loop", ["dn", "reptile", "mamdr"])
def test_scattered_meta_parms_follow_the_reference_semantics(loop):
    """a `meta_parms` list whose tensors are no neighbours in the flat vector (maml.py:167-177 takes any list): theta / phi
    span the range from the first selected tensor to the last, the tensors in between are HOLES that `assign_meta` never
    writes -- they train on undisturbed, as variables outside `model_meta_parms` do in the reference.  Against the oracle's
    loops on a model whose flat vector is the chosen tensors alone: identical bits in every tensor."""
    from oracle import loops as oloops
    g = small_gen()
    sizes = {d: g["data"]["train"][d]["uid"].shape[0] for d in range(3)}
    eng = FakeEngine(g["n_user"], g["n_item"], 3, 64, emb_dim=8, hidden=(16, 8, 4))
    eng.bind_table("user_emb", g["tables"]["user_emb"])
    eng.bind_table("item_emb", g["tables"]["item_emb"])
    for d in range(3):
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    chosen = ["domain_emb", "W1", "b1", "gb"]
    segs = eng.segments
    order = sorted(segs, key=lambda n: segs[n][0])
    lo, hi = segs[chosen[0]][0], segs[chosen[-1]][0] + segs[chosen[-1]][1]
    holes, run = [], None
    for n in order:             # maximal runs of unselected tensors inside [lo, hi)
        o, c = segs[n]
        if n in chosen or o < lo or o >= hi:
            if run:
                holes.append((run[0] - lo, run[1] - run[0]))
            run = None
        else:
            run = (run[0] if run else o, o + c)
    assert len(holes) == 3
    eng.set_meta_range(lo, hi - lo, holes)
    ref = otower.OracleModel({k: v.copy() for k, v in eng.oracle.params.items()}, emb_trainable=False, dropout=eng.oracle.rate,
                             lr=1e-3, hidden=(16, 8, 4), dropout_seed=eng.oracle.seed)
    sub = MetaSubset(ref, chosen)
    data = {d: g["data"]["train"][d] for d in range(3)}
    sh = [mplan.PassShuffler(sizes, 10000, 9, shuffle_fn=orng.shuffle_perm) for _ in range(2)]
    theta = eng.meta_weights.clone()
    theta_o = sub.get_flat()
    seq = [2, 0, 1]
    if loop == "dn":
        tr = meta.dn_epoch(eng, theta, seq, sh[0], 64, 1e-3, 0.5)
        tr_o = oloops.dn_epoch(sub, theta_o, data, seq, sh[1], 64, 0.5)
    elif loop == "reptile":
        tr = meta.reptile_epoch(eng, theta, seq, sh[0], 64, 1e-3, 0.5)
        tr_o = oloops.reptile_epoch(sub, theta_o, data, seq, sh[1], 64, 0.5)
    else:
        plan = {"seq": seq, "dr": [(0, [1, 2]), (2, [0, 1]), (1, [2, 0])]}
        rs = np.random.RandomState(3)
        phis_o = [(rs.standard_normal(theta_o.size) * 1e-3).astype(np.float32) for _ in range(3)]
        phis = []
        for p in phis_o:
            t = torch.zeros_like(theta)
            o = 0
            for n in chosen:
                off, cnt = segs[n]
                t[off - lo:off - lo + cnt] = torch.from_numpy(p[o:o + cnt])
                o += cnt
            phis.append(t)
        tr = meta.mamdr_epoch(eng, theta, phis, plan, sh[0], 64, 1e-3, 0.5)
        tr_o = oloops.mamdr_epoch(sub, theta_o, phis_o, data, plan, sh[1], 64, 0.5)
    assert tr == tr_o
    live, want = eng.oracle.params, ref.params
    for n in order:
        np.testing.assert_array_equal(live[n], want[n], err_msg=n)
    o = 0
    for n in chosen:            # theta of the chosen tensors, bit for bit
        off, cnt = segs[n]
        np.testing.assert_array_equal(theta.numpy()[off - lo:off - lo + cnt], theta_o[o:o + cnt], err_msg=n)
        o += cnt
    if loop == "mamdr":
        for k in range(3):
            o = 0
            for n in chosen:
                off, cnt = segs[n]
                np.testing.assert_array_equal(phis[k].numpy()[off - lo:off - lo + cnt], phis_o[k][o:o + cnt], err_msg=n)
                o += cnt


def test_scattered_meta_parms_through_the_run_entry(tmp_path, monkeypatch):
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp_meta_domain_negotiation", epochs=2)
    cfg["train"]["meta_parms"] = ["W0", "b2"]                 # W1, W2, b0, b1 sit between them: one hole
    avg_loss, avg_auc, _, da = cli.main(cfg, FakeEngine)
    assert np.isfinite(avg_loss) and sorted(da) == [0, 1, 2]
    for name in ("mlp_meta_mamdr", "mlp_meta", "mlp_pcgrad"):
        cfg = tiny_config(tmp_path, name, epochs=2)
        cfg["train"]["meta_parms"] = ["domain_emb", "W1", "gb"]
        avg_loss, avg_auc, _, da = cli.main(cfg, FakeEngine)
        assert np.isfinite(avg_loss) and sorted(da) == [0, 1, 2]
    cfg = tiny_config(tmp_path, "mlp_meta_domain_negotiation", epochs=1)
    cfg["train"]["meta_parms"] = ["W1", "W2"]                 # neighbours: no hole
    avg_loss, avg_auc, _, da = cli.main(cfg, FakeEngine)
    assert np.isfinite(avg_loss) and sorted(da) == [0, 1, 2]


def test_configs_mirror_the_reference():
    """config/ carries the reference's 40 run configurations value for value (every key of the reference's file, same
    value; this build only ADDS keys: `dataset.synthetic`, defaults the reference's loader fills in) plus three BASELINE
    cases the reference has no file for.  Compared against /root/reference where it exists (this container)."""
    ref = "/root/reference/config"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not here")
    ours = os.path.join(ROOT, "config")
    n = 0
    for split in sorted(os.listdir(ref)):
        for name in sorted(os.listdir(os.path.join(ref, split))):
            if not name.endswith(".json"):
                continue
            with open(os.path.join(ref, split, name)) as f:
                a = json.load(f)
            with open(os.path.join(ours, split, name)) as f:
                b = json.load(f)
            for sec, kv in a.items():
                for k, v in kv.items():
                    assert b[sec][k] == v, (split, name, sec, k, v, b[sec].get(k))
            n += 1
    assert n == 40
    extra = {"Amazon_13/star_DN+DR.json", "Amazon_6/deepfm_DN.json", "Taobao_30/deepctr_DN+DR_bs4096.json"}
    have = {"%s/%s" % (s, f) for s in os.listdir(ours) for f in os.listdir(os.path.join(ours, s)) if f.endswith(".json")}
    assert extra <= have and len(have) == 43


def test_compiled_optimizer_strings(tmp_path, monkeypatch):
    """deepctr.py:54-57 / star.py:26-30 / deep_mtl_ctr.py:53-56: train.optimizer "adam" compiles tf.train.AdamOptimizer at
    `learning_rate`; any other value reaches Keras as a STRING = that Keras optimiser at ITS defaults.  "sgd" (SGD, lr 0.01,
    no momentum, whatever learning_rate says) is built: every step of the wrappers runs it; other names raise."""
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp_meta_domain_negotiation", epochs=1)
    cfg["train"].update(optimizer="sgd", learning_rate=0.5)
    ds = mds.MultiDomainDataset(cfg["dataset"])
    model = cli.build_model(cfg, ds, FakeEngine)
    model.train()
    calls = model.model.calls
    assert calls and all(c[2] == "sgd" and c[3] == 0.01 for c in calls), calls[:3]
    cfg["train"]["optimizer"] = "rmsprop"
    with pytest.raises(NotImplementedError, match="rmsprop"):
        cli.build_model(cfg, ds, FakeEngine)


def test_star_plain_dnn_form(tmp_path, monkeypatch):
    """star.py:74-87 with `norm: "none"`, `dense: "dense"`: the plain-DNN form of the Star model = the mlp tower without
    dropout and regularisers, Keras names for the name filters (`dense/kernel` ...), under the plain loop and the meta
    wrappers; the mixed forms and BatchNormalization name what is built."""
    from fake_engine import fake_factory
    patch_emb_dim(monkeypatch)
    for name, extra in (("star", {}), ("star_meta_mamdr", {"meta_parms": ["emb", "kernel"]})):
        cfg = tiny_config(tmp_path / name, name)
        cfg["model"].update(norm="none", dense="dense", auxiliary_net=False, dropout=0.5)       # (a dropout key is ignored: star.py has no Dropout layer)
        cfg["train"].update(extra)
        built = []
        avg_loss, avg_auc, dl, da = cli.main(cfg, fake_factory, on_model=built.append)
        eng = built[0].model
        assert eng.tower == "mlp" and sorted(da) == [0, 1, 2] and np.isfinite(avg_loss)
        assert eng.oracle.dropout == 0.0 and eng.oracle.l2_emb == 0.0
        names = [eng.keras_name(s) for s in eng.segments]
        assert names[:4] == ["domain_emb/embeddings", "dense/kernel", "dense_1/kernel", "dense_2/kernel"]
        assert "dense_3/kernel" in names and "dense_3/bias" in names
        if extra:       # the filter took the embeddings and the four kernels, not the biases
            picked = [s for s in eng.segments if any(k in eng.keras_name(s) for k in extra["meta_parms"])]
            assert picked == ["domain_emb", "W0", "W1", "W2", "wo"]
    for norm, dense in (("bn", "dense"), ("pn", "dense"), ("none", "star")):
        cfg = tiny_config(tmp_path / ("x" + norm + dense), "star")
        cfg["model"].update(norm=norm, dense=dense, auxiliary_net=False)
        with pytest.raises(NotImplementedError, match="plain form"):
            cli.main(cfg, fake_factory)
