"""Independent cross-check of the CPU oracle's hand-derived backward passes (CPU only, no HIP).

oracle/tower.py and oracle/star.py compute every gradient by hand in numpy fp32.  oracle/torch_ref.py writes
down only the FORWARD of the same towers and lets torch.autograd differentiate it in
float64.  Agreement here removes the single-derivation risk of the oracle; it does not pin it to TF
(tensorflow-gpu==1.12.0 / deepctr==0.9.0 are not installable: parity of the inner step stays unpinned).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import rng as orng            # noqa: E402
from oracle import star as ostar          # noqa: E402
from oracle import torch_ref as tref      # noqa: E402
from oracle import tower as otower        # noqa: E402

F32 = np.float32


def _batch(rs, n_user, n_item, n_domain, B, single_domain=None):
    uid = rs.randint(0, n_user, B).astype(np.int32)
    pid = rs.randint(0, n_item, B).astype(np.int32)
    uid[: B // 8] = uid[0]                  # repeated rows: the scatter-add paths must sum them
    pid[B // 2:B // 2 + 5] = pid[1]
    dom = (np.full(B, single_domain) if single_domain is not None else rs.randint(0, n_domain, B)).astype(np.int32)
    label = (rs.uniform(size=B) < 0.3).astype(F32)
    return uid, pid, dom, label


def _params(rs, n_user, n_item, n_domain, tower, uncertainty, hidden=(256, 128, 64)):
    p = otower.init_params(rs, n_user, n_item, n_domain, hidden=hidden)
    p["domain_emb"] = (rs.standard_normal(p["domain_emb"].shape) * 0.05).astype(F32)
    for l in range(len(hidden)):
        p["b%d" % l] = (rs.standard_normal(p["b%d" % l].shape) * 0.05).astype(F32)
    p["gb"] = np.array([0.1], F32)
    if tower in ("deepfm", "wdl"):
        for n in ("lin_user", "lin_item", "lin_domain"):
            p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    if uncertainty:
        p["log_var"] = (1.0 + rs.uniform(-0.3, 0.3, n_domain)).astype(F32)
    return p


def _check_grads(got32, want64, names, rtol=3e-4):
    for n in names:
        g, w = np.asarray(got32[n], np.float64), np.asarray(want64[n], np.float64)
        assert g.shape == w.shape, n
        scale = max(np.abs(w).max(), 1e-12)
        # fp32 contractions over <= 384 terms and <= 256 rows against float64: a few 1e-6 of the tensor's scale
        np.testing.assert_allclose(g, w, rtol=rtol, atol=3e-6 * scale, err_msg=n)
        rel = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-30)
        # (the uncertainty scalar's gradient -2 mean(BCE) / var^3 + 1 / var is a difference of two O(1) fp32 terms: its
        # error is a few ulps of the TERMS, whatever is left of them)
        assert rel < (2e-6 if n != "log_var" else 2e-6 * max(1.0, 2.0 / max(np.linalg.norm(w), 1e-30))), (n, rel)


@pytest.mark.parametrize("tower,emb_trainable,rate,uncertainty", [
    ("mlp", False, 0.5, False), ("mlp", True, 0.5, False), ("mlp", False, 0.0, False), ("mlp", False, 0.5, True),
    ("deepfm", True, 0.5, False), ("deepfm", False, 0.0, False), ("wdl", True, 0.5, False)])
def test_deepctr_towers_gradients_vs_float64_autograd(tower, emb_trainable, rate, uncertainty):
    rs = np.random.RandomState(11)
    n_user, n_item, n_domain, B = 300, 200, 5, 192
    p = _params(rs, n_user, n_item, n_domain, tower, uncertainty)
    uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B, single_domain=2 if uncertainty else None)
    deepfm = {"deepfm": 1, "wdl": 2}.get(tower, 0)
    names = otower.param_names(emb_trainable, deepfm, uncertainty)
    masks = otower.train_masks(1024, 3, B, (256, 128, 64), rate) if rate > 0 else None
    loss32, g32, p32 = otower.loss_and_grads(p, uid, pid, dom, label, masks, rate, emb_trainable, None, deepfm,
                                             uncertainty)
    loss64, g64, p64, _ = tref.loss_and_grads(p, names, uid, pid, dom, label, masks, rate, tower, uncertainty)
    assert abs(float(loss32) - loss64) < 2e-6 * max(1.0, abs(loss64))
    np.testing.assert_allclose(p32, p64, rtol=2e-5, atol=2e-7)
    _check_grads(g32, g64, names)


@pytest.mark.parametrize("tower,emb_trainable,hidden", [("mlp", False, (128, 64)), ("deepfm", True, (256, 128, 64, 64)),
                                                        ("wdl", True, (64,)), ("deepfm", False, (128, 128, 64))])
def test_deepctr_towers_other_hidden_dims_vs_float64_autograd(tower, emb_trainable, hidden):
    """oracle/tower.py with 1, 2 and 4 hidden layers (round 5: `hidden_dim` lists other than the reference configs'
    [256, 128, 64] run on the generic-layer engine; deepctr.py:26-49 passes any list through): the hand-derived gradients
    against float64 autograd of the independently written forward, tensor order = kernels, biases, wo, gb."""
    rs = np.random.RandomState(21)
    n_user, n_item, n_domain, B = 300, 200, 5, 192
    p = _params(rs, n_user, n_item, n_domain, tower, False, hidden)
    uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B)
    deepfm = {"deepfm": 1, "wdl": 2}.get(tower, 0)
    names = otower.param_names(emb_trainable, deepfm, False, len(hidden))
    L = len(hidden)
    assert names[-(2 * L + 2 + (1 if deepfm else 0)):][:2 * L + 2] == tuple("W%d" % l for l in range(L)) + \
        tuple("b%d" % l for l in range(L)) + ("wo", "gb")
    masks = otower.train_masks(1024, 3, B, hidden, 0.5)
    assert len(masks) == L and [m.shape[1] for m in masks] == list(hidden)
    loss32, g32, p32 = otower.loss_and_grads(p, uid, pid, dom, label, masks, 0.5, emb_trainable, None, deepfm)
    loss64, g64, p64, _ = tref.loss_and_grads(p, names, uid, pid, dom, label, masks, 0.5, tower, False)
    assert abs(float(loss32) - loss64) < 2e-6 * max(1.0, abs(loss64))
    np.testing.assert_allclose(p32, p64, rtol=2e-5, atol=2e-7)
    assert sorted(g32) == sorted(names)
    _check_grads(g32, g64, names)
    # ... and a whole OracleModel of that shape steps (TF1 Adam over every named tensor) and evaluates
    model = otower.OracleModel({k: v.copy() for k, v in p.items()}, emb_trainable=emb_trainable, dropout=0.5, hidden=hidden,
                               tower=tower)
    assert model.names == names
    w0 = model.get_flat().copy()
    model.train_on_batch(uid, pid, dom, label)
    assert model.get_flat().shape == w0.shape and np.all(np.isfinite(model.get_flat())) and np.any(model.get_flat() != w0)


@pytest.mark.parametrize("emb_trainable", [True, False])
def test_star_tower_gradients_vs_float64_autograd(emb_trainable):
    rs = np.random.RandomState(12)
    n_user, n_item, n_domain, B = 300, 200, 4, 192
    p = ostar.init_params(rs, n_user, n_item, n_domain)
    for n in ("pn_gamma_shared", "pn_gamma_spec"):
        p[n] = (p[n] + rs.standard_normal(p[n].shape) * 0.2).astype(F32)
    for n in ("pn_beta_shared", "pn_beta_spec", "bs0", "bs1", "bs2", "bd0", "bd1", "bd2", "gb"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    for l in range(3):
        p["Wd%d" % l] = (p["Wd%d" % l] * 8).astype(F32)
    uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B, single_domain=1)
    meta, rest = ostar.param_names(emb_trainable)
    names = meta + rest
    state = ostar.init_state(n_domain)
    loss32, g32, p32, c = ostar.loss_and_grads(p, state, uid, pid, dom, label, emb_trainable)
    loss64, g64, p64, extra = tref.loss_and_grads(p, names, uid, pid, dom, label, tower="star")
    assert abs(float(loss32) - loss64) < 2e-6 * max(1.0, abs(loss64))
    np.testing.assert_allclose(p32, p64, rtol=5e-5, atol=5e-7)
    np.testing.assert_allclose(c["mean"], extra["mean"], rtol=1e-5, atol=1e-7)       # batch statistics
    np.testing.assert_allclose(c["var"], extra["var"], rtol=1e-5, atol=1e-9)
    # the domain row is constant over a single-domain batch: PartitionedNorm maps it to beta, its gradient is
    # rounding residue in fp32 and exactly ~0 in float64
    assert np.abs(g32["domain_emb"]).max() < 1e-5 and np.abs(g64["domain_emb"]).max() < 1e-12
    _check_grads(g32, g64, [n for n in names if n != "domain_emb"], rtol=1e-3)
    # the slices of the other domains carry exactly zero gradient on both sides
    for n in ("Wd0", "bd1", "pn_gamma_spec"):
        assert not np.any(g32[n][0]) and not np.any(g64[n][0])


def test_tf1_adam_step_formula_vs_float64():
    """oracle/tower.Optimizer.adam (fp32, ApplyAdam operation order) against the float64 textbook form of the same
    update from the SAME gradients: first step from zero slots and a later step from non-zero slots."""
    rs = np.random.RandomState(13)
    names = ("a", "b")
    p = {"a": rs.standard_normal((40, 30)).astype(F32), "b": rs.standard_normal(17).astype(F32)}
    opt = otower.Optimizer(p, names)
    for t in range(1, 4):
        g = {n: (rs.standard_normal(p[n].shape) * 10.0 ** rs.randint(-6, 1)).astype(F32) for n in names}
        m0 = {n: opt.m[n].copy() for n in names}
        v0 = {n: opt.v[n].copy() for n in names}
        want = tref.adam_step(p, g, names, 1e-3, t=t, m=m0, v=v0)
        before = {n: p[n].copy() for n in names}
        opt.adam(p, g, 1e-3)
        for n in names:
            step = np.abs(want[n] - before[n]).max()
            np.testing.assert_allclose(p[n], want[n], rtol=0, atol=2e-7 * np.abs(before[n]).max() + 1e-4 * step)


def test_oracle_step_equals_autograd_then_adam_end_to_end():
    """one whole train_on_batch of the oracle (gradients by hand, fp32 Adam) against float64 autograd gradients
    pushed through the float64 Adam step: all but a sliver of the elements within 2 % of lr (Adam's first step
    moves every element by ~lr * sign(g); elements whose gradient is rounding noise may differ by up to 2 lr)."""
    rs = np.random.RandomState(14)
    n_user, n_item, n_domain, B = 300, 200, 5, 192
    p = _params(rs, n_user, n_item, n_domain, "deepfm", False)
    uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B)
    names = otower.param_names(True, 1, False)
    model = otower.OracleModel({k: v.copy() for k, v in p.items()}, emb_trainable=True, dropout=0.5, lr=1e-3,
                               tower="deepfm")
    masks = otower.train_masks(model.seed, 0, B, (256, 128, 64), 0.5)
    _, g64, _, _ = tref.loss_and_grads(p, names, uid, pid, dom, label, masks, 0.5, "deepfm")
    want = tref.adam_step(p, g64, names, 1e-3, t=1)
    model.train_on_batch(uid, pid, dom, label)
    for n in names:
        diff = np.abs(model.params[n].astype(np.float64) - want[n])
        assert diff.max() <= 2.02e-3, n
        assert np.mean(diff > 2e-5) < 2e-3, (n, float(np.mean(diff > 2e-5)))


def test_torch_cpu_model_follows_the_oracle():
    """bench.py's CPU baseline model (fp32 torch, autograd, dense TF1 Adam) takes the same steps as the oracle
    when dropout is off (its dropout masks come from torch's generator): three steps, weights within Adam noise."""
    rs = np.random.RandomState(15)
    n_user, n_item, n_domain, B = 300, 200, 5, 128
    p = _params(rs, n_user, n_item, n_domain, "mlp", False)
    names = otower.param_names(False)
    cpu = tref.TorchCpuModel(p, names, tower="mlp", dropout=0.0, lr=1e-3)
    model = otower.OracleModel({k: v.copy() for k, v in p.items()}, emb_trainable=False, dropout=0.0, lr=1e-3)
    for s in range(3):
        uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B)
        l_cpu = cpu.train_on_batch(uid, pid, dom, label)
        l_ora = model.train_on_batch(uid, pid, dom, label)
        assert abs(l_cpu - float(l_ora)) < 1e-5
    for n in names:
        diff = np.abs(cpu.P[n].detach().numpy() - model.params[n])
        assert np.mean(diff > 1.5e-4) < 2e-3 and diff.max() <= 6.1e-3, n


# ------------------------------------------------------------------ multi-task towers (oracle/mtl.py)
@pytest.mark.parametrize("kind,emb_trainable,rate", [
    ("shared_bottom", False, 0.5), ("shared_bottom", True, 0.0), ("mmoe", False, 0.5), ("mmoe", True, 0.5),
    ("ple", False, 0.5), ("ple", False, 0.0)])
def test_mtl_towers_gradients_vs_float64_autograd(kind, emb_trainable, rate):
    """oracle/mtl.py (deepctr SharedBottom / MMOE / PLE under deep_mtl_ctr.py's per-domain models): the hand-derived
    gradients of task d -- experts, softmax gate and its DNN, tower, head, domain table, trainable tables with
    repeated rows -- against float64 autograd of oracle/torch_ref.mtl_forward; tensors outside task d's model get no
    gradient on either side."""
    from oracle import mtl as omtl
    rs = np.random.RandomState(23)
    n_user, n_item, D, B = 60, 40, 4, 48
    spec = omtl.Spec(kind, D, expert_hidden=(32, 16), tower_hidden=(8,), gate_hidden=(8,), num_experts=3,
                     shared_expert_num=2, specific_expert_num=2, emb_dim=8)
    p = omtl.init_params(rs, spec, n_user, n_item)
    for n in p:
        if "/b" in n or n.endswith("/gb") or n == "domain_emb":
            p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    names = [n for n, _ in spec.tensors(emb_trainable, n_user, n_item)]
    for d in (0, 2):
        uid, pid, dom, label = _batch(rs, n_user, n_item, D, B)
        masks = omtl.train_masks(spec, 1024, 5, B, rate) if rate > 0 else None
        loss, g, pred = omtl.loss_and_grads(p, spec, d, uid, pid, dom, label, masks, rate, emb_trainable)
        loss64, g64, pred64 = tref.mtl_loss_and_grads(p, names, spec, d, uid, pid, dom, label, masks, rate)
        assert abs(float(loss) - loss64) < 2e-6 * max(1.0, abs(loss64))
        np.testing.assert_allclose(pred, pred64, rtol=2e-5, atol=2e-7)
        on_path = [n for n in names if g64[n] is not None]
        assert sorted(on_path) == sorted(g), (sorted(set(on_path) ^ set(g)))
        keep, owners = spec.task_tensors(d, emb_trainable)
        assert all(n in keep or n.split("/")[0] in owners for n in on_path)
        _check_grads(g, g64, on_path)


def test_mtl_adam_shares_the_beta_powers_between_the_domain_models():
    """ONE tf.train.AdamOptimizer for all D compiled models (deep_mtl_ctr.py:53-66): a step of any domain advances the
    beta powers, only that domain's variables move; first step of a variable moves it by ~alpha_t (m / sqrt(v) = 1)."""
    from oracle import mtl as omtl
    rs = np.random.RandomState(4)
    spec = omtl.Spec("mmoe", 3, (16,), (8,), (8,), num_experts=2, emb_dim=8)
    p = omtl.init_params(rs, spec, 30, 20)
    model = omtl.OracleMTL({k: v.copy() for k, v in p.items()}, spec, dropout=0.0, lr=1e-3)
    uid, pid, dom, label = _batch(rs, 30, 20, 3, 32)
    model.train_on_batch(0, uid, pid, dom, label)
    assert np.array_equal(model.params["tower_1/W0"], p["tower_1/W0"]) and not np.array_equal(model.params["tower_0/W0"], p["tower_0/W0"])
    model.train_on_batch(1, uid, pid, dom, label)
    assert model.t == 2
    # tower_1's first own step happens at t = 2: |dp| = alpha_2 * (0.1 g) / (sqrt(0.001 g^2) + eps) for every element with g != 0
    a2 = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    moved = np.abs(model.params["tower_1/W0"] - p["tower_1/W0"])
    nz = moved > 0
    want = a2 * 0.1 / np.sqrt(0.001)            # (elements with a gradient near eps = 1e-8 move a little less)
    assert abs(np.median(moved[nz]) - want) < 2e-3 * want and moved.max() <= want * 1.002 and moved[nz].min() > 0.9 * want


# ------------------------------------------------------------------ NFM / PNN (oracle/fmnets.py)
@pytest.mark.parametrize("kind,emb_trainable,rate,uncertainty", [
    ("nfm", False, 0.5, False), ("nfm", True, 0.5, False), ("pnn", False, 0.5, False), ("pnn", True, 0.0, False),
    ("nfm", True, 0.5, True), ("pnn", False, 0.5, True)])
def test_nfm_pnn_gradients_vs_float64_autograd(kind, emb_trainable, rate, uncertainty):
    from oracle import fmnets as ofm
    rs = np.random.RandomState(31)
    n_user, n_item, D, B = 60, 40, 4, 48
    p = ofm.init_params(rs, kind, n_user, n_item, D, emb_dim=8, hidden=(16, 8, 4))
    p["domain_emb"] = (rs.standard_normal(p["domain_emb"].shape) * 0.05).astype(F32)
    for n in ("b0", "b1", "b2", "lin_user", "lin_item", "lin_domain"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    p["gb"] = np.array([0.1], F32)      # (a logit of exactly 0 -- all four last units dropped -- sits on the kinks of
    #                                      torch's clamp / abs in keras_bce, where autograd's subgradient is not the derivative)
    names = list(ofm.param_names(kind, emb_trainable, uncertainty))
    if uncertainty:            # distinct per-domain scales around the initial value 1 (weighted_loss.py:23-28)
        p["log_var"] = (1.0 + rs.uniform(-0.3, 0.3, D)).astype(F32)
    uid, pid, dom, label = _batch(rs, n_user, n_item, D, B, single_domain=2 if uncertainty else None)
    masks = otower.train_masks(1024, 3, B, (16, 8, 4), rate) if rate > 0 else None
    loss, g, pred = ofm.loss_and_grads(p, kind, uid, pid, dom, label, masks, rate, emb_trainable, None, uncertainty)
    loss64, g64, pred64 = tref.fmnet_loss_and_grads(p, names, kind, uid, pid, dom, label, masks, rate, uncertainty=uncertainty)
    assert abs(float(loss) - loss64) < 2e-6 * max(1.0, abs(loss64))
    np.testing.assert_allclose(pred, pred64, rtol=2e-5, atol=2e-7)
    assert sorted(g) == sorted(names)
    _check_grads(g, g64, names)


@pytest.mark.parametrize("kind,emb_trainable,rate,uncertainty", [
    ("ccpm", False, 0.5, False), ("ccpm", True, 0.0, False), ("autoint", False, 0.5, False), ("autoint", True, 0.5, False),
    ("ccpm", False, 0.5, True), ("autoint", True, 0.5, True)])
def test_ccpm_autoint_gradients_vs_float64_autograd(kind, emb_trainable, rate, uncertainty):
    """oracle/fmnets.py's CCPM (convolution over the field axis, max over the fields, tanh) and AutoInt (three multi-head
    self-attention layers with residuals) -- hand-derived fp32 backward passes -- against float64 autograd of a forward
    written with torch's own conv2d / softmax."""
    from oracle import fmnets as ofm
    rs = np.random.RandomState(37)
    n_user, n_item, D, B = 60, 40, 4, 48
    E = 128 if kind == "autoint" else 8            # (the attention layers' second and third inputs are 32 wide whatever E is)
    E = 8
    p = ofm.init_params_conv(rs, kind, n_user, n_item, D, emb_dim=E, hidden=(16, 8, 4))
    p["domain_emb"] = (rs.standard_normal(p["domain_emb"].shape) * 0.05).astype(F32)
    for n in ("b0", "b1", "b2", "lin_user", "lin_item", "lin_domain"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    if kind == "ccpm":
        p["conv1_b"] = (rs.standard_normal(4) * 0.1).astype(F32)
        p["conv2_b"] = (rs.standard_normal(4) * 0.1).astype(F32)
    else:
        for l in range(3):                         # larger kernels: the softmax is off its uniform point
            p["att%d_w" % l] = (p["att%d_w" % l] * 8).astype(F32)
    p["gb"] = np.array([0.1], F32)
    names = list(ofm.ccpm_param_names(emb_trainable, uncertainty) if kind == "ccpm"
                 else ofm.autoint_param_names(emb_trainable, uncertainty))
    if uncertainty:            # distinct per-domain scales around the initial value 1 (weighted_loss.py:23-28)
        p["log_var"] = (1.0 + rs.uniform(-0.3, 0.3, D)).astype(F32)
    uid, pid, dom, label = _batch(rs, n_user, n_item, D, B, single_domain=2 if uncertainty else None)
    masks = otower.train_masks(1024, 3, B, (16, 8, 4), rate) if rate > 0 else None
    loss, g, pred = ofm.loss_and_grads_conv(p, kind, uid, pid, dom, label, masks, rate, emb_trainable, None, uncertainty)
    loss64, g64, pred64 = tref.convnet_loss_and_grads(p, names, kind, uid, pid, dom, label, masks, rate, uncertainty=uncertainty)
    assert abs(float(loss) - loss64) < 2e-6 * max(1.0, abs(loss64))
    np.testing.assert_allclose(pred, pred64, rtol=2e-5, atol=2e-7)
    assert sorted(g) == sorted(names)
    _check_grads(g, g64, names)


# ---------------------------------------------------------------------------------------------- a third-party anchor
# scikit-learn ships an independent implementation of BOTH halves of the inner step's dense part: a relu MLP with a
# logistic output under the mean binary log-loss (sklearn.neural_network.MLPClassifier._backprop) and the Adam update
# in exactly TensorFlow 1's operation form, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps)
# (sklearn.neural_network._stochastic_optimizers.AdamOptimizer; TF 1.12 python/training/adam.py documents the same
# "epsilon hat" form).  It is not the reference's TensorFlow -- the inner step stays "parity unpinned" -- but it is code
# neither this repository nor its author wrote, float64, and installed here.
def _sklearn_mlp(params, hidden):
    sk = pytest.importorskip("sklearn.neural_network")
    clf = sk.MLPClassifier(hidden_layer_sizes=tuple(hidden), activation="relu", alpha=0.0, solver="adam")
    L = len(hidden)
    clf.n_layers_ = L + 2
    clf.n_outputs_ = 1
    clf.out_activation_ = "logistic"
    clf.coefs_ = [params["W%d" % l].astype(np.float64) for l in range(L)] + [params["wo"].astype(np.float64)]
    clf.intercepts_ = [params["b%d" % l].astype(np.float64) for l in range(L)] + [params["gb"].astype(np.float64)]
    return clf


@pytest.mark.parametrize("hidden", [(256, 128, 64), (64, 32)])
def test_mlp_tower_loss_and_gradients_vs_sklearn_mlp(hidden):
    """oracle/tower.loss_and_grads (mlp tower, no dropout) against scikit-learn's MLPClassifier._backprop on the same
    gathered rows and the same weights: mean BCE and every dense gradient (W_l, b_l, output unit)."""
    rs = np.random.RandomState(21)
    n_user, n_item, n_domain, B = 200, 150, 4, 160
    p = _params(rs, n_user, n_item, n_domain, "mlp", False, hidden=hidden)
    uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B)
    loss, g, pred = otower.loss_and_grads(p, uid, pid, dom, label, None, 0.0, False)
    X = otower.gather(p, uid, pid, dom).astype(np.float64)
    clf = _sklearn_mlp(p, hidden)
    units = [X.shape[1]] + list(hidden) + [1]
    acts = [X] + [np.empty((B, u)) for u in units[1:]]
    deltas = [np.empty_like(a) for a in acts[1:]]
    cg = [np.empty((a, b)) for a, b in zip(units[:-1], units[1:])]
    ig = [np.empty(b) for b in units[1:]]
    sk_loss, cg, ig = clf._backprop(X, label.astype(np.float64).reshape(-1, 1), None, acts, deltas, cg, ig)
    reg = float(otower.reg_loss(p))
    assert abs((float(loss) - reg) - sk_loss) <= 2e-6 * max(1.0, abs(sk_loss)), (float(loss) - reg, sk_loss)
    np.testing.assert_allclose(pred, acts[-1][:, 0], rtol=2e-5, atol=2e-7)
    L = len(hidden)
    for l in range(L):
        np.testing.assert_allclose(g["W%d" % l], cg[l], rtol=2e-4, atol=2e-6 * np.abs(cg[l]).max(), err_msg="W%d" % l)
        np.testing.assert_allclose(g["b%d" % l], ig[l], rtol=2e-4, atol=2e-6 * np.abs(ig[l]).max(), err_msg="b%d" % l)
    np.testing.assert_allclose(g["wo"], cg[L], rtol=2e-4, atol=2e-6 * np.abs(cg[L]).max())
    np.testing.assert_allclose(g["gb"], ig[L], rtol=2e-4, atol=1e-7)


def test_tf1_adam_equals_sklearn_adam_over_several_steps():
    """oracle/tower.Optimizer.adam (fp32) against scikit-learn's AdamOptimizer (float64) fed the same gradient sequence:
    parameters after every one of six steps, from zero slots, with gradient scales spanning six decades."""
    so = pytest.importorskip("sklearn.neural_network._stochastic_optimizers")
    rs = np.random.RandomState(22)
    names = ("a", "b")
    p = {"a": rs.standard_normal((24, 16)).astype(F32), "b": rs.standard_normal(9).astype(F32)}
    ref = [p[n].astype(np.float64).copy() for n in names]
    sk = so.AdamOptimizer(ref, learning_rate_init=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8)
    opt = otower.Optimizer(p, names)
    for t in range(6):
        g = {n: (rs.standard_normal(p[n].shape) * 10.0 ** rs.randint(-6, 1)).astype(F32) for n in names}
        before = {n: p[n].copy() for n in names}
        opt.adam(p, g, 1e-3)
        sk.update_params(ref, [g[n].astype(np.float64) for n in names])
        for k, n in enumerate(names):
            step = np.abs(ref[k] - before[n]).max()
            np.testing.assert_allclose(p[n], ref[k], rtol=0, atol=(t + 1) * (2e-7 * np.abs(before[n]).max() + 1e-4 * step), err_msg=n)


def test_oracle_training_steps_follow_sklearn_end_to_end():
    """Five whole oracle training steps of the mlp tower's dense block (gradients by hand, fp32 TF1 Adam; dropout off, frozen
    tables) against scikit-learn's own _backprop + AdamOptimizer on the same batches: the loss of every step and the
    weights after the fifth."""
    so = pytest.importorskip("sklearn.neural_network._stochastic_optimizers")
    rs = np.random.RandomState(23)
    hidden = (256, 128, 64)
    n_user, n_item, n_domain, B = 200, 150, 4, 128
    p = _params(rs, n_user, n_item, n_domain, "mlp", False, hidden=hidden)
    model = otower.OracleModel({k: v.copy() for k, v in p.items()}, emb_trainable=False, dropout=0.0, lr=1e-3)
    clf = _sklearn_mlp(p, hidden)
    L = len(hidden)
    sk_params = clf.coefs_ + clf.intercepts_
    sk = so.AdamOptimizer(sk_params, learning_rate_init=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8)
    dm0 = p["domain_emb"].copy()
    for step in range(5):
        uid, pid, dom, label = _batch(rs, n_user, n_item, n_domain, B, single_domain=step % n_domain)
        # (the domain table trains in the oracle -- the reference's frozen-table configs keep it trainable -- and not in the
        # scikit-learn net, whose input is a plain matrix: its rows are taken from the oracle's current table)
        X = otower.gather(model.params, uid, pid, dom).astype(np.float64)
        units = [X.shape[1]] + list(hidden) + [1]
        acts = [X] + [np.empty((B, u)) for u in units[1:]]
        deltas = [np.empty_like(a) for a in acts[1:]]
        cg = [np.empty((a, b)) for a, b in zip(units[:-1], units[1:])]
        ig = [np.empty(b) for b in units[1:]]
        sk_loss, cg, ig = clf._backprop(X, label.astype(np.float64).reshape(-1, 1), None, acts, deltas, cg, ig)
        reg = float(otower.reg_loss(model.params, model.frozen_sumsq()))
        loss = model.train_on_batch(uid, pid, dom, label)
        assert abs((float(loss) - reg) - sk_loss) <= 5e-5 * max(1.0, abs(sk_loss)), (step, float(loss) - reg, sk_loss)
        sk.update_params(sk_params, cg + ig)
    assert not np.array_equal(model.params["domain_emb"], dm0)
    for l in range(L):
        d = np.abs(model.params["W%d" % l] - clf.coefs_[l])
        # (five Adam steps move an element by <= 5 lr; elements whose gradient is rounding noise may differ by a step)
        assert d.max() <= 2.02e-3 and np.mean(d > 1e-4) < 5e-3, (l, float(d.max()), float(np.mean(d > 1e-4)))
    d = np.abs(model.params["wo"] - clf.coefs_[L])
    assert d.max() <= 2.02e-3 and np.mean(d > 1e-4) < 2e-2, (float(d.max()), float(np.mean(d > 1e-4)))


def test_partitioned_norm_statement_equals_torch_batch_norm():
    """PartitionedNorm in training (partitioned_norm.py:143-174: nn.moments = population variance, eps 1e-3,
    gamma_shared * gamma_specific[d], beta_shared + beta_specific[d]) as oracle/torch_ref.star_forward writes it down -- the
    form oracle/star.py's hand-derived backward is held to above -- against PyTorch's own batch_norm kernel with the same
    affine: forward values and the gradient through the batch statistics (float64)."""
    import torch.nn.functional as Fn
    rs = np.random.RandomState(31)
    B, C = 96, 384
    x = torch.tensor(rs.standard_normal((B, C)) * 0.3 + 0.1, dtype=torch.float64, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    gs, gd = (torch.tensor(1 + 0.2 * rs.standard_normal(C), dtype=torch.float64) for _ in range(2))
    bs, bd = (torch.tensor(0.05 * rs.standard_normal(C), dtype=torch.float64) for _ in range(2))
    w = torch.tensor(rs.standard_normal((B, C)), dtype=torch.float64)
    mean = x.mean(dim=0)
    var = ((x - mean) ** 2).mean(dim=0)
    h = (x - mean) * torch.rsqrt(var + tref.PN_EPS) * (gs * gd) + (bs + bd)          # star_forward's statement
    ref = Fn.batch_norm(x2, None, None, weight=gs * gd, bias=bs + bd, training=True, eps=float(tref.PN_EPS))
    np.testing.assert_allclose(h.detach().numpy(), ref.detach().numpy(), rtol=1e-12, atol=1e-13)
    (h * w).sum().backward()
    (ref * w).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), x2.grad.numpy(), rtol=1e-9, atol=1e-12)
