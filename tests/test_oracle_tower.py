"""Self-consistency of the inner-step restatement (oracle/tower.py): analytic
gradients vs fp64 finite differences, Adam closed form on step 1, rng streams."""
import numpy as np

from oracle import rng, tower

F32 = np.float32


def small_problem(seed=0, B=32, n_user=50, n_item=40, n_domain=3, emb=8, hidden=(16, 8, 4)):
    rs = np.random.RandomState(seed)
    p = tower.init_params(rs, n_user, n_item, n_domain, emb, hidden)
    p["domain_emb"] = (rs.standard_normal((n_domain, emb)) * 0.1).astype(F32)
    for l in range(3):
        p["b%d" % l] = (rs.standard_normal(hidden[l]) * 0.1).astype(F32)
    uid = rs.randint(0, n_user, B).astype(np.int32)
    pid = rs.randint(0, n_item, B).astype(np.int32)
    dom = rs.randint(0, n_domain, B).astype(np.int32)
    y = (rs.rand(B) < 0.4).astype(F32)
    return p, uid, pid, dom, y, hidden


def loss64(p, uid, pid, dom, y, masks, scale):
    """independent fp64 forward for finite differences."""
    q = {k: v.astype(np.float64) for k, v in p.items()}
    h = np.concatenate([q["user_emb"][uid], q["item_emb"][pid], q["domain_emb"][dom]], 1)
    for l in range(3):
        h = np.maximum(h @ q["W%d" % l] + q["b%d" % l], 0) * scale * masks[l]
    z = (h @ q["wo"])[:, 0] + q["gb"][0]
    ce = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    reg = 1e-5 * sum((q[n] ** 2).sum() for n in ("user_emb", "item_emb", "domain_emb"))
    return ce.mean() + reg


def test_gradients_match_finite_differences():
    p, uid, pid, dom, y, hidden = small_problem()
    masks = tower.train_masks(7, 3, len(uid), hidden, 0.5)
    loss, g, _ = tower.loss_and_grads(p, uid, pid, dom, y, masks, 0.5, emb_trainable=True)
    assert abs(float(loss) - loss64(p, uid, pid, dom, y, masks, 2.0)) < 1e-5
    rs = np.random.RandomState(1)
    for name in tower.param_names(True):
        a = p[name]
        for _ in range(6):
            idx = tuple(rs.randint(0, s) for s in a.shape)
            old = a[idx]
            h = 1e-6
            a[idx] = old + h
            lp = loss64(p, uid, pid, dom, y, masks, 2.0)
            a[idx] = old - h
            lm = loss64(p, uid, pid, dom, y, masks, 2.0)
            a[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - float(g[name][idx])) < 2e-4 + 2e-3 * abs(fd), (name, idx, fd, g[name][idx])


def loss64_deepfm(q, uid, pid, dom, y, masks, scale):
    """q: float64 copies of the parameters (perturbed in float64, not in fp32 storage)"""
    u, it, d = q["user_emb"][uid], q["item_emb"][pid], q["domain_emb"][dom]
    h = np.concatenate([u, it, d], 1)
    for l in range(3):
        h = np.maximum(h @ q["W%d" % l] + q["b%d" % l], 0) * scale * masks[l]
    z = (h @ q["wo"])[:, 0] + q["gb"][0]
    s = u + it + d
    z = z + 0.5 * ((s * s).sum(1) - (u * u).sum(1) - (it * it).sum(1) - (d * d).sum(1))
    z = z + q["lin_user"][uid] + q["lin_item"][pid] + q["lin_domain"][dom]
    ce = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    reg = 1e-5 * sum((q[n] ** 2).sum() for n in ("user_emb", "item_emb", "domain_emb", "lin_user", "lin_item", "lin_domain"))
    return ce.mean() + reg


def test_deepfm_gradients_match_finite_differences():
    p, uid, pid, dom, y, hidden = small_problem(seed=4)
    rs = np.random.RandomState(9)
    for n in ("lin_user", "lin_item", "lin_domain"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.1).astype(F32)
    for n in ("user_emb", "item_emb", "domain_emb"):
        p[n] = (p[n] * 3).astype(F32)
    masks = tower.train_masks(7, 3, len(uid), hidden, 0.5)
    loss, g, _ = tower.loss_and_grads(p, uid, pid, dom, y, masks, 0.5, True, None, True)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    assert abs(float(loss) - loss64_deepfm(q, uid, pid, dom, y, masks, 2.0)) < 1e-5
    assert set(tower.param_names(True, True)) == set(g) and tower.param_names(False, True)[-1] == "lin_domain"
    for name in tower.param_names(True, True):
        a = q[name]
        for _ in range(6):
            idx = tuple(rs.randint(0, s_) for s_ in a.shape)
            old = a[idx]
            h = 1e-6
            a[idx] = old + h
            lp = loss64_deepfm(q, uid, pid, dom, y, masks, 2.0)
            a[idx] = old - h
            lm = loss64_deepfm(q, uid, pid, dom, y, masks, 2.0)
            a[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - float(g[name][idx])) < 2e-4 + 2e-3 * abs(fd), (name, idx, fd, g[name][idx])


def test_adam_first_step_closed_form():
    p, uid, pid, dom, y, hidden = small_problem(seed=2)
    m = tower.OracleModel(p, emb_trainable=False, dropout=0.0, lr=1e-3, hidden=hidden)
    before = m.get_flat().copy()
    _, g, _ = tower.loss_and_grads(p, uid, pid, dom, y, None, 0.0, False)
    gflat = tower.flatten(g, m.names)
    m.train_on_batch(uid, pid, dom, y)
    after = m.get_flat()
    # step 1: m = 0.1 g, v = 0.001 g^2, alpha = lr*sqrt(0.001)/0.1 -> dp ~= lr * g/(|g| + eps')
    g64 = gflat.astype(np.float64)
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = alpha * 0.1 * g64 / (np.sqrt(0.001 * g64 * g64) + 1e-8)
    np.testing.assert_allclose(before - after, want, rtol=1e-3, atol=1e-9)
    assert m.opt.t == 1 and m.opt.b1p == F32(0.9) and m.opt.b2p == F32(0.999)
    # frozen tables untouched
    assert "user_emb" not in m.names


def test_beta_powers_running_product():
    b1, b2 = tower.beta_powers(3)
    assert b1 == F32(F32(F32(0.9) * F32(0.9)) * F32(0.9))
    assert b2 == F32(F32(F32(0.999) * F32(0.999)) * F32(0.999))


def test_dropout_stream_properties():
    u = rng.dropout_u32(1024, 5, 1, 64, 128)
    assert u.dtype == np.uint32 and u.shape == (64, 128)
    # reproducible, and different across step / layer / seed
    assert np.array_equal(u, rng.dropout_u32(1024, 5, 1, 64, 128))
    for other in (rng.dropout_u32(1024, 6, 1, 64, 128), rng.dropout_u32(1024, 5, 2, 64, 128),
                  rng.dropout_u32(1025, 5, 1, 64, 128)):
        assert np.mean(other == u) < 0.01
    keep = rng.dropout_mask(1024, 5, 1, 512, 256, 0.5)
    assert abs(keep.mean() - 0.5) < 0.01
    assert abs(rng.dropout_mask(1, 2, 0, 512, 256, 0.2).mean() - 0.8) < 0.01
    # known answers (pin the integer arithmetic the HIP kernel must reproduce)
    assert int(rng.fmix32(np.uint32(1))) == 0x514E28B7
    assert int(rng.dropout_layer_key(1024, 0, 0)) == int(rng.fmix32(
        rng.fmix32(np.uint32(1024 + 0x9E3779B9)) ^ np.uint32(0x85EBCA6B)))


def test_shuffle_perm_semantics():
    for n, buf in ((0, 10), (1, 10), (17, 100), (1000, 64), (257, 257)):
        perm = rng.shuffle_perm(n, buf, seed=42)
        assert sorted(perm.tolist()) == list(range(n))
        if n > 1:
            assert not np.array_equal(perm, np.arange(n))
            # an element can be emitted at most `buf-1`... positions early: out[i] <= i + buf - 1
            assert np.all(perm <= np.arange(n) + buf - 1)
    assert not np.array_equal(rng.shuffle_perm(100, 100, 1), rng.shuffle_perm(100, 100, 2))


def test_uncertainty_weighted_loss_gradients():
    """weighted_loss.py:30-43: loss = mean(BCE) / var^2 + log var (+ regularisers), var = log_var[domain]:
    every gradient scales by 1 / var^2 and d loss / d var = -2 mean(BCE) / var^3 + 1 / var (finite differences)."""
    rs = np.random.RandomState(4)
    p = tower.init_params(rs, 30, 20, 3, emb_dim=8, hidden=(16, 8, 4))
    p["log_var"] = np.array([1.3, 0.7, 1.0], F32)
    B = 12
    uid = rs.randint(0, 30, B).astype(np.int32)
    pid = rs.randint(0, 20, B).astype(np.int32)
    dom = np.full(B, 1, np.int32)
    y = (rs.uniform(size=B) < 0.4).astype(F32)
    masks = [np.ones((B, h), F32) for h in (16, 8, 4)]
    loss_u, g_u, _ = tower.loss_and_grads(p, uid, pid, dom, y, masks, 0.0, False, None, False, True)
    loss_p, g_p, _ = tower.loss_and_grads(p, uid, pid, dom, y, masks, 0.0, False, None, False, False)
    reg = tower.reg_loss(p)
    var = 0.7
    np.testing.assert_allclose(float(loss_u), (float(loss_p) - float(reg)) / var ** 2 + np.log(var) + float(reg), rtol=1e-5)
    two_l2 = 2e-5
    for n in ("W0", "W2", "b1", "wo", "gb"):
        np.testing.assert_allclose(g_u[n], g_p[n] / var ** 2, rtol=1e-4, atol=1e-7)
    # the domain table carries its regulariser unweighted
    np.testing.assert_allclose(g_u["domain_emb"] - two_l2 * p["domain_emb"],
                               (g_p["domain_emb"] - two_l2 * p["domain_emb"]) / var ** 2, rtol=1e-4, atol=1e-9)
    mean_bce = float(loss_p) - float(reg)
    want = np.zeros(3)
    want[1] = -2 * mean_bce / var ** 3 + 1 / var
    np.testing.assert_allclose(g_u["log_var"], want, rtol=1e-5, atol=1e-7)
    assert tower.param_names(False, False, True)[-1] == "log_var"


def test_rowgrad_adam_is_bitwise_the_dense_formula(monkeypatch):
    """oracle/bigtable.py (the table-sized tensors' Adam / SGD in row blocks on a thread pool, the gradient held as
    touched rows + regulariser coefficient) against the literal dense formulation of oracle/tower.py and
    oracle/star.py: same bits in every parameter and both Adam slots after Adam steps, an SGD step and a
    flatten() of the gradient (accumulate passes), with rows repeated inside a batch and rows no batch touches."""
    from oracle import bigtable, star

    def run(big):
        monkeypatch.setattr(bigtable, "MIN_ELEMENTS", 1 if big else 1 << 40)
        out = {}
        for kind in ("deepfm", "star"):
            rs = np.random.RandomState(11)
            n_user, n_item, D, B = 5000, 3000, 3, 96
            if kind == "star":
                p = star.init_params(rs, n_user, n_item, D)
                model = star.OracleStar(p, emb_trainable=True, lr=1e-3)
            else:
                p = tower.init_params(rs, n_user, n_item, D, pretrained=False)
                p["user_emb"] *= 300
                p["item_emb"] *= 300
                model = tower.OracleModel(p, emb_trainable=True, dropout=0.5, lr=1e-3, tower="deepfm")
            for step in range(6):
                uid = rs.randint(0, n_user, B).astype(np.int32)
                uid[:8] = uid[8:16]
                pid = rs.randint(0, 50, B).astype(np.int32)             # heavy repeats
                dom = np.full(B, step % D, np.int32)
                y = (rs.rand(B) < 0.4).astype(F32)
                model.use_sgd = step == 4
                model.train_on_batch(uid, pid, dom, y)
            out[kind] = [model.params[n].copy() for n in model.names] + [model.opt.m[n].copy() for n in model.names] + \
                        [model.opt.v[n].copy() for n in model.names]
            if kind == "deepfm":
                acc = np.zeros(model.get_flat().size, F32)
                model.accumulate_on_batch(acc, uid, pid, dom, y)
                out[kind].append(acc)
                out[kind].append(np.array([model.evaluate({"uid": uid, "pid": pid, "domain": dom, "label": y}, 64)[0]]))
        return out

    dense, rows = run(False), run(True)
    for kind in dense:
        for a, b in zip(dense[kind], rows[kind]):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), kind
    assert np.abs(dense["deepfm"][0]).max() > 0
