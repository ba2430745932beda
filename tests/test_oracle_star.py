"""Oracle self-checks for the Star tower (PartitionedNorm + StarFCN): float64 finite differences,
moving-statistics rule, eval path."""
import numpy as np

from oracle import star as S
from oracle import tower as T

F32 = np.float32


def loss64(q, uid, pid, dom, y):
    d = int(dom[0])
    x = np.concatenate([q["user_emb"][uid], q["item_emb"][pid], q["domain_emb"][dom]], axis=1)
    mean = x.mean(axis=0)
    var = ((x - mean) ** 2).mean(axis=0)
    gamma = q["pn_gamma_shared"] * q["pn_gamma_spec"][d]
    beta = q["pn_beta_shared"] + q["pn_beta_spec"][d]
    h = (x - mean) / np.sqrt(var + 1e-3) * gamma + beta
    for l in range(3):
        h = np.maximum(h @ (q["Ws%d" % l] * q["Wd%d" % l][d]) + q["bs%d" % l] + q["bd%d" % l][d], 0.0)
    z = (h @ q["wo"])[:, 0] + q["gb"][0]
    p = 1.0 / (1.0 + np.exp(-z))
    return float(np.mean(-(y * np.log(p) + (1 - y) * np.log(1 - p))))


def small_problem(seed=3, n_user=40, n_item=30, n_domain=3, B=24):
    rs = np.random.RandomState(seed)
    p = S.init_params(rs, n_user, n_item, n_domain)
    # move every tensor off its special initial value so that all chain-rule factors are exercised
    for n in ("pn_gamma_shared", "pn_gamma_spec"):
        p[n] = (p[n] + rs.standard_normal(p[n].shape) * 0.2).astype(F32)
    for n in ("pn_beta_shared", "pn_beta_spec", "bs0", "bs1", "bs2", "bd0", "bd1", "bd2", "gb"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.1).astype(F32)
    for l in range(3):
        p["Wd%d" % l] = (p["Wd%d" % l] * 8).astype(F32)      # effective kernels of useful size
    p["user_emb"] *= 4
    p["item_emb"] *= 4
    uid = rs.randint(0, n_user, B).astype(np.int32)
    uid[:4] = uid[4:8]                                        # repeated rows
    pid = rs.randint(0, n_item, B).astype(np.int32)
    dom = np.full(B, 1, np.int32)
    y = (rs.uniform(size=B) < 0.4).astype(F32)
    return p, uid, pid, dom, y


def test_star_gradients_match_finite_differences():
    p, uid, pid, dom, y = small_problem()
    state = S.init_state(3)
    loss, g, _, _ = S.loss_and_grads(p, state, uid, pid, dom, y, True)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    assert abs(float(loss) - loss64(q, uid, pid, dom, y)) < 1e-5
    rs = np.random.RandomState(0)
    meta, rest = S.param_names(True)
    assert set(meta + rest) == set(g)
    h = 1e-6
    for name in meta + rest:
        a = q[name]
        picks = [tuple(rs.randint(0, s) for s in a.shape) for _ in range(6)]
        if name in ("user_emb",):
            picks += [(int(uid[0]), 5), (int(uid[5]), 77)]
        if a.ndim >= 2 and name.startswith(("Wd", "bd", "pn_gamma_spec", "pn_beta_spec", "domain_emb")):
            picks += [(1,) + tuple(rs.randint(0, s) for s in a.shape[1:]) for _ in range(4)]   # the live domain
        for idx in picks:
            old = a[idx]
            a[idx] = old + h
            lp = loss64(q, uid, pid, dom, y)
            a[idx] = old - h
            lm = loss64(q, uid, pid, dom, y)
            a[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - g[name][idx]) < 2e-5 + 2e-3 * abs(fd), (name, idx, fd, g[name][idx])
    # tensors of the other domains get exactly zero gradient
    for name in ("Wd0", "bd1", "pn_gamma_spec", "pn_beta_spec", "domain_emb"):
        assert not g[name][0].any() and not g[name][2].any() and g[name][1].any() or name == "domain_emb"
    # the domain embedding is constant over a single-domain batch: PN removes it, its gradient is rounding noise
    assert np.abs(g["domain_emb"]).max() < 1e-6


def test_moving_statistics_zero_debias_and_eval():
    p, uid, pid, dom, y = small_problem(seed=5)
    m = S.OracleStar({k: v.copy() for k, v in p.items()}, emb_trainable=True, lr=1e-3)
    x = T.gather(m.params, uid, pid, dom)
    mean, var = S.batch_moments(x)
    m.train_on_batch(uid, pid, dom, y)
    # first zero-debiased update returns exactly the batch statistic (0.01 * v / (1 - 0.99))
    np.testing.assert_allclose(m.state["mov_mean"][1], mean, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(m.state["mov_var"][1], var, rtol=1e-5, atol=1e-9)
    assert m.state["steps"].tolist() == [0.0, 1.0, 0.0]
    assert (m.state["mov_var"][0] == 1).all() and (m.state["mov_mean"][2] == 0).all()
    # second update: biased = 0.99 * 0.01 * v1 + 0.01 * v2, debiased by 1 - 0.99^2
    x2 = T.gather(m.params, uid[::-1], pid, dom)
    mean2, _ = S.batch_moments(x2)
    m.train_on_batch(uid[::-1].copy(), pid, dom, y)
    want = (0.99 * 0.01 * mean.astype(np.float64) + 0.01 * mean2) / (1 - 0.99 ** 2)
    np.testing.assert_allclose(m.state["mov_mean"][1], want, rtol=1e-4, atol=1e-6)
    # eval uses domain 1's moving statistics, not the batch's
    data = {"uid": uid, "pid": pid, "domain": dom, "label": y}
    loss, preds = m.evaluate(data, 16)
    p_tr, _ = S.forward(m.params, m.state, uid, pid, dom, True)
    assert np.isfinite(loss) and np.abs(preds - p_tr).max() > 1e-6
    # one Adam step moved every trainable tensor of the live domain and, through the decayed moments,
    # nothing of the other domains yet (their m, v are still zero)
    assert not np.array_equal(m.params["Wd0"][1], p["Wd0"][1]) and np.array_equal(m.params["Wd0"][0], p["Wd0"][0])
