"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for index paths (embedding gather, AUC
binning) and for the outer updates (vs goldens produced by the reference's own
numpy); fp32 contractions within the tolerances written below; per-domain
AUC within 1e-3.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import auc as oauc          # noqa: E402
from oracle import loops as oloops      # noqa: E402
from oracle import outer as oouter      # noqa: E402
from oracle import rng as orng          # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import engine, synthetic
    return engine, synthetic


def make_problem(env, scale=0.05, batch=256, dropout=0.5, seed=7, shape="taobao10", emb_trainable=False,
                 tower="mlp", uncertainty=False):
    engine, synthetic = env
    g = synthetic.generate(shape, batch_size=batch, seed=seed, scale=scale)
    rs = np.random.RandomState(seed)
    params = otower.init_params(rs, g["n_user"], g["n_item"], g["n_domain"])
    params["user_emb"] = g["tables"]["user_emb"].copy()
    params["item_emb"] = g["tables"]["item_emb"].copy()
    params["domain_emb"] = (rs.standard_normal(params["domain_emb"].shape) * 0.05).astype(F32)
    for l in range(3):
        params["b%d" % l] = (rs.standard_normal(params["b%d" % l].shape) * 0.05).astype(F32)
    if tower in ("deepfm", "wdl"):      # non-zero linear tables so that every term of the logit is exercised
        params["lin_domain"] = (rs.standard_normal(g["n_domain"]) * 0.05).astype(F32)
        if emb_trainable:
            params["lin_user"] = (rs.standard_normal(g["n_user"]) * 0.05).astype(F32)
            params["lin_item"] = (rs.standard_normal(g["n_item"]) * 0.05).astype(F32)
    if uncertainty:            # distinct per-domain scales around the initial value 1
        params["log_var"] = (1.0 + rs.uniform(-0.3, 0.3, g["n_domain"])).astype(F32)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], batch, dropout=dropout,
                             emb_trainable=emb_trainable, tower=tower, uncertainty_weight=uncertainty)
    if not emb_trainable:
        eng.bind_table("user_emb", params["user_emb"])
        eng.bind_table("item_emb", params["item_emb"])
    for split in ("train", "val", "test"):
        for d in range(g["n_domain"]):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    eng.set_weights(eng.pack(params))
    model = otower.OracleModel({k: v.copy() for k, v in params.items()}, emb_trainable=emb_trainable, dropout=dropout,
                               lr=1e-3, dropout_seed=eng.dropout_seed, tower=tower, uncertainty=uncertainty)
    return g, eng, model


def assert_adam_close(got, want, n_steps, lr, name, max_frac=1e-3):
    """k Adam steps of two fp32 evaluations.  Adam normalises every update to ~lr, so an element whose
    gradient is within rounding of zero -- or a hidden unit whose pre-activation sits within rounding of
    the relu kink (a handful per batch once the weights differ by 1e-5) -- can legitimately move by up to
    lr per step in different directions (a flipped unit drags the small-gradient elements of its whole
    weight column along).  Bar: all but 1e-3 of the elements within 5 % of k*lr (2e-3 where the callers say so: the
    worst fraction any test of this file has shown is 3.4e-4, MAMDR_TEST_REPORT_FRAC=1 prints them), none
    beyond the 2*k*lr Adam can produce, and the typical element far tighter."""
    diff = np.abs(np.asarray(got, F32).ravel() - np.asarray(want, F32).ravel())
    bound = 0.05 * n_steps * lr
    frac = float(np.mean(diff > bound))
    if os.environ.get("MAMDR_TEST_REPORT_FRAC"):
        print("ADAMFRAC %s n=%d frac=%.3e max_frac=%.0e maxdiff/klr=%.3f med/klr=%.5f" % (name, diff.size, frac, max_frac, diff.max() / (n_steps * lr), np.median(diff) / (n_steps * lr)))
    assert frac <= max_frac, (name, "fraction beyond %.1e: %.2e" % (bound, frac), float(diff.max()))
    assert diff.max() <= 2.02 * n_steps * lr, (name, float(diff.max()))
    assert float(np.median(diff)) < 0.002 * n_steps * lr, (name, float(np.median(diff)))


def same_bits(a, b):
    return np.array_equal(np.asarray(a, F32).view(np.uint32), np.asarray(b, F32).view(np.uint32))


# ------------------------------------------------------------------ K1 gather: bit-exact
def test_gather_bit_exact(env):
    g, eng, model = make_problem(env)
    for d in (0, 3):
        cols = g["data"]["train"][d]
        n = cols["uid"].shape[0]
        perm = orng.shuffle_perm(n, 10000, seed=5 + d)
        out = eng.gather(d, "train", perm=torch.from_numpy(perm).to(eng.device)).cpu().numpy()
        want = otower.gather(model.params, cols["uid"][perm], cols["pid"][perm], cols["domain"][perm])
        assert same_bits(out, want)
        # ragged window, file order
        out = eng.gather(d, "train", first_row=3, n_rows=21).cpu().numpy()
        want = otower.gather(model.params, cols["uid"][3:24], cols["pid"][3:24], cols["domain"][3:24])
        assert same_bits(out, want)
    eng.close()


def test_shuffle_perm_matches_oracle(env):
    engine, _ = env
    for n, buf, seed in ((1, 10, 1), (100, 10, 2), (1000, 10000, 3), (5000, 128, 2 ** 63 + 5)):
        assert np.array_equal(engine.shuffle_perm(n, buf, seed), orng.shuffle_perm(n, buf, seed))


# ------------------------------------------------------------------ K7 outer updates: bit-exact
def _flat(G, prefix, pad_to=None):
    v = np.concatenate([G["%s_%d" % (prefix, i)].ravel() for i in range(int(G["n_tensors"]))])
    return v


def test_outer_updates_bit_exact_vs_reference_goldens(env, golden_dir):
    engine, _ = env
    G = np.load(os.path.join(golden_dir, "outer_goldens.npz"))
    g, eng, _ = make_problem(env, scale=0.02)
    dev = eng.device

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a, F32)).to(dev)

    theta, new, new2, phi = (_flat(G, k) for k in ("theta", "new", "new2", "phi"))
    n = theta.size                     # 2753: not a multiple of 4 -> exercises the scalar tail
    assert n % 4 != 0
    for lr_name, lr in (("lr0p1", 0.1), ("lr1", 1.0), ("lr0p5", 0.5)):
        t = T(theta)
        eng.interp(t, T(new), t, lr)                      # DN / Reptile / MAMDR-DN
        assert same_bits(t.cpu().numpy(), _flat(G, "dn_" + lr_name))
        assert same_bits(t.cpu().numpy(), _flat(G, "mamdr_dn_" + lr_name))
        for method in ("plus", "times"):
            merged = torch.empty(n, device=dev)
            eng.merge(merged, T(theta), T(phi), method)
            assert same_bits(merged.cpu().numpy(), _flat(G, "merged_" + method))
            p = T(phi)
            eng.interp(p, T(new), merged, lr)             # MAMDR DR: phi += (new - merged) * lr
            assert same_bits(p.cpu().numpy(), _flat(G, "mamdr_dr_%s_%s" % (method, lr_name)))
    # Reptile batch variant
    acc = torch.zeros(n, device=dev)
    t = T(theta)
    eng.accumulate(acc, T(new), t)
    eng.accumulate(acc, T(new2), t)
    assert same_bits(acc.cpu().numpy(), _flat(G, "reptile_acc"))
    eng.apply_accumulated(t, acc, 0.0, 0.1)
    assert same_bits(t.cpu().numpy(), _flat(G, "reptile_batch"))
    assert not acc.cpu().numpy().any()
    # MAMDR batch variant + phi = new - merged
    for method in ("plus", "times"):
        merged = torch.empty(n, device=dev)
        eng.merge(merged, T(theta), T(phi), method)
        acc = torch.zeros(n, device=dev)
        shared = T(theta) if method == "times" else None
        eng.accumulate(acc, T(new), merged, shared, 1.0)
        eng.accumulate(acc, T(new2), merged, shared, 1.0)
        assert same_bits(acc.cpu().numpy(), _flat(G, "mamdr_acc_" + method))
        p = T(phi)
        eng.apply_accumulated(p, acc, 5.0, 0.1)
        assert same_bits(p.cpu().numpy(), _flat(G, "mamdr_batch_" + method))
        dw = torch.empty(n, device=dev)
        eng.sub(dw, T(new), merged)
        assert same_bits(dw.cpu().numpy(), _flat(G, "mamdr_domain_weights_" + method))
    eng.close()


@pytest.mark.parametrize("method", ["plus", "times"])
def test_dr_advance_is_bit_identical_to_interp_merge_assign(env, method):
    """mamdr_dr_advance = one DR support step in a single pass (phi += (live - merged) gamma; merged = theta (+|*) phi;
    model := merged): the same bits as mamdr_interp + mamdr_merge + mamdr_copy, on the meta prefix of the live
    vector (the Star tower's theta is a prefix), with and without the model assignment."""
    g, eng, model = make_problem(env, scale=0.05, batch=256)
    n = eng.n_meta - 3                                  # not a multiple of 4: the scalar tail runs too
    rs = np.random.RandomState(9)
    vec = lambda: torch.from_numpy((rs.standard_normal(n) * 0.1).astype(F32)).to(eng.device)
    live0 = (rs.standard_normal(eng.n_params) * 0.1).astype(F32)
    for assign in (True, False):
        theta, phi0, merged0 = vec(), vec(), vec()
        eng.set_weights(torch.from_numpy(live0).to(eng.device))
        phi_a, merged_a = phi0.clone(), merged0.clone()
        eng.interp(phi_a, eng.weights[:n], merged_a, 0.37)
        eng.merge(merged_a, theta, phi_a, method)
        want_live = live0.copy()
        if assign:
            want_live[:n] = merged_a.cpu().numpy()
        phi_b, merged_b = phi0.clone(), merged0.clone()
        eng.set_weights(torch.from_numpy(live0).to(eng.device))
        eng.dr_advance(phi_b, merged_b, theta, 0.37, method, assign_model=assign)
        assert same_bits(phi_b.cpu().numpy(), phi_a.cpu().numpy())
        assert same_bits(merged_b.cpu().numpy(), merged_a.cpu().numpy())
        assert same_bits(eng.get_weights().cpu().numpy(), want_live)
        assert float((phi_b - phi0).abs().max()) > 0
    eng.close()


def test_outer_updates_random_vs_oracle_large(env):
    """full flat-vector size, random data, vs oracle/outer.py (itself pinned to the goldens)."""
    g, eng, _ = make_problem(env, scale=0.02)
    rs = np.random.RandomState(3)
    n = eng.n_params
    a, b, c = ((rs.standard_normal(n) * 0.1).astype(F32) for _ in range(3))
    ta, tb, tc = (torch.from_numpy(x).to(eng.device) for x in (a, b, c))
    t = ta.clone()
    eng.interp(t, tb, tc, 0.1)
    assert same_bits(t.cpu().numpy(), oouter.mamdr_update(a.copy(), b, c, 0.1))
    # scale 0 is the identity, bit-for-bit
    t = ta.clone()
    eng.interp(t, tb, tc, 0.0)
    assert same_bits(t.cpu().numpy(), a)
    eng.close()


# ------------------------------------------------------------------ inner step vs oracle
def _grad_via_sgd(eng, d, perm_t, step, bs):
    """SGD with lr=1 turns the update into p_old - p_new = gradient."""
    before = eng.get_weights().cpu().numpy()
    eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", batch_size=bs)
    after = eng.get_weights().cpu().numpy()
    return before - after


@pytest.mark.parametrize("dropout", [0.5, 0.0])
def test_one_step_gradients_match_oracle(env, dropout):
    g, eng, model = make_problem(env, batch=256, dropout=dropout)
    d = 5
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=11)
    perm_t = torch.from_numpy(perm).to(eng.device)
    bs = 256
    n_step = -(-n // bs)
    for step in (0, n_step - 1):          # a full batch and the final (partial) batch
        idx = perm[step * bs:(step + 1) * bs]
        masks = otower.train_masks(model.seed, model.step, len(idx), model.hidden, dropout) if dropout > 0 else None
        if masks is None:
            masks = [np.ones((len(idx), h), F32) for h in model.hidden]
        loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx],
                                               cols["domain"][idx], cols["label"][idx], masks, dropout, False)
        want = eng.pack({**{k: np.zeros_like(v) for k, v in model.params.items()}, **grads}).cpu().numpy()
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", batch_size=bs,
                        loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)               # undo the lr=1 step; dropout counter advanced by 1
        model.step += 1
        # tolerance: fp32 contractions with different summation order; gradients are O(1e-3)
        scale = np.abs(want).max()
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6 * max(scale, 1e-3))
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    eng.close()


def test_adam_pass_matches_oracle(env):
    """a whole shuffled pass (incl. the partial batch) with Adam + dropout."""
    g, eng, model = make_problem(env, batch=256, dropout=0.5)
    d = 9
    cols = g["data"]["train"][d]
    perm = orng.shuffle_perm(cols["uid"].shape[0], 10000, seed=3)
    n_steps = min(6, -(-perm.shape[0] // 256))
    losses_t = torch.zeros(n_steps, device=eng.device)
    eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), n_steps=n_steps, lr=1e-3, loss_out=losses_t)
    want_losses = model.train_pass(cols, perm, 256, max_steps=n_steps)
    got = eng.unpack(eng.get_weights())
    np.testing.assert_allclose(losses_t.cpu().numpy(), np.array(want_losses, F32), rtol=2e-5, atol=2e-6)
    # Adam normalises the step to ~lr per element, so compare in units of lr: after k steps
    # two fp32 evaluations stay within a small fraction of k*lr of each other.
    for name in model.names:
        diff = np.abs(got[name].reshape(model.params[name].shape) - model.params[name]).max()
        assert diff < 0.05 * n_steps * 1e-3, (name, diff)
    assert int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == n_steps == model.opt.t
    eng.close()


@pytest.mark.parametrize("batch", [1000, 250])
def test_adam_pass_with_a_batch_size_that_is_no_multiple_of_16(env, batch):
    """Every step's 16-row padding is counted from its own first row (s * batch), so with batch % 16 != 0 the
    tiles of consecutive steps overlap in the pre-gathered pass buffer and the last step's tile ends past the
    call's rows (k_pass_prep zero-fills 16 rows behind them).  Whole shuffled pass on the k_wgrad_adam path
    (mlp tower, frozen tables) incl. a short last batch, against the oracle."""
    g, eng, model = make_problem(env, scale=0.25 if batch == 1000 else 0.05, batch=batch, dropout=0.5)
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 1
    sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(10)]
    # a domain whose last batch is short and not a multiple of 16 either
    d = max((k for k in range(10) if sizes[k] % batch and (sizes[k] % batch) % 16), key=lambda k: sizes[k])
    cols = g["data"]["train"][d]
    perm = orng.shuffle_perm(sizes[d], 10000, seed=8)
    n_steps = -(-sizes[d] // batch)
    first = max(0, n_steps - 5)
    model.step = 0
    eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), first_step=first, n_steps=n_steps - first, lr=1e-3,
                    batch_size=batch)
    for s_ in range(first, n_steps):
        ii = perm[s_ * batch:(s_ + 1) * batch]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert np.isfinite(got[name]).all(), name
        assert_adam_close(got[name], model.params[name], n_steps - first, 1e-3, name, max_frac=2e-3)
    eng.close()


# ------------------------------------------------------------------ trainable user / item tables
def test_trainable_tables_gradients_and_adam(env):
    """emb_trainable: scatter-add of row gradients + dense regulariser on EVERY row, dense Adam
    over the whole table (TF1 semantics); rows repeated inside a batch included."""
    g, eng, model = make_problem(env, scale=0.1, batch=256, dropout=0.5, emb_trainable=True)
    assert "user_emb" in eng.segments and eng.n_params == 128 * (g["n_user"] + g["n_item"]) + 139777 + 128 * 10 + 3
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=4)
    perm_t = torch.from_numpy(perm).to(eng.device)
    idx = perm[:256]
    assert len(np.unique(cols["uid"][idx])) < 256          # repeated rows exercise the segment sum
    masks = otower.train_masks(model.seed, model.step, 256, model.hidden, 0.5)
    loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                           cols["label"][idx], masks, 0.5, True)
    want = eng.pack(grads).cpu().numpy()
    w0 = eng.get_weights()
    loss_t = torch.zeros(1, device=eng.device)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
    got = (w0 - eng.get_weights()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6 * max(np.abs(want).max(), 1e-3))
    assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    # untouched rows still move by the regulariser: g = 2 * l2 * w
    untouched = np.setdiff1d(np.arange(g["n_user"]), cols["uid"][idx])[:5]
    off = eng.segments["user_emb"][0]
    for r in untouched:
        # (p_old - p_new recovers g only to ~ulp(p) = 1e-8 for p ~ 0.1)
        np.testing.assert_allclose(got[off + r * 128: off + (r + 1) * 128], 2e-5 * model.params["user_emb"][r],
                                   rtol=1e-3, atol=2e-8)
    # a few Adam steps, including the partial last batch
    eng.set_weights(w0)
    model.step = 1
    n_steps = -(-n // 256)
    first = max(0, n_steps - 3)
    eng.train_steps(d, perm=perm_t, first_step=first, n_steps=n_steps - first, lr=1e-3)
    for s_ in range(first, n_steps):
        ii = perm[s_ * 256:(s_ + 1) * 256]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        diff = np.abs(got[name].reshape(model.params[name].shape) - model.params[name]).max()
        assert diff < 0.05 * 3 * 1e-3, (name, diff)
    # eval uses the live tables (regulariser term of the loss recomputed)
    loss_g, auc_g = eng.evaluate(d, "val")
    loss_o, preds = model.evaluate(g["data"]["val"][d], 256)
    assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    eng.close()


@pytest.mark.parametrize("batch", [256, 1024, 4096])
def test_mixed_domain_ids_in_one_batch(env, batch):
    """The domain-table gradient and dW0[256:384] are rebuilt from S = onehot(domain)^T dz1 by linearity (k_wgrad
    carries no tiles for those rows of W0): exact for ANY mix of domain ids inside a batch, not only for the
    one-domain batches a pass normally produces.  A split whose domain column cycles through several ids,
    one SGD step at lr 1: every gradient against the oracle, at one and at several row groups per tile, with
    the 4-row and the 16-row tower."""
    g, eng, model = make_problem(env, scale=0.3 if batch > 1024 else 0.1, batch=batch, dropout=0.5)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = {k: v.copy() for k, v in g["data"]["train"][d].items()}
    n = cols["uid"].shape[0]
    assert n >= batch
    rs = np.random.RandomState(3)
    cols["domain"][:] = rs.choice([1, 4, 7, 9], size=n).astype(cols["domain"].dtype)
    eng.bind_domain_data(d, "train", cols["uid"], cols["pid"], cols["domain"], cols["label"])
    perm = orng.shuffle_perm(n, 10000, seed=8)
    perm_t = torch.from_numpy(perm).to(eng.device)
    idx = perm[:batch]
    assert len(np.unique(cols["domain"][idx])) == 4
    masks = otower.train_masks(model.seed, model.step, batch, model.hidden, 0.5)
    loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                           cols["label"][idx], masks, 0.5, False)
    want = eng.pack(grads).cpu().numpy()
    w0 = eng.get_weights()
    loss_t = torch.zeros(1, device=eng.device)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
    got = (w0 - eng.get_weights()).cpu().numpy()
    for name, (off, cnt) in eng.segments.items():
        w = want[off:off + cnt]
        np.testing.assert_allclose(got[off:off + cnt], w, rtol=2e-4, atol=max(2e-6 * max(np.abs(w).max(), 1e-3), 1e-8),
                                   err_msg=name)
    assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    off = eng.segments["domain_emb"][0]
    mag = np.abs(want[off:off + 1280].reshape(10, 128)).max(axis=1)
    assert mag[[1, 4, 7, 9]].min() > 100 * mag[[0, 2, 3, 5, 6, 8]].max()   # the other rows see the regulariser only
    eng.close()


def _mlp_state(eng):
    return (eng.weights.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy())


def _fused_workout(env, batch, mixed):
    """a call pattern that exercises the pending domain-table step of the k_wgrad_adam path: whole passes (the final
    batch partial), single-step calls, a call starting in the middle of a pass, SGD and accumulate steps, a weight
    assignment in between -- on one-domain batches, or on a split whose domain column mixes four ids."""
    g, eng, model = make_problem(env, scale=0.3 if batch > 1024 else 0.1, batch=batch, dropout=0.5)
    sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(10)]
    order = sorted(range(10), key=lambda k: -sizes[k])[:3]
    if mixed:
        d = order[0]
        cols = {k: v.copy() for k, v in g["data"]["train"][d].items()}
        cols["domain"][:] = np.random.RandomState(3).choice([1, 4, 7, 9], size=sizes[d]).astype(cols["domain"].dtype)
        eng.bind_domain_data(d, "train", cols["uid"], cols["pid"], cols["domain"], cols["label"])
    acc = eng.new_vector()
    eng.bind_accumulator(acc)
    for rep_ in range(2):
        for d in order:
            perm = torch.from_numpy(orng.shuffle_perm(sizes[d], 10000, seed=41 + d + rep_)).to(eng.device)
            nb = -(-sizes[d] // batch)
            eng.train_steps(d, perm=perm, lr=1e-3)
            eng.train_steps(d, perm=perm, first_step=nb - 1, n_steps=1, lr=1e-3)
            if nb > 2:
                eng.train_steps(d, perm=perm, first_step=1, n_steps=2, lr=1e-3, optimizer="sgd")
            eng.train_steps(d, perm=perm, first_step=0, n_steps=min(2, -(-sizes[d] // batch)), lr=1e-3, optimizer="accumulate")
        if rep_ == 0:
            snap = eng.get_weights()
            eng.set_weights((snap + eng.get_weights()) * 0.5 + 1e-3)
    out = _mlp_state(eng) + (acc.cpu().numpy().copy(),)
    eng.close()
    return out


@pytest.mark.parametrize("batch,mixed", [(256, False), (1024, False), (1024, True), (4096, False)])
def test_pending_domain_table_step_is_bitwise(env, batch, mixed):
    """k_wgrad_adam path: the domain table's optimiser step is applied by the NEXT step's tower kernel (every
    workgroup recomputes the row its batch uses; waves 6, 7 of workgroup d write row d back) and by k_dm_finish for
    the last step of a call.  MAMDR_DM_EACH=1 materialises it after every step instead.  Same arithmetic at
    every site (dm_pair / dm_elem_finish / dm_step4) -> weights, both Adam slots and a MAML accumulator agree
    BITWISE, on one-domain batches (the fast path) and on mixed-domain batches (every lane on its own)."""
    res = {}
    for mode in ("pending", "each"):
        os.environ["MAMDR_FUSED"] = "2"
        os.environ["MAMDR_DM_EACH"] = "1" if mode == "each" else "0"
        try:
            res[mode] = _fused_workout(env, batch, mixed)
        finally:
            os.environ.pop("MAMDR_FUSED", None)
            os.environ.pop("MAMDR_DM_EACH", None)
    for a, b, name in zip(res["pending"], res["each"], ("weights", "adam_m", "adam_v", "accumulator")):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert same_bits(a, b), (name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))


@pytest.mark.parametrize("batch", [256, 1000])
def test_per_call_duties_across_calls_are_bitwise(env, batch):
    """Round 4, k_wgrad_adam path: (1) an Adam call leaves its last domain-table step PENDING for the next call's first
    tower (materialised by mamdr_sync_tables / mamdr_dr_advance_live / any other kind of call); (2) the call's first
    tower reads W2 itself when the transposed copies are stale (no k_transpose_w); (3) the passes of a whole window of
    calls are gathered in one launch (meta.PassWindow -> mamdr_pregather_passes: plan.EpochShuffles can show its next
    permutations); (4) a DR support step materialises the pending step inside mamdr_dr_advance_live's launch.
    Against the round-3 behaviour (MAMDR_DM_CALL=1: k_dm_finish closes every call; MAMDR_NO_W2_DIRECT=1: k_transpose_w
    opens it; a perm_fn that cannot be peeked: every call gathers its own pass): two MAMDR epochs + SGD / accumulate /
    one-step calls / an evaluation in between -> theta, every phi, the live weights, both Adam slots and a MAML
    accumulator agree BITWISE."""
    from mamdr_amd import meta, plan as mplan
    res = {}
    for mode in ("new", "old"):
        if mode == "old":
            os.environ["MAMDR_DM_CALL"] = "1"
            os.environ["MAMDR_NO_W2_DIRECT"] = "1"
        try:
            g, eng, model = make_problem(env, scale=0.25, batch=batch, dropout=0.5)
        finally:
            os.environ.pop("MAMDR_DM_CALL", None)
            os.environ.pop("MAMDR_NO_W2_DIRECT", None)
        assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 1
        D = g["n_domain"]
        sizes = [eng.n_rows(d, "train") for d in range(D)]
        theta = eng.get_weights()
        rs = np.random.RandomState(3)
        phis = {d: torch.from_numpy((rs.standard_normal(eng.n_params) * 1e-3).astype(F32)).to(eng.device) for d in range(D)}
        es = mplan.EpochShuffles(mplan.PassShuffler(sizes, 10000, 77), eng.device)
        planner = mplan.EpochPlanner(range(D), 3, True, True, seed=5)
        acc = eng.new_vector()
        eng.bind_accumulator(acc)
        for ep in range(2):
            plan = planner.next_epoch()
            es.prepare(mplan.epoch_passes(plan))
            perm_fn = es if mode == "new" else (lambda d, es=es: es(d))
            meta.mamdr_epoch(eng, theta, phis, plan, perm_fn, batch, lr=1e-3, meta_lr=0.1)
            if ep == 0:
                # other kinds of calls find the table materialised: an Adam call, then SGD, accumulate, a one-step Adam
                # call, an evaluation, and Adam again
                big = sorted(range(D), key=lambda d: -sizes[d])[:2]          # (at least two batches each)
                assert sizes[big[1]] > batch
                pm = torch.from_numpy(orng.shuffle_perm(sizes[big[0]], 10000, seed=9)).to(eng.device)
                eng.train_steps(big[0], perm=pm, lr=1e-3)
                eng.train_steps(big[0], perm=pm, first_step=0, n_steps=1, lr=1e-2, optimizer="sgd")
                eng.train_steps(big[0], perm=pm, first_step=1, n_steps=1, lr=1e-3, optimizer="accumulate")
                eng.train_steps(big[1], first_step=0, n_steps=1, lr=1e-3)
                eng.evaluate(1, "val")
                eng.train_steps(big[1], first_step=1, n_steps=1, lr=1e-3)
        hits = int(eng.lib.mamdr_pregather_hits(eng.ctx))
        assert (hits > 20) if mode == "new" else (hits == 0), hits
        res[mode] = [theta.cpu().numpy().copy()] + [phis[d].cpu().numpy().copy() for d in range(D)] + \
                    [eng.get_weights().cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(),
                     eng.adam_v.cpu().numpy().copy(), acc.cpu().numpy().copy()]
        eng.close()
    for k, (a, b) in enumerate(zip(res["new"], res["old"])):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert same_bits(a, b), (k, int((a.view(np.uint32) != b.view(np.uint32)).sum()))


def test_tower4_w1_image_is_bitwise(env):
    """k_tower4 with W1 as an LDS image (grids of up to one tile per CU) against the variant that streams W1 / W1^T
    (MAMDR_T4_NO_W1L=1): the same MFMA sequences per output element and the same split-k sums -- identical bits, on the
    k_wgrad_adam path (pre-gathered passes) and on the slab path."""
    engine, synthetic = env
    out = {}
    for fused in ("1", "0"):
        # ("gather": the k_wgrad_adam path without pre-gathered passes = the PRE = false instance with the domain-table
        # duty; MAMDR_NO_PREGATHER promises bit-identical results)
        for mode in ("image", "stream") + (("gather",) if fused == "1" else ()):
            os.environ["MAMDR_T4_NO_W1L"] = "1" if mode == "stream" else "0"
            os.environ["MAMDR_NO_PREGATHER"] = "1" if mode == "gather" else "0"
            os.environ["MAMDR_FUSED"] = fused
            try:
                g, eng, model = make_problem(env, scale=0.05, batch=1024, dropout=0.5)
            finally:
                os.environ.pop("MAMDR_T4_NO_W1L", None)
                os.environ.pop("MAMDR_NO_PREGATHER", None)
                os.environ.pop("MAMDR_FUSED", None)
            d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
            n = g["data"]["train"][d]["uid"].shape[0]
            perm = torch.from_numpy(orng.shuffle_perm(n, 10000, seed=11)).to(eng.device)
            eng.train_steps(d, perm=perm, first_step=0, n_steps=min(4, -(-n // 1024)), lr=1e-3)
            out[(fused, mode)] = eng.get_weights().cpu().numpy().copy()
            eng.close()
        assert np.array_equal(out[(fused, "image")], out[(fused, "stream")])
        if fused == "1":
            assert np.array_equal(out[(fused, "image")], out[(fused, "gather")])


@pytest.mark.parametrize("batch", [256, 1024, 4096])
def test_fused_wgrad_adam_matches_the_slab_path(env, batch):
    """k_wgrad_adam (output-stationary tiles, optimiser step in the same launch, S workgroups for the domain-row
    terms) against k_wgrad -> slabs -> k_update on the same steps: one SGD step at lr 1 (= the gradient of every
    dense tensor) within fp32 summation-order noise, and the workout above within the multi-step Adam bar."""
    grads, runs = {}, {}
    for mode in ("fused", "slabs"):
        os.environ["MAMDR_FUSED"] = "2" if mode == "fused" else "0"
        try:
            g, eng, model = make_problem(env, scale=0.3 if batch > 1024 else 0.1, batch=batch, dropout=0.5)
            d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
            n = g["data"]["train"][d]["uid"].shape[0]
            perm_t = torch.from_numpy(orng.shuffle_perm(n, 10000, seed=8)).to(eng.device)
            w0 = eng.get_weights()
            eng.train_steps(d, perm=perm_t, first_step=0, n_steps=1, lr=1.0, optimizer="sgd")
            grads[mode] = (w0 - eng.get_weights()).cpu().numpy()
            eng.close()
            runs[mode] = _fused_workout(env, batch, False)
        finally:
            os.environ.pop("MAMDR_FUSED", None)
    ga, gb = grads["fused"], grads["slabs"]
    np.testing.assert_allclose(ga, gb, rtol=2e-4, atol=2e-6 * max(np.abs(gb).max(), 1e-3))
    n_adam = 2 * 3 * 12
    for a, b, name in zip(runs["fused"][:1], runs["slabs"][:1], ("weights",)):
        assert_adam_close(a, b, n_adam, 1e-3, name, max_frac=2e-3)


@pytest.mark.parametrize("tower", ["mlp", "deepfm"])
def test_trainable_tables_heavy_duplicates(env, tower):
    """A batch of 4096 in which one user occupies 3000 positions and five items share all of them: the row
    gradient of a repeated row is the sum over its positions in position order (np.add.at in the oracle) --
    lists long enough that the scanning wave has to drain its position list several times; with DeepFM the
    1-d linear tables sum dlogit over the same lists.  One SGD step at lr 1 exposes the gradients; then the
    lazy Adam path on the same batches against the oracle."""
    g, eng, model = make_problem(env, scale=0.3, batch=4096, dropout=0.5, emb_trainable=True, tower=tower)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = {k: v.copy() for k, v in g["data"]["train"][d].items()}
    n = cols["uid"].shape[0]
    assert n >= 2 * 4096
    rs = np.random.RandomState(5)
    hot = rs.choice(n, 3000 + 900, replace=False)
    cols["uid"][hot[:3000]] = cols["uid"][hot[0]]                      # 3000 positions, one user
    cols["uid"][hot[3000:]] = cols["uid"][hot[3000:3000 + 3]][rs.randint(0, 3, 900)]   # three users, ~300 each
    items = cols["pid"][rs.choice(n, 5, replace=False)]
    cols["pid"][:] = items[rs.randint(0, 5, n)]                        # five items share every position
    eng.bind_domain_data(d, "train", cols["uid"], cols["pid"], cols["domain"], cols["label"])
    perm = np.concatenate([hot[:3000], np.setdiff1d(np.arange(n), hot[:3000])]).astype(np.int32)   # hot user first
    perm[:4096] = perm[:4096][rs.permutation(4096)]
    perm_t = torch.from_numpy(perm).to(eng.device)
    idx = perm[:4096]
    assert (cols["uid"][idx] == cols["uid"][hot[0]]).sum() >= 3000
    masks = otower.train_masks(model.seed, model.step, 4096, model.hidden, 0.5)
    loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                           cols["label"][idx], masks, 0.5, True, None, model.deepfm)
    want = eng.pack({**{k: np.zeros_like(v) for k, v in model.params.items()}, **grads}).cpu().numpy()
    w0 = eng.get_weights()
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=1, lr=1.0, optimizer="sgd")
    got = (w0 - eng.get_weights()).cpu().numpy()
    eng.set_weights(w0)
    for name, (off, cnt) in eng.segments.items():
        w = want[off:off + cnt]
        floor = 6e-8 if name in ("user_emb", "item_emb") else 1e-8
        np.testing.assert_allclose(got[off:off + cnt], w, rtol=2e-4,
                                   atol=max(2e-6 * max(np.abs(w).max(), 1e-3), floor), err_msg=name)
    u_off = eng.segments["user_emb"][0] + int(cols["uid"][hot[0]]) * 128
    assert np.abs(want[u_off:u_off + 128]).max() > 0
    # lazy Adam over the same two batches (the hot rows are stepped by the reducing wave itself)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=2, lr=1e-3)
    model.step = 1                                     # the SGD step above consumed dropout step 0
    for s_ in range(2):
        ii = perm[s_ * 4096:(s_ + 1) * 4096]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert_adam_close(got[name], model.params[name], 2, 1e-3, name, max_frac=2e-3)
    eng.close()


# ------------------------------------------------------------------ DeepFM tower (BASELINE config 3)
@pytest.mark.parametrize("tower", ["deepfm", "wdl"])
@pytest.mark.parametrize("emb_trainable", [False, True])
def test_deepfm_gradients_adam_eval(env, emb_trainable, tower):
    """logit += linear tables + FM second-order term: gradients of every segment (incl. the FM part of
    the embedding gradients and the 1-d linear tables), a few Adam steps, eval.  `wdl` (deepctr.py:29-32) is
    the same tower without the FM term."""
    g, eng, model = make_problem(env, scale=0.1, batch=256, dropout=0.5, emb_trainable=emb_trainable, tower=tower)
    assert "lin_domain" in eng.segments and ("lin_user" in eng.segments) == emb_trainable
    assert [n for n in model.names if n not in eng.segments] == []
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=4)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_steps = -(-n // 256)
    # a few Adam steps, including the partial last batch
    first = max(0, n_steps - 3)
    eng.train_steps(d, perm=perm_t, first_step=first, n_steps=n_steps - first, lr=1e-3)
    for s_ in range(first, n_steps):
        ii = perm[s_ * 256:(s_ + 1) * 256]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert_adam_close(got[name], model.params[name], n_steps - first, 1e-3, name, max_frac=2e-3)
    eng.set_weights(eng.pack(model.params))      # re-synchronise before the tight gradient comparison
    for step in (0, n_steps - 1):          # a full batch and the final (partial) batch
        idx = perm[step * 256:(step + 1) * 256]
        masks = otower.train_masks(model.seed, model.step, len(idx), model.hidden, 0.5)
        loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                               cols["label"][idx], masks, 0.5, emb_trainable, None, model.deepfm)
        want = eng.pack({**{k: np.zeros_like(v) for k, v in model.params.items()}, **grads}).cpu().numpy()
        w0 = eng.get_weights()
        loss_t = torch.zeros(1, device=eng.device)
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)
        model.step += 1
        for name, (off, cnt) in eng.segments.items():
            w = want[off:off + cnt]
            # (p_old - p_new recovers g only to ~ulp(p): up to 6e-8 for table values up to 0.5)
            floor = 6e-8 if name in ("user_emb", "item_emb") else 1e-8      # W0 ~0.1: ulp 7e-9
            np.testing.assert_allclose(got[off:off + cnt], w, rtol=2e-4,
                                       atol=max(2e-6 * max(np.abs(w).max(), 1e-3), floor), err_msg=name)
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    assert np.abs(grads["lin_domain"]).max() > 0
    # eval: predictions carry the FM + linear terms
    loss_g, auc_g, hist, preds = eng.evaluate(d, "val", want_preds=True)
    loss_o, preds_o = model.evaluate(g["data"]["val"][d], 256)
    np.testing.assert_allclose(preds, preds_o, rtol=2e-4, atol=2e-5)
    assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    mlp_p, _ = otower.forward(model.params, g["data"]["val"][d]["uid"], g["data"]["val"][d]["pid"],
                              g["data"]["val"][d]["domain"])
    assert np.abs(mlp_p - preds_o).max() > 1e-3      # the extra terms change the predictions
    eng.close()


# ------------------------------------------------------------------ value-level parity at the largest BASELINE batches
def _assert_grads(eng, got, want, floors=None):
    for name, (off, cnt) in eng.segments.items():
        w = want[off:off + cnt]
        floor = (floors or {}).get(name, 1e-8)
        np.testing.assert_allclose(got[off:off + cnt], w, rtol=2e-4, atol=max(2e-6 * max(np.abs(w).max(), 1e-3), floor),
                                   err_msg=name)


@pytest.mark.parametrize("tower,emb_trainable,batch", [("mlp", False, 8192), ("mlp", True, 8192),
                                                      ("deepfm", False, 4096), ("deepfm", False, 8192)])
def test_large_batch_gradients_match_oracle(env, tower, emb_trainable, batch):
    """BASELINE config 5's batch size (8192) and config 3's tower with frozen tables above 2,048 rows: one SGD step
    at lr 1 = every gradient against the oracle, a full batch and the pass's final partial batch.  These sizes take
    code no small test reaches: 16-row tower tiles for all rows (k_tower<true, 0 | 256, FM>), k_wgrad with 512-row
    groups and 16 slabs, 512 loss partials."""
    g, eng, model = make_problem(env, scale=1.0, batch=batch, dropout=0.5, emb_trainable=emb_trainable, tower=tower)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    assert n > batch
    perm = orng.shuffle_perm(n, 10000, seed=12)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // batch)
    floors = {"user_emb": 6e-8, "item_emb": 6e-8}
    for step in (0, n_step - 1):
        idx = perm[step * batch:(step + 1) * batch]
        masks = otower.train_masks(model.seed, model.step, len(idx), model.hidden, 0.5)
        loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                               cols["label"][idx], masks, 0.5, emb_trainable, None, model.deepfm)
        want = eng.pack({**{k: np.zeros_like(v) for k, v in model.params.items()}, **grads}).cpu().numpy()
        w0 = eng.get_weights()
        loss_t = torch.zeros(1, device=eng.device)
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)
        model.step += 1
        _assert_grads(eng, got, want, floors)
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
    # a few Adam steps at this size too (the multi-step bar)
    k = min(3, n_step)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=k, lr=1e-3)
    for s_ in range(k):
        ii = perm[s_ * batch:(s_ + 1) * batch]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert_adam_close(got[name], model.params[name], k, 1e-3, name, max_frac=2e-3)
    eng.close()


@pytest.mark.parametrize("emb_trainable", [True, False])
def test_star_step_at_8192_rows(env, emb_trainable):
    """BASELINE config 5's batch: the Star step's gradients of EVERY shared and specific tensor, the loss and
    PartitionedNorm's moving statistics against oracle/star.py at 8192 rows (512 tower tiles, 512 statistics
    chunks, k_wgrad with 512-row groups and 16 slabs), full batch and the final partial batch."""
    from oracle import star as ostar
    g, eng, model = make_star_problem(env, emb_trainable, scale=1.0, batch=8192)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    assert n > 8192
    perm = orng.shuffle_perm(n, 10000, seed=4)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_steps = -(-n // 8192)
    for step in (0, n_steps - 1):
        idx = perm[step * 8192:(step + 1) * 8192]
        loss, grads, _, c = ostar.loss_and_grads(model.params, model.state, cols["uid"][idx], cols["pid"][idx],
                                                 cols["domain"][idx], cols["label"][idx], emb_trainable)
        ostar.update_moving(model.state, c["d"], c["mean"], c["var"])
        want = eng.pack(grads).cpu().numpy()
        w0 = eng.get_weights()
        loss_t = torch.zeros(1, device=eng.device)
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)
        for name, (off, cnt) in eng.segments.items():
            w = want[off:off + cnt]
            if name == "domain_emb":       # constant over a single-domain batch: rounding residue on both sides
                assert np.abs(got[off:off + cnt]).max() < 1e-5 and np.abs(w).max() < 1e-5
                continue
            floor = 2e-7 if name.startswith("pn_gamma") else (6e-8 if name in ("user_emb", "item_emb") or
                                                              name.startswith("Wd") else 3e-8)
            np.testing.assert_allclose(got[off:off + cnt], w, rtol=5e-4,
                                       atol=max(4e-6 * max(np.abs(w).max(), 1e-3), floor), err_msg=name)
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        aux = eng.aux_state()
        np.testing.assert_allclose(aux["steps"], model.state["steps"])
        np.testing.assert_allclose(aux["mov_mean"], model.state["mov_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(aux["mov_var"], model.state["mov_var"], rtol=1e-4, atol=1e-7)
    eng.close()


# ------------------------------------------------------------------ eval: loss, AUC bins
def test_eval_matches_oracle_and_auc_bins_exact(env):
    g, eng, model = make_problem(env, batch=256, dropout=0.5)
    thr = oauc.thresholds(500)
    for d in (0, 5):
        cols = g["data"]["val"][d]
        loss, auc, hist, preds = eng.evaluate(d, "val", want_preds=True)
        want_loss, want_preds = model.evaluate(cols, 256)
        np.testing.assert_allclose(preds, want_preds, rtol=2e-5, atol=2e-6)
        assert abs(loss - float(want_loss)) < 2e-6 * max(1.0, abs(float(want_loss)))
        # AUC confusion counts are exact integers for the kernel's own predictions
        tp, fp, tn, fn = oauc.confusion_counts(cols["label"], preds, thr)
        from mamdr_amd.engine import auc_from_histogram
        got_auc, (gtp, gfp, gtn, gfn) = auc_from_histogram(hist)
        for a, b in ((tp, gtp), (fp, gfp), (tn, gtn), (fn, gfn)):
            assert np.array_equal(a, b)
        assert got_auc == float(oauc.auc_from_counts(tp, fp, tn, fn))
        # and the AUC of the two paths agrees far inside the 1e-3 bar
        assert abs(auc - float(oauc.auc500(cols["label"], want_preds, 256))) < 1e-4
    eng.close()


# ------------------------------------------------------------------ meta loop: AUC within 1e-3
def test_mamdr_epoch_auc_parity(env):
    """two DN+DR epochs (~770 inner steps) on a 4-domain problem, same sequences / perms /
    dropout masks on both sides; meta lr 0.5 so that the model actually learns (AUC ~0.77,
    predictions spread over the threshold grid) before per-domain AUCs are compared."""
    engine, synthetic = env
    from mamdr_amd import meta
    shape = dict(synthetic.SHAPES["taobao10"], n_domain=4)
    g, eng, model = make_problem(env, scale=0.15, batch=256, dropout=0.5, shape=shape)
    D = g["n_domain"]
    plan = {"seq": [2, 0, 3, 1], "dr": [(2, [0, 3, 2]), (0, [1, 2, 0]), (3, [2, 1, 3]), (1, [3, 0, 1])]}
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]

    def make_perm_fn():
        counter = [0]

        def perm_fn(d):
            counter[0] += 1
            return orng.shuffle_perm(sizes[d], 10000, seed=1000 + counter[0])
        return perm_fn

    rs = np.random.RandomState(5)
    theta0 = model.get_flat().copy()
    phis0 = [(rs.standard_normal(theta0.size) * 0.001).astype(F32) for _ in range(D)]
    META_LR, EPOCHS = 0.5, 2
    # --- oracle
    theta_o = theta0.copy()
    phis_o = [p.copy() for p in phis0]
    pf = make_perm_fn()
    trace_o = []
    for _ in range(EPOCHS):
        trace_o += oloops.mamdr_epoch(model, theta_o, phis_o, g["data"]["train"], plan, pf, 256, META_LR)
    # --- HIP path
    def to_dev(flat):
        named, o = {}, 0
        for nme in model.names:
            sz = model.params[nme].size
            named[nme] = flat[o:o + sz]
            o += sz
        return eng.pack(named)

    theta_g = to_dev(theta0)
    phis_g = [to_dev(p) for p in phis0]
    pf = make_perm_fn()
    trace_g = []
    for _ in range(EPOCHS):
        trace_g += meta.mamdr_epoch(eng, theta_g, phis_g, plan, pf, 256, lr=1e-3, meta_lr=META_LR)
    assert trace_g == trace_o
    # per-domain val AUC with merged weights theta + phi_d
    merged = eng.new_vector()
    for d in range(D):
        eng.merge(merged, theta_g, phis_g[d], "plus")
        eng.set_weights(merged)
        _, auc_g = eng.evaluate(d, "val")
        model.set_flat(oouter.merge(theta_o, phis_o[d], "plus"))
        _, preds = model.evaluate(g["data"]["val"][d], 256)
        auc_o = float(oauc.auc500(g["data"]["val"][d]["label"], preds, 256))
        print("domain %d: val rows %d  AUC hip %.5f oracle %.5f" % (d, len(preds), auc_g, auc_o))
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)
        assert auc_o > 0.6          # the comparison is made on a model that has actually learnt
    eng.close()


@pytest.mark.parametrize("loop", ["dn", "mamdr"])
def test_scattered_meta_parms_epochs_match_oracle(env, loop):
    """`meta_parms` lists whose tensors are no neighbours in the flat vector (maml.py:167-177): theta / phi span the range
    from the first selected tensor to the last, `assign_meta` leaves the tensors in between (holes) to the inner steps.
    One DN epoch / one DN + DR epoch against the oracle's loops on a model whose flat vector is the chosen tensors alone
    (tests/test_host_logic.py checks the same bit for bit on the CPU stand-in): equal traces, theta and every LIVE tensor
    within the Adam bar of the steps taken since its last reset, per-domain val AUC within 1e-3."""
    engine, synthetic = env
    from fake_engine import MetaSubset
    from mamdr_amd import meta
    shape = dict(synthetic.SHAPES["taobao10"], n_domain=4)
    g, eng, model = make_problem(env, scale=0.1, batch=256, dropout=0.5, shape=shape)
    D = g["n_domain"]
    chosen = ["domain_emb", "W1", "b1", "gb"]
    segs = eng.segments
    order = sorted(segs, key=lambda n: segs[n][0])
    lo, hi = segs[chosen[0]][0], segs[chosen[-1]][0] + segs[chosen[-1]][1]
    holes, run = [], None
    for n in order:
        o, c = segs[n]
        if n in chosen:
            if run:
                holes.append((run[0] - lo, run[1] - run[0]))
            run = None
        else:
            run = (run[0] if run else o, o + c)
    assert len(holes) == 3 and lo == 0
    eng.set_meta_range(lo, hi - lo, holes)
    sub = MetaSubset(model, chosen)
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]

    def make_perm_fn():
        counter = [0]

        def perm_fn(d):
            counter[0] += 1
            return orng.shuffle_perm(sizes[d], 10000, seed=2000 + counter[0])
        return perm_fn

    def to_dev(flat):           # chosen tensors -> a vector over the whole range (holes zero)
        t = torch.zeros(hi - lo, dtype=torch.float32)
        o = 0
        for n in chosen:
            off, cnt = segs[n]
            t[off - lo:off - lo + cnt] = torch.from_numpy(flat[o:o + cnt])
            o += cnt
        return t.to(eng.device)

    def from_dev(t):
        h = t.cpu().numpy()
        return np.concatenate([h[segs[n][0] - lo:segs[n][0] - lo + segs[n][1]] for n in chosen])

    init = {k: v.copy() for k, v in model.params.items()}
    theta_o = sub.get_flat().copy()
    theta_g = to_dev(theta_o)
    META_LR = 0.5
    seq = [2, 0, 3, 1]
    if loop == "dn":
        trace_o = oloops.dn_epoch(sub, theta_o, g["data"]["train"], seq, make_perm_fn(), 256, META_LR)
        trace_g = meta.dn_epoch(eng, theta_g, seq, make_perm_fn(), 256, 1e-3, META_LR)
    else:
        plan = {"seq": seq, "dr": [(2, [0, 3]), (0, [1, 2]), (3, [2, 1]), (1, [3, 0])]}
        rs = np.random.RandomState(5)
        phis_o = [(rs.standard_normal(theta_o.size) * 0.001).astype(F32) for _ in range(D)]
        phis_g = [to_dev(p) for p in phis_o]
        trace_o = oloops.mamdr_epoch(sub, theta_o, phis_o, g["data"]["train"], plan, make_perm_fn(), 256, META_LR)
        trace_g = meta.mamdr_epoch(eng, theta_g, phis_g, plan, make_perm_fn(), 256, lr=1e-3, meta_lr=META_LR)
    assert trace_g == trace_o
    n_steps = sum(t[2] for t in trace_o)
    got = eng.unpack(eng.get_weights())
    th = from_dev(theta_g)
    o = 0
    for n in chosen:
        cnt = segs[n][1]
        assert_adam_close(th[o:o + cnt], theta_o[o:o + cnt], n_steps, 1e-3, "theta " + n, max_frac=2e-3)
        o += cnt
    for n in order:
        if n in chosen:
            continue
        # a hole tensor was never reset: it is where the inner steps of the whole epoch took it -- far from its start
        # (had an assignment reached it, it would sit at the start value or at an interpolation of it)
        moved = float(np.abs(model.params[n] - init[n]).mean())
        diff = float(np.abs(got[n].ravel() - model.params[n].ravel()).mean())
        assert moved > 0 and diff < 0.05 * moved, (n, diff, moved)
    if loop == "mamdr":
        for d in range(D):
            ph = from_dev(phis_g[d])
            assert float(np.abs(ph - phis_o[d]).mean()) < 0.05 * float(np.abs(phis_o[d]).mean()), d
    merged = eng.new_vector(meta=True) if hasattr(eng, "new_vector") else None
    for d in range(D):
        if loop == "mamdr":
            eng.merge(merged, theta_g, phis_g[d], "plus")
            eng.assign_meta(merged)
            sub.set_flat(oouter.merge(theta_o, phis_o[d], "plus"))
        _, auc_g = eng.evaluate(d, "val")
        _, preds = model.evaluate(g["data"]["val"][d], 256)
        auc_o = float(oauc.auc500(g["data"]["val"][d]["label"], preds, 256))
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)
    eng.close()


# ------------------------------------------------------------------ MAML: accumulate mode + outer Adam
def test_maml_epoch_matches_oracle(env):
    """first-order MAML epoch (maml.py:62-116): inner Adam pass, meta pass accumulating gradients with
    dropout off, outer TF1 Adam on theta with its own slots."""
    engine, synthetic = env
    from mamdr_amd import meta
    shape = dict(synthetic.SHAPES["taobao10"], n_domain=3)
    g, eng, model = make_problem(env, scale=0.05, batch=256, dropout=0.5, shape=shape)
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(3)]

    def make_perm_fn():
        k = [0]

        def f(d):
            k[0] += 1
            return orng.shuffle_perm(sizes[d], 10000, seed=500 + k[0])
        return f

    # accumulate mode alone: one batch, gradient at the current weights, dropout off, no update
    acc = eng.new_vector()
    eng.bind_accumulator(acc)
    w0 = eng.get_weights().cpu().numpy()
    perm = orng.shuffle_perm(sizes[0], 10000, seed=3)
    eng.train_steps(0, perm=torch.from_numpy(perm).to(eng.device), first_step=0, n_steps=2, optimizer="accumulate")
    acc_o = np.zeros(model.get_flat().size, F32)
    cols = g["data"]["train"][0]
    for s_ in range(2):
        ii = perm[s_ * 256:(s_ + 1) * 256]
        model.accumulate_on_batch(acc_o, cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(acc)
    want = {}
    o = 0
    for nme in model.names:
        want[nme] = acc_o[o:o + model.params[nme].size]
        o += model.params[nme].size
    for nme in model.names:
        np.testing.assert_allclose(got[nme], want[nme], rtol=2e-4, atol=2e-6 * max(np.abs(want[nme]).max(), 1e-3))
    assert np.array_equal(eng.get_weights().cpu().numpy(), w0)            # weights untouched
    assert int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == 0
    # a full epoch, per-domain outer steps
    acc.zero_()
    theta_o = model.get_flat().copy()
    tr_o = oloops.maml_epoch(model, theta_o, otower.OuterAdam(theta_o.size), np.zeros_like(theta_o), g["data"]["train"],
                             [2, 0, 1], make_perm_fn(), 256, 0.01)
    theta_g = eng.get_weights()
    tr_g = meta.maml_epoch(eng, theta_g, meta.OuterAdamState(eng), acc, [2, 0, 1], make_perm_fn(), 256, lr=1e-3,
                           meta_lr=0.01)
    assert tr_g == tr_o
    got = eng.unpack(theta_g)
    o = 0
    for nme in model.names:
        sz = model.params[nme].size
        # three outer Adam steps of size ~meta_lr each (see assert_adam_close for the bar).  The outer Adam's
        # first steps are sign-like: a hidden unit that is dead on one path and revived by one rounding-level
        # activation on the other moves its incoming weight column and outgoing weight row (<1 % of a tensor) by ~meta_lr.
        assert_adam_close(got[nme], theta_o[o:o + sz], 3, 0.01, nme, max_frac=2e-3)
        o += sz
    eng.close()


def test_moving_average_op_and_meta_pass(env):
    """average_meta_grad == "moving_mean" (maml.py:219-220): mamdr_moving_average is bit-exact against the oracle's
    restatement of TF 1.12's zero-debiased moving average; an accumulate pass under set_moving_average leaves the
    debiased average of the batch gradients in the accumulator (hidden state kept across a cleared accumulator)."""
    engine, synthetic = env
    from mamdr_amd import _lib as L
    shape = dict(synthetic.SHAPES["taobao10"], n_domain=3)
    g, eng, model = make_problem(env, scale=0.05, batch=256, dropout=0.5, shape=shape)
    rs = np.random.RandomState(5)
    n = 100003
    u = rs.standard_normal(n).astype(F32)
    b = (rs.standard_normal(n) * 0.1).astype(F32)
    ud, bd = torch.from_numpy(u).to(eng.device), torch.from_numpy(b).to(eng.device)
    step = 0
    for k in range(4):
        v = rs.standard_normal(n).astype(F32)
        step = oouter.moving_average_update(u, b, v, 0.999, step)
        decay = F32(1.0 - 0.999)
        denom = F32(1.0) - np.power(F32(1.0) - decay, F32(step), dtype=F32)
        vd = torch.from_numpy(v).to(eng.device)
        L.check(eng.lib.mamdr_moving_average(ud.data_ptr(), bd.data_ptr(), vd.data_ptr(), float(decay), float(denom), n, None))
        assert np.array_equal(ud.cpu().numpy(), u) and np.array_equal(bd.cpu().numpy(), b)
    # the meta pass
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(3)]
    acc = eng.new_vector()
    eng.set_moving_average(0.999)
    eng.bind_accumulator(acc)
    model.moving_average = {"momentum": 0.999, "step": 0, "biased": np.zeros(model.get_flat().size, F32)}
    acc_o = np.zeros(model.get_flat().size, F32)
    for rnd, d in enumerate((0, 2)):
        perm = orng.shuffle_perm(sizes[d], 10000, seed=3 + rnd)
        eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), first_step=0, n_steps=3, optimizer="accumulate")
        cols = g["data"]["train"][d]
        for s_ in range(3):
            ii = perm[s_ * 256:(s_ + 1) * 256]
            model.accumulate_on_batch(acc_o, cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
        got = eng.unpack(acc)
        o = 0
        for nme in model.names:
            sz = model.params[nme].size
            want = acc_o[o:o + sz]
            # (an average of batch gradients of either sign: each agrees to 2e-4 of ITS size, the bar is absolute)
            np.testing.assert_allclose(got[nme], want, rtol=3e-4, atol=1e-3 * max(np.abs(want).max(), 1e-3), err_msg=nme)
            o += sz
        acc.zero_()                 # clear_grads (maml.py:203) leaves `biased` / `local_step` alone
        acc_o[...] = 0
    assert eng._ema["step"] == 6 == model.moving_average["step"]
    eng.close()


# ------------------------------------------------------------------ full size: size-independent properties
def test_full_size_properties_taobao10(env):
    """BASELINE config sizes (Taobao-10, bs 1024): bit-exact gather, bitwise run-to-run
    determinism of a pass (no float atomics), loss goes down, AUC leaves 0.5."""
    g, eng, model = make_problem(env, scale=1.0, batch=1024, dropout=0.5)
    d = 5                                     # the largest domain (31k rows)
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=77)
    perm_t = torch.from_numpy(perm).to(eng.device)
    out = eng.gather(d, "train", perm=perm_t).cpu().numpy()
    assert same_bits(out, otower.gather(model.params, cols["uid"][perm], cols["pid"][perm], cols["domain"][perm]))
    n_steps = -(-n // 1024)
    losses = torch.zeros(n_steps, device=eng.device)
    _, auc_before = eng.evaluate(d, "val")
    eng.train_steps(d, perm=perm_t, lr=1e-3, loss_out=losses)
    for _ in range(3):
        eng.train_steps(d, perm=perm_t, lr=1e-3)
    w1 = eng.get_weights().cpu().numpy()
    _, auc_after = eng.evaluate(d, "val")
    assert np.isfinite(w1).all()
    l = losses.cpu().numpy()
    assert l[-3:].mean() < l[:3].mean()
    assert auc_after > max(0.55, auc_before - 0.02)
    eng.close()
    # determinism without dropout: two engines, same inputs -> same bits after a full pass
    outs = []
    for _ in range(2):
        g2, e2, _ = make_problem(env, scale=1.0, batch=1024, dropout=0.0)
        e2.train_steps(d, perm=perm_t, lr=1e-3)
        outs.append(e2.get_weights().cpu().numpy())
        e2.close()
    assert same_bits(outs[0], outs[1])


# ------------------------------------------------------------------ Star tower
def make_star_problem(env, emb_trainable, scale=0.1, batch=256, seed=11):
    from oracle import star as ostar
    engine, synthetic = env
    g = synthetic.generate("taobao10", batch_size=batch, seed=seed, scale=scale)
    rs = np.random.RandomState(seed)
    p = ostar.init_params(rs, g["n_user"], g["n_item"], g["n_domain"])
    p["user_emb"] = g["tables"]["user_emb"].copy()
    p["item_emb"] = g["tables"]["item_emb"].copy()
    # off the special initial values, effective kernels of useful size
    for n in ("pn_gamma_shared", "pn_gamma_spec"):
        p[n] = (p[n] + rs.standard_normal(p[n].shape) * 0.2).astype(F32)
    for n in ("pn_beta_shared", "pn_beta_spec", "bs0", "bs1", "bs2", "bd0", "bd1", "bd2", "gb"):
        p[n] = (rs.standard_normal(p[n].shape) * 0.05).astype(F32)
    for l in range(3):
        p["Wd%d" % l] = (p["Wd%d" % l] * 8).astype(F32)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], batch, dropout=0.0,
                             emb_trainable=emb_trainable, tower="star")
    if not emb_trainable:
        eng.bind_table("user_emb", p["user_emb"])
        eng.bind_table("item_emb", p["item_emb"])
    for split in ("train", "val", "test"):
        for d in range(g["n_domain"]):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    eng.set_weights(eng.pack(p))
    model = ostar.OracleStar({k: v.copy() for k, v in p.items()}, emb_trainable=emb_trainable, lr=1e-3)
    return g, eng, model


@pytest.mark.parametrize("emb_trainable", [True, False])
def test_star_lazy_slices_equal_the_per_step_sweep(env, emb_trainable):
    """Star tower: the per-domain slices of the domains a batch does not carry only decay (zero gradient, TF1 dense
    Adam).  Inside a mamdr_train_steps call they are replayed once, at its end (k_star_catchup), instead of being swept
    every step (MAMDR_STAR_DENSE_SLICES=1): weights and both Adam slots identical bit for bit, over calls on different
    domains, single-step calls (swept as before), a call starting in the middle of a pass and SGD steps in between."""
    out = {}
    for mode in ("lazy", "dense"):
        os.environ["MAMDR_STAR_DENSE_SLICES"] = "1" if mode == "dense" else "0"
        try:
            g, eng, model = make_star_problem(env, emb_trainable, batch=256)
        finally:
            os.environ.pop("MAMDR_STAR_DENSE_SLICES", None)
        sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(g["n_domain"])]
        order = sorted(range(g["n_domain"]), key=lambda d: -sizes[d])[:4]
        for k, d in enumerate(order):
            perm = torch.from_numpy(orng.shuffle_perm(sizes[d], 10000, seed=21 + k)).to(eng.device)
            n = -(-sizes[d] // 256)
            eng.train_steps(d, perm=perm, first_step=0, n_steps=min(n, 5), lr=1e-3)          # several steps: replayed
            eng.train_steps(d, perm=perm, first_step=min(n, 5) - 1, n_steps=1, lr=1e-3)     # one step: swept
            if k == 1:
                eng.train_steps(d, perm=perm, first_step=1, n_steps=2, lr=1e-3, optimizer="sgd")
                eng.train_steps(d, perm=perm, first_step=2, n_steps=3, lr=1e-3)             # from the middle of a pass
        out[mode] = [eng.get_weights().cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy()]
        eng.close()
    for a, b, what in zip(out["lazy"], out["dense"], ("weights", "adam m", "adam v")):
        assert same_bits(a, b), (what, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
    assert np.abs(out["lazy"][0]).max() > 0


@pytest.mark.parametrize("emb_trainable", [True, False])
def test_star_step_adam_eval(env, emb_trainable):
    """PartitionedNorm (batch statistics, zero-debiased moving statistics, backward through the statistics)
    + StarFCN (shared * specific kernels): gradients of every tensor incl. the zero-gradient slices of the
    other domains, moving statistics, a few Adam steps, inference with the moving statistics."""
    from oracle import star as ostar
    g, eng, model = make_star_problem(env, emb_trainable)
    meta_names, rest_names = ostar.param_names(emb_trainable)
    assert [n for n in meta_names + rest_names if n not in eng.segments] == []
    n_meta = sum(eng.segments[n][1] for n in meta_names)
    assert eng.n_meta == n_meta and max(eng.segments[n][0] + eng.segments[n][1] for n in meta_names) == n_meta
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=4)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_steps = -(-n // 256)
    for step in (0, n_steps - 1):          # a full batch and the final (partial) batch
        idx = perm[step * 256:(step + 1) * 256]
        loss, grads, _, c = ostar.loss_and_grads(model.params, model.state, cols["uid"][idx], cols["pid"][idx],
                                                 cols["domain"][idx], cols["label"][idx], emb_trainable)
        ostar.update_moving(model.state, c["d"], c["mean"], c["var"])
        want = eng.pack(grads).cpu().numpy()
        w0 = eng.get_weights()
        loss_t = torch.zeros(1, device=eng.device)
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)
        for name, (off, cnt) in eng.segments.items():
            w = want[off:off + cnt]
            if name == "domain_emb":
                # constant over a single-domain batch: PartitionedNorm removes it; both sides hold rounding residue
                assert np.abs(got[off:off + cnt]).max() < 1e-5 and np.abs(w).max() < 1e-5
                continue
            # p_old - p_new recovers g only to ~ulp(p): gamma ~ 1, tables / specific kernels up to ~0.8
            floor = 2e-7 if name.startswith("pn_gamma") else (6e-8 if name in ("user_emb", "item_emb") or
                                                              name.startswith("Wd") else 3e-8)
            np.testing.assert_allclose(got[off:off + cnt], w, rtol=5e-4,
                                       atol=max(4e-6 * max(np.abs(w).max(), 1e-3), floor), err_msg=name)
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        aux = eng.aux_state()
        np.testing.assert_allclose(aux["steps"], model.state["steps"])
        np.testing.assert_allclose(aux["mov_mean"], model.state["mov_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(aux["mov_var"], model.state["mov_var"], rtol=1e-4, atol=1e-7)
    # zero-gradient slices of the other domains did not move under SGD
    # a few Adam steps incl. the partial last batch; the other domains' specific tensors stay put (m = v = 0)
    first = max(0, n_steps - 3)
    eng.train_steps(d, perm=perm_t, first_step=first, n_steps=n_steps - first, lr=1e-3)
    for s_ in range(first, n_steps):
        ii = perm[s_ * 256:(s_ + 1) * 256]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        if name == "domain_emb":
            continue          # Adam normalises its rounding-residue gradient: not comparable
        assert_adam_close(got[name], model.params[name], n_steps - first, 1e-3, name)
    other = (d + 1) % 10
    assert np.array_equal(got["Wd0"].reshape(10, 384, 256)[other], model.params["Wd0"][other])
    # inference uses domain d's moving statistics
    eng.set_weights(eng.pack(model.params))
    loss_g, auc_g, hist, preds = eng.evaluate(d, "val", want_preds=True)
    loss_o, preds_o = model.evaluate(g["data"]["val"][d], 256)
    np.testing.assert_allclose(preds, preds_o, rtol=5e-4, atol=5e-5)
    assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    # a domain that never trained evaluates with the initial statistics (mean 0, variance 1)
    o2 = (d + 2) % 10
    loss_g2, _, _, preds2 = eng.evaluate(o2, "val", want_preds=True)
    loss_o2, preds_o2 = model.evaluate(g["data"]["val"][o2], 256)
    np.testing.assert_allclose(preds2, preds_o2, rtol=5e-4, atol=5e-5)
    eng.close()


# ------------------------------------------------------------------ lazy dense Adam over trainable tables
@pytest.mark.parametrize("tower", ["mlp", "deepfm", "star", "mlp-ring8"])
def test_lazy_table_adam_is_bit_identical_to_the_dense_sweep(env, tower):
    """TF1's Adam moves every table row every step.  The default path replays the steps of rows no batch
    touched lazily (catch-up before a row is gathered, flush before the weights are read); with
    MAMDR_DENSE_ADAM=1 the library sweeps the whole table every step.  Same per-element arithmetic in the
    same order -> the two must agree BITWISE in weights and both Adam slots, including rows with long gaps,
    rows repeated inside a batch, an SGD step in between and a weight assignment in between."""
    results = {}
    ring = tower.endswith("-ring8")        # an 8-entry alpha ring: the library must flush before it wraps
    tower = tower.split("-")[0]
    for mode in ("lazy", "dense"):
        os.environ["MAMDR_DENSE_ADAM"] = "1" if mode == "dense" else "0"
        if ring:
            os.environ["MAMDR_LAZY_LOG_CAP"] = "8"
        try:
            if tower == "star":
                g, eng, model = make_star_problem(env, True)
            else:
                g, eng, model = make_problem(env, scale=0.1, batch=256, dropout=0.5, emb_trainable=True, tower=tower)
        finally:
            os.environ.pop("MAMDR_DENSE_ADAM", None)
            os.environ.pop("MAMDR_LAZY_LOG_CAP", None)
        sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(10)]
        order = sorted(range(10), key=lambda k: -sizes[k])[:3]
        snap = None
        for rep_ in range(2):
            for d in order:
                perm = torch.from_numpy(orng.shuffle_perm(sizes[d], 10000, seed=17 + d + rep_)).to(eng.device)
                eng.train_steps(d, perm=perm, lr=1e-3)
            if rep_ == 0:
                snap = eng.get_weights()                                  # a read in between (forces a flush)
                eng.train_steps(order[0], first_step=0, n_steps=1, lr=1e-3, optimizer="sgd")   # dense SGD step
                eng.set_weights((snap + eng.get_weights()) * 0.5)        # weight assignment; slots persist
        results[mode] = (eng.weights.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(),
                         eng.adam_v.cpu().numpy().copy(), int(eng.lib.mamdr_optimizer_steps(eng.ctx)))
        eng.close()
    assert results["lazy"][3] == results["dense"][3] > 10
    for a, b, name in zip(results["lazy"][:3], results["dense"][:3], ("weights", "adam_m", "adam_v")):
        assert np.isfinite(a).all()
        assert same_bits(a, b), (tower, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))


@pytest.mark.parametrize("tower", ["mlp", "deepfm", "star"])
def test_fused_tail_launches_are_bit_identical_to_separate_kernels(env, tower):
    """With lazy table Adam a training step's table kernels ride in the dense kernels' launches (k_wgrad_reduce:
    k_emb_reduce(t) + the NEXT step's k_emb_rows; k_update_lin / k_star_update_catchup: k_lin_sweep(t) + the next
    step's k_emb_catchup; row ids and representative maps double-buffered).  MAMDR_NO_TAILFUSE=1 launches every
    kernel on its own.  Same bodies, same order per element -> weights and both Adam slots must agree BITWISE,
    across pass boundaries, an SGD step, a read (flush) and a weight assignment in between."""
    results = {}
    for mode in ("fused", "separate"):
        os.environ["MAMDR_NO_TAILFUSE"] = "1" if mode == "separate" else "0"
        try:
            if tower == "star":
                g, eng, model = make_star_problem(env, True)
            else:
                g, eng, model = make_problem(env, scale=0.1, batch=256, dropout=0.5, emb_trainable=True, tower=tower)
        finally:
            os.environ.pop("MAMDR_NO_TAILFUSE", None)
        sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(10)]
        order = sorted(range(10), key=lambda k: -sizes[k])[:3]
        for rep_ in range(2):
            for d in order:
                perm = torch.from_numpy(orng.shuffle_perm(sizes[d], 10000, seed=31 + d + rep_)).to(eng.device)
                eng.train_steps(d, perm=perm, lr=1e-3)
                eng.train_steps(d, perm=perm, first_step=1, n_steps=1, lr=1e-3)      # a one-step call: nothing to pre-launch
            if rep_ == 0:
                snap = eng.get_weights()
                eng.train_steps(order[0], first_step=0, n_steps=1, lr=1e-3, optimizer="sgd")
                eng.set_weights((snap + eng.get_weights()) * 0.5)
        results[mode] = (eng.get_weights().cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(),
                         eng.adam_v.cpu().numpy().copy(), int(eng.lib.mamdr_optimizer_steps(eng.ctx)))
        eng.close()
    assert results["fused"][3] == results["separate"][3] > 40
    for a, b, name in zip(results["fused"][:3], results["separate"][:3], ("weights", "adam_m", "adam_v")):
        assert np.isfinite(a).all()
        assert same_bits(a, b), (tower, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))


@pytest.mark.parametrize("tower", ["mlp", "star"])
def test_meta_epoch_lazy_equals_dense_with_trainable_tables(env, tower):
    """One DN + DR epoch (meta.mamdr_epoch: DN passes, the outer interpolation, per query domain the fused DR
    support step mamdr_dr_advance that reads the live rows into phi and replaces them by the merged weights) with
    trainable tables inside theta / phi.  Every host-side read or replacement of the live state must see the
    table rows at the current Adam step: lazy replay and MAMDR_DENSE_ADAM=1 agree BITWISE in theta, every phi,
    the live weights and both Adam slots."""
    from mamdr_amd import meta
    results = {}
    for mode in ("lazy", "dense"):
        os.environ["MAMDR_DENSE_ADAM"] = "1" if mode == "dense" else "0"
        try:
            if tower == "star":
                g, eng, model = make_star_problem(env, True, scale=0.05)
            else:
                g, eng, model = make_problem(env, scale=0.05, batch=256, dropout=0.5, emb_trainable=True, tower=tower)
        finally:
            os.environ.pop("MAMDR_DENSE_ADAM", None)
        sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(10)]
        theta = eng.get_weights()[:eng.n_meta].clone()
        rs = np.random.RandomState(5)
        phis = {d: (theta * 0 + torch.from_numpy((rs.standard_normal(eng.n_meta) * 1e-3).astype(F32)).to(eng.device))
                for d in (1, 4)}
        plan = {"seq": [3, 1, 4], "dr": [(1, [4, 3, 1]), (4, [1, 4])]}
        make_perm = _perm_fn_factory(sizes)
        for ep in range(2):
            meta.mamdr_epoch(eng, theta, phis, plan, make_perm(), 256, lr=1e-3, meta_lr=0.1)
        results[mode] = [theta.cpu().numpy().copy(), phis[1].cpu().numpy().copy(), phis[4].cpu().numpy().copy(),
                         eng.weights.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(),
                         eng.adam_v.cpu().numpy().copy()]
        eng.close()
    for a, b, name in zip(results["lazy"], results["dense"], ("theta", "phi1", "phi4", "weights", "adam_m", "adam_v")):
        assert np.isfinite(a).all()
        assert same_bits(a, b), (tower, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))


# ------------------------------------------------------------------ AUC parity of the other two BASELINE towers
def _perm_fn_factory(sizes):
    def make():
        counter = [0]

        def perm_fn(d):
            counter[0] += 1
            return orng.shuffle_perm(sizes[d], 10000, seed=2000 + counter[0])
        return perm_fn
    return make


def test_deepfm_dn_auc_parity_trainable_tables(env):
    """BASELINE config 3 in miniature: DeepFM under Domain Negotiation with TRAINABLE tables (lazy table Adam
    on the HIP side, dense numpy Adam in the oracle), three epochs, per-domain val AUC within 1e-3."""
    engine, synthetic = env
    from mamdr_amd import meta
    shape = dict(synthetic.SHAPES["taobao10"], n_domain=4)
    g, eng, model = make_problem(env, scale=0.15, batch=256, dropout=0.5, shape=shape, emb_trainable=True,
                                 tower="deepfm")
    D = g["n_domain"]
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
    make = _perm_fn_factory(sizes)
    seqs = [[2, 0, 3, 1], [1, 3, 0, 2], [0, 2, 1, 3]]
    theta_o = model.get_flat().copy()
    pf = make()
    trace_o = []
    for seq in seqs:
        trace_o += oloops.dn_epoch(model, theta_o, g["data"]["train"], seq, pf, 256, 0.5)
    theta_g = eng.get_weights()
    pf = make()
    trace_g = []
    for seq in seqs:
        trace_g += meta.dn_epoch(eng, theta_g, seq, pf, 256, lr=1e-3, meta_lr=0.5)
    assert trace_g == trace_o
    eng.set_weights(theta_g)
    model.set_flat(theta_o)
    for d in range(D):
        _, auc_g = eng.evaluate(d, "val")
        _, preds = model.evaluate(g["data"]["val"][d], 256)
        auc_o = float(oauc.auc500(g["data"]["val"][d]["label"], preds, 256))
        print("deepfm domain %d: AUC hip %.5f oracle %.5f" % (d, auc_g, auc_o))
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)
        assert auc_o > 0.6
    eng.close()


class _StarMeta(object):
    """oracle Star model seen through its meta parameters (what the MAMDR loop reads and assigns)."""

    def __init__(self, m):
        self.m = m

    def get_flat(self):
        return self.m.get_flat(meta_only=True)

    def set_flat(self, vec):
        self.m.set_flat(vec, meta_only=True)

    def train_pass(self, data, perm, batch_size, max_steps=0, accumulate_into=None):
        assert accumulate_into is None
        return self.m.train_pass(data, perm, batch_size, max_steps)


def test_star_mamdr_auc_parity(env):
    """BASELINE config 5 in miniature: Star tower under MAMDR, theta / phi over the reference's meta filter
    (shared kernels / biases + domain table; pretrained tables frozen as in config/Taobao-10/star_taobao.json),
    two DN+DR epochs, per-domain val AUC with theta + phi_d within 1e-3.

    Conditioning: fp32 training is chaotic (PartitionedNorm's batch statistics amplify rounding), so the bar only
    means something where the ORACLE ITSELF is stable to rounding.  The test therefore runs the oracle twice --
    once from theta, once from theta moved by ONE ulp -- and takes the per-domain AUC shift between the two runs
    as the width of the oracle's own answer (measured on this problem: 1e-5 .. 1e-3 depending on the domain).
    Bar: within 1e-3 of the nominal oracle run (north_star); the self-divergence is printed next to every domain so
    that a failure can be read against the oracle's own conditioning, it is not used as slack."""
    STAR_META_LR = 0.2
    from oracle import star as ostar
    engine, synthetic = env
    from mamdr_amd import meta
    g, eng, model = make_star_problem(env, False, scale=0.3)
    D = 4                                   # the first four domains take part
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(10)]
    make = _perm_fn_factory(sizes)
    plan = {"seq": [2, 0, 3, 1], "dr": [(2, [0, 3, 2]), (0, [1, 2, 0]), (3, [2, 1, 3]), (1, [3, 0, 1])]}
    model0 = {k: v.copy() for k, v in model.params.items()}          # (the perturbed twin starts from the same state)
    wrapped = _StarMeta(model)
    theta0 = wrapped.get_flat().copy()
    assert theta0.size == eng.n_meta
    rs = np.random.RandomState(5)
    phis0 = [(rs.standard_normal(theta0.size) * 0.001).astype(F32) for _ in range(D)]
    theta_o, phis_o = theta0.copy(), [p.copy() for p in phis0]
    pf = make()
    trace_o = []
    for _ in range(2):
        trace_o += oloops.mamdr_epoch(wrapped, theta_o, phis_o, g["data"]["train"], plan, pf, 256, STAR_META_LR)

    def oracle_aucs(wrapped_x, model_x, theta_x, phis_x):
        out = []
        for d in range(D):
            wrapped_x.set_flat(oouter.merge(theta_x, phis_x[d], "plus"))
            _, preds = model_x.evaluate(g["data"]["val"][d], 256)
            out.append(float(oauc.auc500(g["data"]["val"][d]["label"], preds, 256)))
        return out
    aucs_o = oracle_aucs(wrapped, model, theta_o, phis_o)
    # the same run from theta + 1 ulp: how far the oracle's own AUCs move under rounding-level noise
    model_p = ostar.OracleStar({k: v.copy() for k, v in model0.items()}, emb_trainable=False, lr=1e-3)
    wrapped_p = _StarMeta(model_p)
    theta_p = np.nextafter(theta0, F32(np.inf)).astype(F32)
    phis_p = [p.copy() for p in phis0]
    pf = make()
    for _ in range(2):
        oloops.mamdr_epoch(wrapped_p, theta_p, phis_p, g["data"]["train"], plan, pf, 256, STAR_META_LR)
    aucs_p = oracle_aucs(wrapped_p, model_p, theta_p, phis_p)
    self_div = [abs(a - b) for a, b in zip(aucs_o, aucs_p)]
    theta_g = torch.from_numpy(theta0).to(eng.device)
    phis_g = [torch.from_numpy(p).to(eng.device) for p in phis0]
    pf = make()
    trace_g = []
    for _ in range(2):
        trace_g += meta.mamdr_epoch(eng, theta_g, phis_g, plan, pf, 256, lr=1e-3, meta_lr=STAR_META_LR)
    assert trace_g == trace_o
    merged = eng.new_vector(meta=True)
    for d in range(D):
        eng.merge(merged, theta_g, phis_g[d], "plus")
        eng.set_weights(merged)
        _, auc_g = eng.evaluate(d, "val")
        auc_o = aucs_o[d]
        print("star domain %d: AUC hip %.5f oracle %.5f (oracle self-divergence %.1e)" % (d, auc_g, auc_o, self_div[d]))
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o, self_div[d])       # north_star: per-domain AUC within 1e-3
        assert auc_o > 0.6
    eng.close()


# ------------------------------------------------------------------ sub-range passes (meta-train / meta-val split)
def test_pass_window_matches_oracle(env):
    """mamdr_train_steps_n: a pass over a take / skip slice of the split (maml.py:300-330): the permutation
    lists only the slice's rows and the final partial batch ends at the slice's end."""
    g, eng, model = make_problem(env, batch=256, dropout=0.5)
    d = 9
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    n_train = int(n * 0.8)
    for (b, e), seed in (((0, n_train), 3), ((n_train, n), 4)):
        perm = (orng.shuffle_perm(e - b, 10000, seed=seed) + b).astype(np.int32)
        n_steps = -(-(e - b) // 256)
        losses_t = torch.zeros(n_steps, device=eng.device)
        got_steps = eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), lr=1e-3, loss_out=losses_t,
                                    pass_rows=e - b)
        assert got_steps == n_steps
        want = model.train_pass(cols, perm, 256)
        assert len(want) == n_steps and (e - b) % 256 != 0           # a partial final batch is exercised
        np.testing.assert_allclose(losses_t.cpu().numpy(), np.array(want, F32), rtol=2e-5, atol=2e-6)
    assert int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == model.opt.t
    # a pass longer than the split is rejected
    with pytest.raises(Exception):
        eng.train_steps(d, pass_rows=n + 1)
    eng.close()


# ------------------------------------------------------------------ uncertainty weighting
@pytest.mark.parametrize("batch", [256, 4096])
def test_uncertainty_weighted_step_matches_oracle(env, batch):
    """weighted_loss.py:30-43 on the step kernels (4-row tower at 256, 16-row tower at 4096): every gradient
    scaled by 1 / var_d^2, d loss / d var_d, zero gradient for the other domains' scalars; a few Adam steps;
    evaluation unweighted."""
    g, eng, model = make_problem(env, batch=batch, dropout=0.5, uncertainty=True, scale=0.3 if batch > 256 else 0.05)
    assert eng.segments["log_var"][1] == 10 and model.names[-1] == "log_var"
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=11)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // batch)
    for step in (0, n_step - 1):
        idx = perm[step * batch:(step + 1) * batch]
        masks = otower.train_masks(model.seed, model.step, len(idx), model.hidden, 0.5)
        loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                               cols["label"][idx], masks, 0.5, False, None, False, True)
        want = eng.pack({**{k: np.zeros_like(v) for k, v in model.params.items()}, **grads}).cpu().numpy()
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = (w0 - eng.get_weights()).cpu().numpy()
        eng.set_weights(w0)
        model.step += 1
        for name, (off, cnt) in eng.segments.items():
            w = want[off:off + cnt]
            np.testing.assert_allclose(got[off:off + cnt], w, rtol=2e-4, atol=max(2e-6 * max(np.abs(w).max(), 1e-3), 1e-7),
                                       err_msg=name)
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        lv = got[eng.segments["log_var"][0]:][:10]
        assert lv[d] != 0 and not np.delete(lv, d).any()
    k = min(3, n_step)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=k, lr=1e-3)
    for s_ in range(k):
        ii = perm[s_ * batch:(s_ + 1) * batch]
        model.train_on_batch(cols["uid"][ii], cols["pid"][ii], cols["domain"][ii], cols["label"][ii])
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert_adam_close(got[name], model.params[name], k, 1e-3, name, max_frac=2e-3)
    # evaluation is the base model's: unweighted loss
    loss_g, _ = eng.evaluate(d, "val")
    loss_o, _ = model.evaluate(g["data"]["val"][d], eng.eval_batch)
    assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    eng.close()


# ------------------------------------------------------------------ PCGrad projection: bit-exact vs the reference's numpy
def test_pcgrad_projection_bit_exact_vs_reference_goldens(env, golden_dir):
    """mamdr_pcgrad_project against vectors produced by the reference's own PCGrad.PCGrad
    (tests/golden/make_pcgrad_goldens.py): the kernel walks every slice in numpy's pairwise-summation order."""
    g, eng, model = make_problem(env)
    G = np.load(os.path.join(golden_dir, "pcgrad_goldens.npz"))
    n = int(G["n_tensors"])
    tensors, off = [], 0
    for i in range(n):
        a = G["current_%d" % i]
        cols = a.shape[-1]
        tensors.append((off, a.size // cols, cols))
        off += (a.size + 3) // 4 * 4           # 16-B aligned starts, as in the engine's flat vectors

    def flat(prefix):
        v = np.zeros(off, F32)
        for i, (o, r, c) in enumerate(tensors):
            v[o:o + r * c] = G["%s_%d" % (prefix, i)].ravel()
        return torch.from_numpy(v).to(eng.device)

    fin = flat("current")
    for tag, step in (("aux1", "after1"), ("aux2", "after2")):
        aux = flat(tag)
        eng.pcgrad_project(fin, aux, tensors)
        fh, ah = fin.cpu().numpy(), aux.cpu().numpy()
        for i, (o, r, c) in enumerate(tensors):
            assert same_bits(fh[o:o + r * c], G["%s_final_%d" % (step, i)].ravel()), (step, "final", i)
            assert same_bits(ah[o:o + r * c], G["%s_aux_%d" % (step, i)].ravel()), (step, "aux", i)
    # the engine's own tensor table covers the whole flat vector exactly once
    shapes = eng.segment_shapes()
    assert sum(r * c for r, c in shapes.values()) == sum(cnt for _, cnt in eng.segments.values())
    eng.close()


def test_pcgrad_epoch_matches_oracle(env):
    """one PCGrad epoch (pcgrad.py:62-124) on three domains: accumulate-mode passes, bit-exact projection of
    the auxiliary gradients, outer Adam on the live model."""
    from mamdr_amd import meta
    shape = dict(env[1].SHAPES["taobao10"], n_domain=3)
    g, eng, model = make_problem(env, scale=0.15, batch=256, dropout=0.5, shape=shape)
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(3)]
    make = _perm_fn_factory(sizes)
    aux_plan = {1: [0, 2], 0: [2, 1], 2: [1]}
    tr_o = oloops.pcgrad_epoch(model, otower.OuterAdam(model.get_flat().size), g["data"]["train"], [1, 0, 2], aux_plan,
                               make(), 256, 0.003)
    cur, aux = eng.new_vector(), eng.new_vector()
    tr_g = meta.pcgrad_epoch(eng, meta.OuterAdamState(eng), cur, aux, [1, 0, 2], aux_plan, make(), 256, lr=1e-3,
                             meta_lr=0.003)
    assert tr_g == tr_o
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        assert_adam_close(got[name], model.params[name], 3, 0.003, name, max_frac=2e-3)
    assert int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == 0          # the inner optimiser never stepped
    eng.close()


# ------------------------------------------------------------------ the other BASELINE configs at full table / batch size
@pytest.mark.parametrize("workload", ["taobao30", "amazon6", "amazon13"])
def test_full_size_properties_other_baseline_configs(env, workload):
    """BASELINE configs 3-5 at their full table sizes and batch sizes (Taobao-30 bs 4096 frozen; Amazon-6
    DeepFM bs 1024 and Amazon-13 Star bs 8192 with trainable 79 M / 92 M-parameter tables; a 0.3 % sample of
    the rows), checked through size-independent properties: bit-exact gather, the lazy table Adam bitwise equal
    to the per-step dense sweep (weights and both Adam slots), bitwise run-to-run determinism, finite and
    non-increasing loss."""
    import bench
    engine, synthetic = env
    wl = bench.WORKLOADS[workload]
    batch, trainable, tower = wl["batch"], bool(wl.get("emb_trainable")), wl.get("tower", "mlp")
    g = synthetic.generate(wl["shape"], batch_size=batch, seed=123, row_scale={"amazon6": 0.003, "amazon13": 0.02}.get(workload, 1.0))
    D = g["n_domain"]
    if tower == "star":
        from mamdr_amd.model_zoo.star import initial_tensors
        params = initial_tensors(np.random.RandomState(1), g["n_user"], g["n_item"], D, 128, (256, 128, 64))
    else:
        params = bench.init_params(g, 1)
        rs = np.random.RandomState(2)
        params["user_emb"] = g["tables"]["user_emb"] if not trainable else \
            (rs.standard_normal((g["n_user"], 128)) * 0.01).astype(F32)
        params["item_emb"] = g["tables"]["item_emb"] if not trainable else \
            (rs.standard_normal((g["n_item"], 128)) * 0.01).astype(F32)
    sizes = [g["data"]["train"][k]["uid"].shape[0] for k in range(D)]
    order = sorted(range(D), key=lambda k: -sizes[k])[:2]
    perms = {k: orng.shuffle_perm(sizes[k], 10000, seed=5 + k) for k in order}

    def run(dense):
        os.environ["MAMDR_DENSE_ADAM"] = "1" if dense else "0"
        try:
            eng = bench.setup_engine(g, batch, trainable, tower)
        finally:
            os.environ.pop("MAMDR_DENSE_ADAM", None)
        eng.set_weights(eng.pack(params))
        losses = []
        for rep_ in range(2):
            for k in order:
                n_steps = -(-sizes[k] // batch)
                lt = torch.zeros(n_steps, device=eng.device)
                eng.train_steps(k, perm=torch.from_numpy(perms[k]).to(eng.device), lr=1e-3, loss_out=lt)
                losses.append(lt.cpu().numpy())
        return eng, losses

    eng, losses = run(False)
    # bit-exact gather of the largest domain in shuffled order (first 4096 positions)
    k = order[0]
    cols = g["data"]["train"][k]
    m = min(4096, sizes[k])
    out = eng.gather(k, "train", perm=torch.from_numpy(perms[k]).to(eng.device), n_rows=m).cpu().numpy()
    w = eng.unpack(eng.weights)
    U = w["user_emb"].reshape(-1, 128) if trainable else params["user_emb"]
    I = w["item_emb"].reshape(-1, 128) if trainable else params["item_emb"]
    idx = perms[k][:m]
    want = np.concatenate([U[cols["uid"][idx]], I[cols["pid"][idx]], w["domain_emb"].reshape(D, 128)[cols["domain"][idx]]],
                          axis=1)
    assert same_bits(out, want)
    state = (eng.weights.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy())
    eng.close()
    assert all(np.isfinite(l).all() for l in losses) and np.isfinite(state[0]).all()
    first, last = losses[0], losses[-2]               # the largest domain's first and second pass
    assert last.mean() <= first.mean() + 1e-3, (float(first.mean()), float(last.mean()))
    # the same run again (dense table sweep for the trainable configs): same bits
    eng2, losses2 = run(trainable)
    for a, b, name in zip(state, (eng2.weights, eng2.adam_m, eng2.adam_v), ("weights", "adam_m", "adam_v")):
        assert same_bits(a, b.cpu().numpy()), (workload, name)
    for a, b in zip(losses, losses2):
        assert same_bits(a, b)
    eng2.close()
