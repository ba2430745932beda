"""GPU edge cases of the hot path, through the C ABI: empty and ragged inputs, single rows, one-class splits, the largest
batch a context allows, out-of-range requests.  The reference meets these through tf.data (`utils/dataset.py:20-38`: the final
partial batch of a pass is kept, a pass over n rows has ceil(n / B) steps) and Keras (`model.evaluate(steps=n_step)`,
`base_model.py:111-144`); what it leaves undefined (empty domains crash its iterators) is an error code here, never a
silent no-op with garbage outputs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import auc as oauc          # noqa: E402
from oracle import rng as orng          # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32


def bits(a):
    return np.asarray(a, F32).view(np.uint32)


def make(batch=256, max_rows=None, tower="mlp", emb_trainable=False, dropout=0.5, seed=11, scale=0.05):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import engine, synthetic
    g = synthetic.generate("taobao10", batch_size=batch, seed=seed, scale=scale)
    rs = np.random.RandomState(seed)
    params = otower.init_params(rs, g["n_user"], g["n_item"], g["n_domain"])
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"].copy(), g["tables"]["item_emb"].copy()
    params["domain_emb"] = (rs.standard_normal(params["domain_emb"].shape) * 0.05).astype(F32)
    for l in range(3):
        params["b%d" % l] = (rs.standard_normal(params["b%d" % l].shape) * 0.05).astype(F32)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], batch, dropout=dropout, emb_trainable=emb_trainable,
                             tower=tower)
    if not emb_trainable:
        eng.bind_table("user_emb", params["user_emb"])
        eng.bind_table("item_emb", params["item_emb"])
    data = {}
    for split in ("train", "val", "test"):
        for d in range(g["n_domain"]):
            c = g["data"][split][d]
            if max_rows is not None and d in max_rows:
                c = {k: v[:max_rows[d]] for k, v in c.items()}
            data[(split, d)] = c
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    eng.set_weights(eng.pack(params))
    model = otower.OracleModel({k: v.copy() for k, v in params.items()}, emb_trainable=emb_trainable, dropout=dropout,
                               lr=1e-3, dropout_seed=eng.dropout_seed, tower=tower)
    return g, eng, model, data


def test_zero_steps_is_a_no_op():
    """n_steps = 0 (a caller whose meta_train_step caps a pass at nothing): weights, both Adam slots and the optimiser /
    dropout step counts stay bit-unchanged."""
    g, eng, model, data = make()
    w0, m0, v0 = eng.weights.clone(), eng.adam_m.clone(), eng.adam_v.clone()
    t0, s0 = eng.lib.mamdr_optimizer_steps(eng.ctx), eng.lib.mamdr_dropout_steps(eng.ctx)
    assert eng.train_steps(0, n_steps=0) == 0
    eng.sync()
    assert torch.equal(eng.weights, w0) and torch.equal(eng.adam_m, m0) and torch.equal(eng.adam_v, v0)
    assert eng.lib.mamdr_optimizer_steps(eng.ctx) == t0 and eng.lib.mamdr_dropout_steps(eng.ctx) == s0
    eng.close()


@pytest.mark.parametrize("rows", [1, 17, 257])
def test_tiny_and_ragged_passes_match_the_oracle(rows):
    """a domain of 1 row (one step on a single-row batch), of 17 rows (one ragged batch) and of 257 rows (a full batch and
    a final batch of ONE row -- `utils/dataset.py:25` keeps it): losses and weights after the pass against the oracle."""
    from test_gpu_parity import assert_adam_close
    g, eng, model, data = make(max_rows={3: rows})
    c = data[("train", 3)]
    assert c["uid"].shape[0] == rows
    perm = orng.shuffle_perm(rows, 10000, seed=5)
    n_steps = -(-rows // 256)
    losses = torch.zeros(n_steps, device=eng.device)
    assert eng.train_steps(3, perm=torch.from_numpy(perm).to(eng.device), lr=1e-3, loss_out=losses) == n_steps
    want = model.train_pass(c, perm, 256)
    np.testing.assert_allclose(losses.cpu().numpy(), np.asarray(want, F32), rtol=3e-5, atol=1e-6)
    got = eng.unpack(eng.get_weights())
    for name in ("W0", "W1", "W2", "b0", "wo", "domain_emb"):
        assert_adam_close(got[name], model.params[name], n_steps, 1e-3, name, max_frac=2e-3)
    # steps beyond the pass are refused, nothing runs
    from mamdr_amd import _lib as L
    with pytest.raises(L.MamdrError):
        eng.train_steps(3, first_step=n_steps, n_steps=1)
    eng.close()


def test_single_row_eval_and_one_class_split():
    """evaluation of a 1-row split and of a split whose labels are all 0: loss against the oracle, integer confusion
    counts exact, AUC-500 of a one-class split as the reference's metric gives it (`utils/auc.py:248-281`: 0 / 0 guarded by
    div_no_nan -> 0)."""
    g, eng, model, data = make(max_rows={2: 1})
    c = {k: v.copy() for k, v in data[("val", 4)].items()}
    c["label"][:] = 0.0
    eng.bind_domain_data(4, "val", c["uid"], c["pid"], c["domain"], c["label"])
    for d, split, cols in ((2, "val", data[("val", 2)]), (4, "val", c)):
        loss, auc, hist, preds = eng.evaluate(d, split, want_preds=True)
        loss_o, preds_o = model.evaluate(cols, 256)
        np.testing.assert_allclose(preds, preds_o, rtol=2e-5, atol=2e-7)
        assert abs(loss - float(loss_o)) < 2e-6 * max(1.0, abs(float(loss_o)))
        assert abs(auc - float(oauc.auc500(cols["label"], preds, 256))) < 1e-7
        assert int(hist.sum()) == cols["label"].shape[0]
    eng.close()


def test_empty_split_and_bad_arguments_are_errors():
    from mamdr_amd import _lib as L
    g, eng, model, data = make(max_rows={1: 0})
    assert eng.n_rows(1, "train") == 0
    assert eng.train_steps(1) == 0                      # ceil(0 / B) = 0 steps: nothing to run, nothing changes
    with pytest.raises(L.MamdrError):
        eng.train_steps(1, n_steps=1)                   # a step on an empty domain
    with pytest.raises(L.MamdrError):
        eng.evaluate(1, "val")                          # `model.evaluate(steps=0)` has no defined result
    with pytest.raises(L.MamdrError):
        eng.train_steps(0, batch_size=eng.batch_size + 16)      # beyond max_batch
    with pytest.raises(L.MamdrError):
        eng.train_steps(0, batch_size=-16)
    with pytest.raises(L.MamdrError):                   # no such domain (through the C entry point: the binding has no columns for it)
        L.check(eng.lib.mamdr_train_steps(eng.ctx, g["n_domain"], None, 0, 1, eng.batch_size, eng.dropout_seed, L.OPT_ADAM, 1e-3, None))
    with pytest.raises(ValueError):
        bad = data[("train", 0)]
        eng.bind_domain_data(0, "train", bad["uid"] + g["n_user"], bad["pid"], bad["domain"], bad["label"])   # ids beyond the table
    # the context still works after the refusals
    w0 = eng.weights.clone()
    eng.train_steps(0, n_steps=1)
    eng.sync()
    assert not torch.equal(eng.weights, w0)
    eng.close()


def test_outer_updates_on_empty_and_odd_lengths():
    """the outer-update entry points on n = 0 and on lengths that are no multiple of the 4-float vector width, in place
    (dst aliasing an operand, as `old += (new - old) * lr` does in the reference: domain_negotiation.py:118-123)."""
    from oracle import outer as oouter
    g, eng, model, data = make()
    dev = eng.device
    rs = np.random.RandomState(3)
    for n in (0, 1, 3, 5, 1023):
        a, b = rs.standard_normal(n).astype(F32), rs.standard_normal(n).astype(F32)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        eng.interp(ta, tb, ta, 0.1)                     # a += (b - a) * 0.1
        want = oouter.dn_update(a.copy(), b, 0.1)
        assert np.array_equal(bits(ta.cpu().numpy()), bits(want)), n
        tm = torch.empty(n, device=dev)
        eng.merge(tm, ta, tb, "times")
        assert np.array_equal(bits(tm.cpu().numpy()), bits(oouter.merge(want, b, "times"))), n
        ts = torch.empty(n, device=dev)
        eng.sub(ts, tb, ta)
        assert np.array_equal(bits(ts.cpu().numpy()), bits((b - want).astype(F32))), n
    eng.close()


def test_largest_batch_of_the_context_with_a_one_row_remainder():
    """max_batch = 8,192 (BASELINE configs[4]'s batch) on a domain of 8,193 rows: a full batch of the largest size and a
    last batch of one row, value-level against the oracle."""
    from test_gpu_parity import assert_adam_close
    g, eng, model, data = make(batch=8192, scale=0.6)
    d = max(range(g["n_domain"]), key=lambda k: data[("train", k)]["uid"].shape[0])
    n = data[("train", d)]["uid"].shape[0]
    if n < 8193:
        pytest.skip("largest synthetic domain has only %d rows" % n)
    c = {k: v[:8193] for k, v in data[("train", d)].items()}
    eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    perm = orng.shuffle_perm(8193, 10000, seed=9)
    losses = torch.zeros(2, device=eng.device)
    assert eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), lr=1e-3, loss_out=losses) == 2
    want = model.train_pass(c, perm, 8192)
    np.testing.assert_allclose(losses.cpu().numpy(), np.asarray(want, F32), rtol=3e-5, atol=1e-6)
    got = eng.unpack(eng.get_weights())
    for name in ("W0", "W1", "W2", "wo", "domain_emb"):
        assert_adam_close(got[name], model.params[name], 2, 1e-3, name, max_frac=2e-3)
    eng.close()


def test_pregather_hint_edge_cases():
    """mamdr_pregather_passes is a HINT (include/mamdr_hip.h): more than 16 passes (the later ones gather themselves),
    calls that skip hinted passes, a call that was never hinted (drops the hint, gathers as before), a window of a
    pass (`pass_rows`), step-at-a-time calls on a hinted pass, an empty hint, an unbound domain in the list (error,
    nothing changed) -- weights and both Adam slots bit-identical to the same calls without any hint."""
    out = {}
    for mode in ("hint", "plain"):
        g, eng, model, data = make(batch=256, scale=0.1)
        assert int(eng.lib.mamdr_step_path(eng.ctx, 256)) == 1
        D = g["n_domain"]
        n = [eng.n_rows(d, "train") for d in range(D)]
        perms = {}
        for k in range(24):
            d = k % D
            perms[k] = (d, torch.from_numpy(orng.shuffle_perm(n[d], 10000, seed=100 + k)).to(eng.device))
        hint = mode == "hint"
        if hint:
            eng.pregather([perms[k] for k in range(24)])             # 24 listed: the first 16 are gathered
        for k in (0, 1, 3, 4, 15, 16, 20):                            # skips 2, 5..14; 16 and 20 gather themselves
            eng.train_steps(perms[k][0], perm=perms[k][1], lr=1e-3)
        hits0 = int(eng.lib.mamdr_pregather_hits(eng.ctx))
        if hint:
            assert hits0 == 5, hits0
            eng.pregather([perms[k] for k in (21, 22, 23)])
        eng.train_steps(perms[5][0], perm=perms[5][1], lr=1e-3)       # not in the hint: drops it
        eng.train_steps(perms[21][0], perm=perms[21][1], lr=1e-3)     # ... so this one gathers itself too
        if hint:
            assert int(eng.lib.mamdr_pregather_hits(eng.ctx)) == hits0
            w = 300                                                   # a window of the pass, then step-at-a-time calls
            dbig = max(range(D), key=lambda d: n[d])
            pw = torch.from_numpy(orng.shuffle_perm(w, 10000, seed=7)).to(eng.device)
            pb = torch.from_numpy(orng.shuffle_perm(n[dbig], 10000, seed=8)).to(eng.device)
            eng.pregather([(dbig, pw, w), (dbig, pb)])
        else:
            w = 300
            dbig = max(range(D), key=lambda d: n[d])
            pw = torch.from_numpy(orng.shuffle_perm(w, 10000, seed=7)).to(eng.device)
            pb = torch.from_numpy(orng.shuffle_perm(n[dbig], 10000, seed=8)).to(eng.device)
        eng.train_steps(dbig, perm=pw, lr=1e-3, pass_rows=w)
        for s in range(min(3, -(-n[dbig] // 256))):
            eng.train_steps(dbig, perm=pb, first_step=s, n_steps=1, lr=1e-3)
        if hint:
            assert int(eng.lib.mamdr_pregather_hits(eng.ctx)) == hits0 + 1 + min(3, -(-n[dbig] // 256))
            eng.pregather([])                                         # forget it
            with pytest.raises(Exception):
                eng.pregather([(D + 3, None)])                        # unknown domain
        eng.train_steps(0, lr=1e-3)                                   # file order, no permutation
        out[mode] = (eng.weights.cpu().numpy().copy(), eng.adam_m.cpu().numpy().copy(), eng.adam_v.cpu().numpy().copy())
        eng.close()
    for a, b in zip(out["hint"], out["plain"]):
        assert np.isfinite(a).all() and np.array_equal(bits(a), bits(b))
