"""GPU parity of deepctr's NFM and PNN towers (model_zoo/DeepCTR/deepctr.py:33-35,44-46): the
generic-layer HIP engine (`mamdr_graph_*`, kinds MAMDR_GRAPH_NFM / MAMDR_GRAPH_PNN) against oracle/fmnets.py (parity
unpinned: deepctr is not in the reference tree; the oracle's gradients are held to float64 autograd in
tests/test_oracle_crosscheck.py).  Bars as for the other towers: one-step gradients rtol 2e-4, loss 2e-6, evaluation
predictions rtol 2e-5, per-domain AUC within 1e-3 after a Domain Negotiation run on the tower.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_jobs                      # noqa: E402
from ensemble import Ensemble, TWIN_SEEDS      # noqa: E402
from oracle import auc as oauc          # noqa: E402
from oracle import fmnets as ofm        # noqa: E402
from oracle import loops as oloops      # noqa: E402
from oracle import rng as orng          # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32
HIDDEN = (256, 128, 64)


def make_problem(kind, batch=256, dropout=0.5, scale=0.05, seed=7, emb_trainable=False, uncertainty=False):
    """kind "pnn@step" / "nfm@step": the tower on the STEP kernels (mamdr_create, MAMDR_TOWER_PNN / _NFM: k_tower4's FM
    instances; round 4) instead of the generic-layer engine -- the same oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import engine, graph_engine, synthetic
    step = kind.endswith("@step")
    kind = kind.split("@")[0]
    g = synthetic.generate("taobao10", batch_size=batch, seed=seed, scale=scale)
    D = g["n_domain"]
    rs = np.random.RandomState(seed)
    if kind in ("ccpm", "autoint"):
        params = ofm.init_params_conv(rs, kind, g["n_user"], g["n_item"], D, hidden=HIDDEN)
        if kind == "ccpm":
            params["conv1_b"] = (rs.standard_normal(4) * 0.1).astype(F32)
            params["conv2_b"] = (rs.standard_normal(4) * 0.1).astype(F32)
        else:
            for l in range(3):          # larger attention kernels: the softmax is off its uniform point
                params["att%d_w" % l] = (params["att%d_w" % l] * 4).astype(F32)
    else:
        params = ofm.init_params(rs, kind, g["n_user"], g["n_item"], D, hidden=HIDDEN)
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"].copy(), g["tables"]["item_emb"].copy()
    params["domain_emb"] = (rs.standard_normal(params["domain_emb"].shape) * 0.05).astype(F32)
    for n in ("b0", "b1", "b2", "lin_domain", "lin_user", "lin_item"):
        params[n] = (rs.standard_normal(params[n].shape) * 0.05).astype(F32)
    params["gb"] = np.array([0.1], F32)
    if not emb_trainable:           # frozen linear tables stay at their zero initialisation (deepctr: same feature column)
        params["lin_user"][...] = 0
        params["lin_item"][...] = 0
    if uncertainty:            # distinct per-domain scales around the initial value 1 (weighted_loss.py:23-28)
        params["log_var"] = (1.0 + rs.uniform(-0.3, 0.3, D)).astype(F32)
    if step:
        eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=dropout, emb_trainable=emb_trainable, tower=kind,
                                 uncertainty_weight=uncertainty)
    else:
        eng = graph_engine.GraphEngine(kind, g["n_user"], g["n_item"], D, batch, HIDDEN, (), dropout=dropout,
                                       emb_trainable=emb_trainable, uncertainty_weight=uncertainty)
    if not emb_trainable:
        eng.bind_table("user_emb", params["user_emb"])
        eng.bind_table("item_emb", params["item_emb"])
    for split in ("train", "val", "test"):
        for d in range(D):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    names = list(ofm.ccpm_param_names(emb_trainable, uncertainty) if kind == "ccpm" else
                 (ofm.autoint_param_names(emb_trainable, uncertainty) if kind == "autoint"
                  else ofm.param_names(kind, emb_trainable, uncertainty)))
    if step:            # (the step kernels keep the inner products' rows of W0 as a segment of their own, behind the rest)
        assert sorted(set(eng.segments) - {"W0x"}) == sorted(names), (list(eng.segments), names)
    else:
        assert list(eng.segments) == names, (list(eng.segments), names)
    eng.set_weights(eng.pack(params))
    model = ofm.OracleNet({k: v.copy() for k, v in params.items()}, kind, emb_trainable=emb_trainable, dropout=dropout,
                          lr=1e-3, hidden=HIDDEN, dropout_seed=eng.dropout_seed, **({"uncertainty": True} if uncertainty else {}))
    return g, eng, model


@pytest.mark.parametrize("kind", ["nfm", "nfm@step", "pnn", "pnn@step", "ccpm", "autoint"])
@pytest.mark.parametrize("emb_trainable", [False, True])
def test_one_step_gradients_match_oracle(kind, emb_trainable):
    g, eng, model = make_problem(kind, dropout=0.5, scale=0.1 if emb_trainable else 0.05, emb_trainable=emb_trainable)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = {k: v.copy() for k, v in g["data"]["train"][d].items()}
    cols["domain"] = (np.arange(cols["domain"].shape[0]) % 3).astype(np.int32)      # mixed domain ids in one batch
    eng.bind_domain_data(d, "train", cols["uid"], cols["pid"], cols["domain"], cols["label"])
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=11)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // 256)
    for step in (0, n_step - 1):
        idx = perm[step * 256:(step + 1) * 256]
        masks = otower.train_masks(model.seed, model.step, len(idx), HIDDEN, 0.5)
        fn = ofm.loss_and_grads_conv if kind in ("ccpm", "autoint") else ofm.loss_and_grads
        loss, grads, _ = fn(model.params, kind.split("@")[0], cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                            cols["label"][idx], masks, 0.5, emb_trainable, model.frozen_sumsq())
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = eng.unpack(w0 - eng.get_weights())
        eng.set_weights(w0)
        model.step += 1
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        for name, want in grads.items():
            want = want.ravel()
            floor = 4e-8 if name in ("user_emb", "item_emb") else 1.5e-8       # read back as w0 - (w0 - g): the weights' ulp
            np.testing.assert_allclose(got[name], want, rtol=2e-4, atol=max(2e-6 * max(np.abs(want).max(), 1e-3), floor),
                                       err_msg=name)
    eng.close()


@pytest.mark.parametrize("kind", ["pnn@step", "nfm@step"])
@pytest.mark.parametrize("emb_trainable", [False, True])
@pytest.mark.parametrize("batch", [1024, 2048])
def test_step_kernel_towers_at_config_batch_sizes(kind, emb_trainable, batch):
    """PNN / NFM on the step kernels at the reference configs' batch size (1,024: one four-row tile per CU, the W1 image
    in LDS) and at the largest batch this path takes (2,048: two rounds of workgroups, the streaming instance with the
    transposed W1 / W2 copies kept current by k_update): one-step gradients of every tensor and the loss against the
    oracle, on a full and on the pass's last (partial) batch, then three Adam steps (copies stale after the first)."""
    g, eng, model = make_problem(kind, batch=batch, dropout=0.5, scale=0.5, emb_trainable=emb_trainable)
    base = kind.split("@")[0]
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    assert n > 2 * batch
    perm = orng.shuffle_perm(n, 10000, seed=13)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // batch)
    for step in (0, n_step - 1):
        idx = perm[step * batch:(step + 1) * batch]
        masks = otower.train_masks(model.seed, model.step, len(idx), HIDDEN, 0.5)
        loss, grads, _ = ofm.loss_and_grads(model.params, base, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                            cols["label"][idx], masks, 0.5, emb_trainable, model.frozen_sumsq())
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = eng.unpack(w0 - eng.get_weights())
        eng.set_weights(w0)
        model.step += 1
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        for name, want in grads.items():
            want = want.ravel()
            floor = 4e-8 if name in ("user_emb", "item_emb") else 1.5e-8
            np.testing.assert_allclose(got[name], want, rtol=2e-4, atol=max(2e-6 * max(np.abs(want).max(), 1e-3), floor),
                                       err_msg=name)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=3, lr=1e-3)
    want_losses = model.train_pass(cols, perm, batch, max_steps=3)
    got = eng.unpack(eng.get_weights())
    from test_gpu_parity import assert_adam_close      # (Adam normalises rounding-level gradients to steps of ~lr)
    for name in model.names:
        assert_adam_close(got[name], model.params[name], 3, 1e-3, name, max_frac=2e-3)
    assert np.isfinite(np.array(want_losses, F32)).all()
    eng.close()


@pytest.mark.parametrize("kind", ["pnn@step", "nfm@step", "pnn", "nfm", "ccpm", "autoint"])
def test_uncertainty_weighted_step_on_the_step_kernel_towers(kind):
    """run.py:49-50 wraps ANY tower in the weighted loss (weighted_loss.py:30-43); on the step kernels' PNN / NFM (round 4)
    as on mlp / deepfm, and on the generic-layer engine's single-output towers (CCPM, AutoInt, the PNN / NFM twins): every
    gradient scaled by 1 / var_d^2, d loss / d var_d, zero gradient for the other domains' scalars, a few Adam steps
    (without a loss output: the scalar's gradient must not depend on one), evaluation unweighted."""
    g, eng, model = make_problem(kind, batch=256, dropout=0.5, uncertainty=True)
    base = kind.split("@")[0]
    assert eng.segments["log_var"][1] == 10 and model.names[-1] == "log_var"
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=11)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // 256)
    for step in (0, n_step - 1):
        idx = perm[step * 256:(step + 1) * 256]
        masks = otower.train_masks(model.seed, model.step, len(idx), HIDDEN, 0.5)
        fn = ofm.loss_and_grads_conv if base in ("ccpm", "autoint") else ofm.loss_and_grads
        loss, grads, _ = fn(model.params, base, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                            cols["label"][idx], masks, 0.5, False, model.frozen_sumsq(), True)
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = eng.unpack(w0 - eng.get_weights())
        eng.set_weights(w0)
        model.step += 1
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        for name, want in grads.items():
            want = want.ravel()
            np.testing.assert_allclose(got[name], want, rtol=2e-4, atol=max(2e-6 * max(np.abs(want).max(), 1e-3), 1e-7),
                                       err_msg=name)
        lv = got["log_var"]
        assert lv[d] != 0 and not np.delete(lv, d).any()
    k = min(3, n_step)
    eng.train_steps(d, perm=perm_t, first_step=0, n_steps=k, lr=1e-3)
    model.train_pass(cols, perm, 256, max_steps=k)
    got = eng.unpack(eng.get_weights())
    from test_gpu_parity import assert_adam_close
    for name in model.names:
        assert_adam_close(got[name], model.params[name], k, 1e-3, name, max_frac=2e-3)
    loss_g, _ = eng.evaluate(d, "val")
    loss_o, _ = model.evaluate(g["data"]["val"][d], eng.eval_batch)
    assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    eng.close()


@pytest.mark.parametrize("kind", ["nfm", "nfm@step", "pnn", "pnn@step", "ccpm", "autoint"])
def test_adam_pass_and_eval(kind):
    g, eng, model = make_problem(kind, dropout=0.5)
    d = 9
    cols = g["data"]["train"][d]
    perm = orng.shuffle_perm(cols["uid"].shape[0], 10000, seed=3)
    n_steps = min(6, -(-perm.shape[0] // 256))
    losses_t = torch.zeros(n_steps, device=eng.device)
    eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), n_steps=n_steps, lr=1e-3, loss_out=losses_t)
    want_losses = model.train_pass(cols, perm, 256, max_steps=n_steps)
    np.testing.assert_allclose(losses_t.cpu().numpy(), np.array(want_losses, F32), rtol=2e-5, atol=2e-6)
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        diff = np.abs(got[name].reshape(model.params[name].shape) - model.params[name]).max()
        assert diff < 0.05 * n_steps * 1e-3, (name, diff)
    for dd in (0, 5):
        c = g["data"]["val"][dd]
        loss, auc, hist, preds = eng.evaluate(dd, "val", want_preds=True)
        loss_o, preds_o = model.evaluate(c, 256)
        np.testing.assert_allclose(preds, preds_o, rtol=3e-3, atol=3e-5)
        assert abs(loss - float(loss_o)) < 2e-3 * max(1.0, abs(float(loss_o)))
    eng.close()


def test_eval_predictions_at_equal_weights():
    for kind in ("nfm", "nfm@step", "pnn", "pnn@step", "ccpm", "autoint"):
        g, eng, model = make_problem(kind)
        for d in (1, 5):
            c = g["data"]["test"][d]
            loss, auc, hist, preds = eng.evaluate(d, "test", want_preds=True)
            loss_o, preds_o = model.evaluate(c, 256)
            np.testing.assert_allclose(preds, preds_o, rtol=2e-5, atol=2e-7)
            assert abs(loss - float(loss_o)) < 2e-6 * max(1.0, abs(float(loss_o)))
            assert abs(auc - float(oauc.auc500(c["label"], preds_o, 256))) < 1e-4
        eng.close()


@pytest.mark.parametrize("kind", ["nfm", "nfm@step", "pnn", "pnn@step", "ccpm", "autoint"])
@pytest.mark.parametrize("emb_trainable", [False, True])
def test_accumulate_steps_match_oracle(kind, emb_trainable):
    """MAMDR_OPT_ACCUMULATE on the generic-layer engine (the meta passes of MAML / MLDG / PCGrad, maml.py:107-109,196-229):
    dropout off, weights and Adam slots bit-unchanged, the gradients of the steps ADDED to the bound accumulator; on a
    window of the split (`mamdr_graph_train_steps_n`: the meta-train / meta-val sub-datasets, maml.py:300-330)."""
    g, eng, model = make_problem(kind, emb_trainable=emb_trainable)
    d = 1
    c = g["data"]["train"][d]
    n = c["uid"].shape[0]
    window = min(n, 600)
    perm = orng.shuffle_perm(n, 10000, seed=33)[:window].copy()
    acc = eng.new_vector()
    acc.fill_(0.25)                         # added to, not overwritten
    eng.bind_accumulator(acc)
    w0, m0, v0 = eng.weights.clone(), eng.adam_m.clone(), eng.adam_v.clone()
    n_steps = eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), optimizer="accumulate", pass_rows=window)
    assert n_steps == -(-window // 256)
    assert torch.equal(eng.weights, w0) and torch.equal(eng.adam_m, m0) and torch.equal(eng.adam_v, v0)
    acc_o = np.full(eng.n_params, 0.25, F32)
    flat_names = list(eng.segments)
    want = {}
    tmp = np.zeros(sum(model.params[k].size for k in model.names), F32)
    model.train_pass(c, perm, 256, accumulate_into=tmp)
    off = 0
    for k in model.names:
        want[k] = tmp[off:off + model.params[k].size]
        off += model.params[k].size
    got = eng.unpack(acc)
    for k in model.names:
        gk = np.asarray(got[k], F32).ravel() - F32(0.25)
        scale = max(float(np.abs(want[k]).max()), 1e-6)
        if kind == "ccpm":      # a maximum over the fields within rounding of a tie routes ONE batch row's gradient through another
            bad = np.abs(gk - want[k]) > 3e-4 * np.abs(want[k]) + 3e-4 * scale + 6e-8     # field on the two sides: a few rows' worth
            assert bad.sum() <= max(2e-3 * bad.size, 4 * 128) and np.abs(gk - want[k]).max() <= 0.05 * scale + 1e-5, (k, bad.mean())
        else:
            np.testing.assert_allclose(gk, want[k], rtol=3e-4, atol=3e-4 * scale + 6e-8, err_msg=k)
    assert flat_names == list(model.names) or kind.endswith("@step")
    eng.close()


@pytest.mark.parametrize("kind", ["nfm", "nfm@step", "pnn", "pnn@step", "ccpm", "autoint"])
def test_domain_negotiation_auc_parity(kind):
    """the meta wrappers run on these towers too (deepctr.py's registry is orthogonal to run.py's wrappers): five
    Domain Negotiation epochs (domain_negotiation.py:49-88) on 4 domains, same order / shuffles / masks on both sides.
    After the FIRST epoch -- before the amplification -- every domain is held to north_star's plain 1e-3.  At the end of the
    five epochs fp32 training of these towers at this learning rate is chaotic (an oracle twin whose weights differ by one
    rounding ends 8e-4 .. 5e-3 per domain away for PNN, up to 3e-3 for NFM, up to 1e-2 for CCPM -- an argmax over the fields
    behind every unit): there the HIP run must not be an outlier of an ensemble of six oracle runs (round 6, tests/ensemble.py;
    rounds 3 - 5 asserted 1e-3 plus twice one twin's shift).  One-step gradients, Adam passes, evaluation (above) and the
    teacher-forced epochs (below) are what pins the arithmetic."""
    from mamdr_amd import meta
    g, eng, model = make_problem(kind, scale=0.15)
    D = 4
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]

    def make_perm_fn():
        k = [0]

        def f(d):
            k[0] += 1
            return orng.shuffle_perm(sizes[d], 10000, seed=700 + k[0])
        return f
    seqs = ([2, 0, 3, 1], [1, 3, 0, 2], [0, 1, 2, 3], [3, 1, 2, 0], [2, 3, 1, 0])
    LR = 5e-3           # (NFM's inputs are products of 0.1-scale embeddings: at 1e-3 five epochs leave it near AUC 0.57)
    theta0 = model.get_flat().copy()
    params0 = {k: v.copy() for k, v in model.params.items()}

    def oracle_run(theta_start, shake_seed=None):
        twin = ofm.OracleNet({k: v.copy() for k, v in params0.items()}, kind.split("@")[0], dropout=0.5, lr=LR, hidden=HIDDEN,
                             dropout_seed=eng.dropout_seed)
        if shake_seed is not None:          # a perturbed twin rounds differently in every pass (tests/ensemble.py)
            twin = oracle_jobs.PassShaker(twin, np.random.RandomState(shake_seed), 2e-7)
        theta = theta_start.copy()
        pf, traces = make_perm_fn(), []

        def val_aucs():
            twin.set_flat(theta)
            out = []
            for d in range(D):
                _, preds = twin.evaluate(g["data"]["val"][d], 256)
                out.append(float(oauc.auc500(g["data"]["val"][d]["label"], preds, 256)))
            return out
        first = None
        for seq in seqs:
            traces += oloops.dn_epoch(twin, theta, g["data"]["train"], seq, pf, 256, 0.5)
            if first is None:
                first = val_aucs()
        return val_aucs(), traces, first
    auc_o, tr_o, first_o = oracle_run(theta0)
    twins = [oracle_run(oracle_jobs.perturbed(theta0, np.random.RandomState(sd), 2e-7), shake_seed=sd + 7919)[0] for sd in TWIN_SEEDS]
    theta_g = eng.get_weights()
    pf_g, tr_g = make_perm_fn(), []
    for k, seq in enumerate(seqs):
        tr_g += meta.dn_epoch(eng, theta_g, seq, pf_g, 256, lr=LR, meta_lr=0.5)
        if k == 0:
            # after ONE epoch the rounding-level differences have not been amplified yet: the plain 1e-3 bar, every domain
            eng.set_weights(theta_g)
            for d in range(D):
                _, a1 = eng.evaluate(d, "val")
                print("%s domain %d after the first epoch: AUC hip %.5f oracle %.5f" % (kind, d, a1, first_o[d]))
                assert abs(a1 - first_o[d]) <= 1e-3, (d, a1, first_o[d])
    assert tr_o == tr_g
    eng.set_weights(theta_g)
    got = []
    for d in range(D):
        _, auc_g = eng.evaluate(d, "val")
        got.append(auc_g)
        print("%s domain %d: AUC hip %.5f oracle %.5f (perturbed oracles %s)"
              % (kind, d, auc_g, auc_o[d], " ".join("%.5f" % t[d] for t in twins)))
    # end of training: not an outlier of the ensemble of six oracle runs (no factor on a single draw; tests/ensemble.py)
    ens = Ensemble([None] * (1 + len(twins)))
    ens.check("val", dict(enumerate(got)), [dict(enumerate(m)) for m in [auc_o] + twins])
    ens.aggregate(kind)
    assert np.mean(auc_o) > 0.57
    eng.close()


@pytest.mark.parametrize("kind", ["nfm", "pnn", "ccpm", "autoint"])
def test_domain_negotiation_teacher_forced(kind):
    """the chaos-free counterpart of test_domain_negotiation_auc_parity (round 6, tests/teacher.py) on the generic-layer engine:
    the same five Domain Negotiation epochs on 4 domains at the same learning rate, but every pass starts from the ORACLE's
    state (weights, Adam slots, step count with TF's beta powers, dropout position) and is compared pass by pass -- per-step
    losses and end state -- with the bars of tests/test_gpu_teacher.py and no self-divergence term.  (These towers at 5e-3
    are the most chaotic training in the suite: the free-running test above needs up to 1e-2 of oracle self-divergence; one
    pass from a common state does not.)"""
    import teacher
    from teacher_bars import BARS
    g, eng, model = make_problem(kind, scale=0.15)
    D = 4
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
    k = [0]

    def perm_fn(d):
        k[0] += 1
        return orng.shuffle_perm(sizes[d], 10000, seed=700 + k[0])
    seqs = ([2, 0, 3, 1], [1, 3, 0, 2], [0, 1, 2, 3], [3, 1, 2, 0], [2, 3, 1, 0])
    LR = 5e-3
    model.lr = LR
    bars = dict(BARS, frac=2e-3)
    # (AutoInt -- attention kernels x 4, lr 5e-3 -- amplifies fastest: one 19-step pass from a common state ends 2.3e-4 apart in
    # the loss, just over the bar; like Amazon-6's DeepFM it is forced in 8-step chunks.  The others: whole passes.)
    ls = teacher.LockStep(model, model, eng, g["data"]["train"], LR, bars, chunk=8 if kind == "autoint" else 1 << 20)
    theta = model.get_flat().copy()
    trace = []
    for seq in seqs:
        trace += oloops.dn_epoch(ls, theta, g["data"]["train"], seq, perm_fn, 256, 0.5)
    out = ls.summary()
    print("%s DN teacher-forced: %d passes / %d steps; worst first-step loss rel %.1e, any step %.1e; weights frac %.1e max %.3f k lr "
          "median %.5f k lr; slots %.1e / %.1e" % (kind, out["passes"], out["steps"], out["loss_first"], out["loss_rel"], out["frac"],
                                                   out["max_klr"], out["med_klr"], out["m_rel"], out["v_rel"]))
    for v in ls.bad[:10]:
        print("  VIOLATION", v)
    assert not ls.bad, "%d violations (first: %r)" % (len(ls.bad), ls.bad[0])
    assert out["passes"] == len(trace) == 20 and out["steps"] == sum(t[2] for t in trace)
    eng.close()
