"""GPU parity of the mlp / wdl / deepfm towers with a `hidden_dim` OTHER than the reference configs' [256, 128, 64]
(model_zoo/DeepCTR/deepctr.py:26-32,36-38,118-136 pass any list through as deepctr's `dnn_hidden_units`).  The step
kernels are built for that one shape; every other one runs on the generic-layer engine (kinds MAMDR_GRAPH_MLP / _WDL /
_DEEPFM, round 5) against oracle/tower.py generalised to 1 .. 4 hidden layers (its gradients held to float64 autograd
in tests/test_oracle_crosscheck.py).  Bars as for the other towers: loss 2e-6, one-step gradients rtol 2e-4, an Adam pass
of 20+ steps within rounding-level displacement error, evaluation loss / AUC, and the registry's routing through run.py.
"""
import copy
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import auc as oauc          # noqa: E402
from oracle import rng as orng          # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(128, 64), (256, 128, 64, 64), (64,), (128, 128, 64)]


def make_problem(kind, hidden, batch=256, dropout=0.5, scale=0.05, seed=7, emb_trainable=False):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import graph_engine, synthetic
    g = synthetic.generate("taobao10", batch_size=batch, seed=seed, scale=scale)
    D = g["n_domain"]
    rs = np.random.RandomState(seed)
    params = otower.init_params(rs, g["n_user"], g["n_item"], D, hidden=hidden)
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"].copy(), g["tables"]["item_emb"].copy()
    params["domain_emb"] = (rs.standard_normal(params["domain_emb"].shape) * 0.05).astype(F32)
    for n in ["b%d" % l for l in range(len(hidden))] + ["lin_domain", "lin_user", "lin_item"]:
        params[n] = (rs.standard_normal(params[n].shape) * 0.05).astype(F32)
    params["gb"] = np.array([0.1], F32)
    if not emb_trainable:           # frozen linear tables stay at their zero initialisation (deepctr: same feature column)
        params["lin_user"][...] = 0
        params["lin_item"][...] = 0
    eng = graph_engine.GraphEngine(kind, g["n_user"], g["n_item"], D, batch, hidden, (), dropout=dropout,
                                   emb_trainable=emb_trainable)
    if not emb_trainable:
        eng.bind_table("user_emb", params["user_emb"])
        eng.bind_table("item_emb", params["item_emb"])
    for split in ("train", "val"):
        for d in range(D):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    model = otower.OracleModel({k: v.copy() for k, v in params.items()}, emb_trainable=emb_trainable, dropout=dropout, lr=1e-3,
                               hidden=hidden, dropout_seed=eng.dropout_seed, tower=kind)
    # the flat layout IS the oracle's (Keras trainable_weights order)
    assert list(eng.segments) == list(model.names), (list(eng.segments), model.names)
    eng.set_weights(eng.pack(params))
    return g, eng, model


@pytest.mark.parametrize("hidden", SHAPES[:3], ids=lambda h: "x".join(map(str, h)))
@pytest.mark.parametrize("kind,emb_trainable", [("mlp", False), ("wdl", False), ("deepfm", False), ("deepfm", True), ("mlp", True)])
def test_one_step_gradients_match_oracle(kind, emb_trainable, hidden):
    g, eng, model = make_problem(kind, hidden, scale=0.1 if emb_trainable else 0.05, emb_trainable=emb_trainable)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = {k: v.copy() for k, v in g["data"]["train"][d].items()}
    cols["domain"] = (np.arange(cols["domain"].shape[0]) % 3).astype(np.int32)      # mixed domain ids in one batch
    eng.bind_domain_data(d, "train", cols["uid"], cols["pid"], cols["domain"], cols["label"])
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=11)
    perm_t = torch.from_numpy(perm).to(eng.device)
    n_step = -(-n // 256)
    for step in (0, n_step - 1):            # a full batch and the pass's short last one
        idx = perm[step * 256:(step + 1) * 256]
        masks = otower.train_masks(model.seed, model.step, len(idx), hidden, 0.5)
        loss, grads, _ = otower.loss_and_grads(model.params, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                               cols["label"][idx], masks, 0.5, emb_trainable, model.frozen_sumsq(), model.deepfm)
        loss_t = torch.zeros(1, device=eng.device)
        w0 = eng.get_weights()
        eng.train_steps(d, perm=perm_t, first_step=step, n_steps=1, lr=1.0, optimizer="sgd", loss_out=loss_t)
        got = eng.unpack(w0 - eng.get_weights())
        eng.set_weights(w0)
        model.step += 1
        assert abs(float(loss_t.cpu()[0]) - float(loss)) < 2e-6 * max(1.0, abs(float(loss)))
        assert sorted(grads) == sorted(model.names)
        for name, want in grads.items():
            want = np.asarray(want).ravel()
            floor = 4e-8 if name in ("user_emb", "item_emb") else 1.5e-8       # read back as w0 - (w0 - g): the weights' ulp
            np.testing.assert_allclose(got[name], want, rtol=2e-4, atol=max(2e-6 * max(np.abs(want).max(), 1e-3), floor),
                                       err_msg=name)
    eng.close()


@pytest.mark.parametrize("hidden", SHAPES, ids=lambda h: "x".join(map(str, h)))
@pytest.mark.parametrize("kind", ["mlp", "wdl", "deepfm"])
def test_adam_pass_and_evaluation_match_oracle(kind, hidden):
    """one pass of TF1 Adam steps (dropout on) over the largest domain, then evaluation of another domain's val split: the
    displacement of every tensor against the oracle's, loss and AUC-500."""
    g, eng, model = make_problem(kind, hidden, scale=0.2)
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    c = g["data"]["train"][d]
    n = c["uid"].shape[0]
    perm = orng.shuffle_perm(n, 10000, seed=5)
    w0 = eng.unpack(eng.get_weights())
    n_steps = eng.train_steps(d, perm=torch.from_numpy(perm).to(eng.device), lr=1e-3)
    assert n_steps == -(-n // 256) and n_steps >= 20
    model.train_pass(c, perm, 256)
    got = eng.unpack(eng.get_weights())
    for name in model.names:
        a, o, s = np.asarray(got[name]).ravel(), model.params[name].ravel(), np.asarray(w0[name]).ravel()
        nrm = float(np.linalg.norm(o - s))
        err = float(np.linalg.norm(a - o))
        # (Adam turns rounding-level gradient differences of near-zero gradients into steps of +- lr: a few percent of the
        # displacement's norm for the small tensors; the one-step gradients above are what pins the arithmetic)
        assert err <= 3e-2 * nrm + 1e-6 * float(np.linalg.norm(s)) + 1e-7, (name, err, nrm)
    for dv in (d, (d + 1) % 10):
        loss_g, auc_g = eng.evaluate(dv, "val")
        loss_o, preds = model.evaluate(g["data"]["val"][dv], 256)
        auc_o = float(oauc.auc500(g["data"]["val"][dv]["label"], preds, 256))
        assert abs(loss_g - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o))), (loss_g, float(loss_o))
        assert abs(auc_g - auc_o) <= 1e-3, (auc_g, auc_o)
    eng.close()


@pytest.mark.parametrize("name,hidden", [("mlp_meta_mamdr_finetune", [128, 64]), ("deepfm_meta_domain_negotiation", [256, 128, 64, 64]),
                                         ("wdl", [64])])
def test_run_config_with_other_hidden_dims(tmp_path, name, hidden):
    """run.py's entry with a hidden_dim the step kernels are not built for: the registry routes the tower onto the
    generic-layer engine (round 4 raised ValueError here) and the wrappers run on it unchanged."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import cli, graph_engine
    with open(os.path.join(ROOT, "config", "Taobao-10", "deepctr_DN+DR.json")) as f:
        cfg = copy.deepcopy(json.load(f))
    cfg["model"].update(name=name, hidden_dim=hidden)
    cfg["train"].update(epoch=3, patience=1, sample_num=2, meta_learning_rate=0.5,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    cfg["dataset"].update(batch_size=256, synthetic="taobao10", synthetic_scale=0.1)
    built = []
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg, on_model=built.append)
    assert isinstance(built[0].model, graph_engine.GraphEngine) and built[0].model.kind == name.split("_")[0]
    assert len(domain_auc) == 10 and np.isfinite(avg_loss)
    assert avg_auc > 0.6, (name, avg_auc)          # the tower learns the planted signal
