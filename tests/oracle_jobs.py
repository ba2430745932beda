"""The heavy runs of the numpy oracle that the GPU parity tests compare with, as jobs of a worker-process pool
(test infrastructure; nothing under mamdr_amd/ imports this).

Round 4's GPU suite spent ~500 of its 606 s inside the oracle, one case after the other, while the GPU idled.  The jobs
below are pure functions of their arguments (seeded inputs, no GPU): `conftest.pytest_collection_finish` starts the
jobs of every SELECTED test (marker `oracle_job(name, **kwargs)`) in spawned worker processes when the session starts,
and a test collects its result with `result(name, **kwargs)` after its HIP side has run.  The oracle still runs LIVE in
every session -- no stored outputs that could go stale -- only concurrently (tests/oracle_pool.py: one worker per job,
each on its own block of physical cores).  `problem_*` helpers build the seeded
inputs; the tests call the same helpers for the HIP side, so both sides see the same tensors by construction.
"""
import os
import time

import numpy as np

F32 = np.float32
SHUFFLE_SEED = 0x5eed
DROPOUT_SEED = 1024          # mamdr_amd.engine.TowerEngine's default (asserted by the tests)

from oracle_pool import result, shutdown  # noqa: E402,F401  (the pool itself: tests/oracle_pool.py)


def start(keys):
    import oracle_pool
    oracle_pool.start(keys, COST)


# ---------------------------------------------------------------------------------------------- MAMDR epochs, frozen tables
def problem_fullsize(shape, batch, epochs, seed=123):
    """BASELINE.json configs[1] / configs[3]: every domain, full tables, full splits, the config's sample_num 5 +
    add_query_domain + shuffled sequence (config/Taobao-10/deepctr_DN+DR.json), phi_d = a second random initialisation of
    the whole model (mamdr.py:31-33)."""
    from mamdr_amd import plan as mplan
    from mamdr_amd import synthetic
    from oracle import tower as otower
    g = synthetic.generate(shape, batch_size=batch, seed=seed)
    params = otower.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], g["n_domain"])
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
    D = g["n_domain"]
    planner = mplan.EpochPlanner(range(D), 5, True, True, seed=123)
    plans = [planner.next_epoch() for _ in range(epochs)]
    names = otower.param_names(False)
    phis0 = [otower.flatten(otower.init_params(np.random.RandomState(2000 + d), 8, 8, D), names) for d in range(D)]
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
    return dict(g=g, params=params, plans=plans, names=names, phis0=phis0, sizes=sizes, D=D)


def perturbed(a, prs, rel):
    """a * (1 + rel * N(0, 1)) elementwise in fp32: rel = 2e-7 is a rounding-level change of every element."""
    return (a * (F32(1) + F32(rel) * prs.standard_normal(a.shape).astype(F32))).astype(F32)


def dump_root():
    """where the per-pass states of the teacher-forced tests are written: one directory per pytest session (conftest sets
    MAMDR_TEST_DUMP_ROOT and removes it when the session ends); the system's temporary directory otherwise."""
    import tempfile
    root = os.environ.get("MAMDR_TEST_DUMP_ROOT") or tempfile.gettempdir()
    os.makedirs(root, exist_ok=True)
    return root


class PassDump(object):
    """The oracle model behind a recorder (teacher-forced epochs).  Every `train_pass` of the loop is
    written out with the state it STARTED from -- weights, Adam m / v, Adam step count, position of the dropout stream --
    its per-step losses and the weights it ended with, as one record of a raw float32 file:
        record k = [w_start | m_start | v_start | w_end]   (4 P floats; P = the model's flat length, oracle order)
    and behind the last record [m_end | v_end] of the last pass (m / v at the end of pass k = m / v at the start of pass k + 1:
    nothing touches the slots between passes -- SURVEY A.5).  tests/test_gpu_teacher.py starts the HIP engine from every
    record and compares the pass it then runs: no chaos term, every pass of the epoch, every ragged last batch."""

    def __init__(self, model, tag):
        import tempfile
        self.model = model
        self.dir = tempfile.mkdtemp(prefix="mamdr_tf_%s_" % tag, dir=dump_root())
        self.path = os.path.join(self.dir, "passes.f32")
        self.f = open(self.path, "wb")
        self.meta = []          # per pass: (adam t at the start, dropout step at the start, n_steps)
        self.losses = []        # per pass: float32 array of the steps' total losses
        self.P = None

    def state(self):
        from oracle import tower as otower
        m = self.model
        return (m.get_flat(), otower.flatten(m.opt.m, m.names), otower.flatten(m.opt.v, m.names))

    def __getattr__(self, name):           # get_flat / set_flat / evaluate / params / ...: the model's own
        return getattr(self.model, name)

    def train_pass(self, data, perm, batch_size, max_steps=0, accumulate_into=None):
        w, m, v = self.state()
        self.P = w.size
        t0, s0 = int(self.model.opt.t), int(self.model.step)
        for a in (w, m, v):
            self.f.write(np.ascontiguousarray(a, F32).tobytes())
        losses = self.model.train_pass(data, perm, batch_size, max_steps, accumulate_into)
        self.f.write(np.ascontiguousarray(self.model.get_flat(), F32).tobytes())
        self.meta.append((t0, s0, len(losses)))
        self.losses.append(np.array([np.nan if l is None else l for l in losses], F32))
        return losses

    def close(self):
        _, m, v = self.state()
        for a in (m, v):
            self.f.write(np.ascontiguousarray(a, F32).tobytes())
        self.f.close()
        return dict(dir=self.dir, path=self.path, P=int(self.P), meta=self.meta, losses=self.losses)


def read_dump(dump):
    """-> (records [n_pass][4][P] (w_start, m_start, v_start, w_end), tail [2][P] (m, v after the last pass)) as memmaps."""
    n, P = len(dump["meta"]), dump["P"]
    raw = np.memmap(dump["path"], dtype=F32, mode="r")
    assert raw.size == (4 * n + 2) * P, (raw.size, n, P)
    return raw[:4 * n * P].reshape(n, 4, P), raw[4 * n * P:].reshape(2, P)


class PassShaker(object):
    """The oracle model of a perturbed TWIN: the live weights are changed by one fp32 rounding (relative `perturb` per element)
    after every training pass -- "another fp32 evaluation of this training" rounds differently in every pass, not only in the
    initial weights (run_pipeline's twins do the same on the engine level; tests/ensemble.py)."""

    def __init__(self, model, rs, perturb):
        self.model, self.rs, self.perturb = model, rs, perturb

    def __getattr__(self, name):
        return getattr(self.model, name)

    def train_pass(self, *a, **k):
        out = self.model.train_pass(*a, **k)
        self.model.set_flat(perturbed(self.model.get_flat(), self.rs, self.perturb))
        return out


def job_fullsize_mamdr(shape, batch, meta_lr, epochs, perturb=0.0, dump=False, pseed=99):
    """oracle/loops.mamdr_epoch (model_zoo/mamdr.py:41-108) x epochs, then every domain's validation AUC with the merged
    weights theta + phi_d (specific_base_model.py:64-97).  The per-pass shuffles are plan.PassShuffler's stream -- the
    one plan.EpochShuffles hands the HIP side.  perturb > 0: a perturbed TWIN (tests/ensemble.py) -- every trainable initial
    tensor (theta's and the phi_d's; not the frozen tables) and the live weights after every pass changed at rounding level."""
    from mamdr_amd import plan as mplan
    from oracle import auc as oauc
    from oracle import loops as oloops
    from oracle import outer as oouter
    from oracle import tower as otower
    pb = problem_fullsize(shape, batch, epochs)
    g, params = pb["g"], pb["params"]
    model = otower.OracleModel({k: (v if k in ("user_emb", "item_emb") else v.copy()) for k, v in params.items()},
                               dropout=0.5, lr=1e-3, dropout_seed=DROPOUT_SEED)
    theta = model.get_flat().copy()
    phis = [p.copy() for p in pb["phis0"]]
    rec = PassDump(model, "%s_bs%d" % (shape, batch)) if dump else None
    if perturb > 0:         # (`pseed` draws the twin: the K twins of an ensemble differ in it)
        prs = np.random.RandomState(pseed)
        theta = perturbed(theta, prs, perturb)
        phis = [perturbed(p, prs, perturb) for p in phis]
        assert rec is None
        rec = PassShaker(model, np.random.RandomState(pseed + 7919), perturb)
    shuf = mplan.PassShuffler(pb["sizes"], 10000, SHUFFLE_SEED)
    t0 = time.time()
    trace = []
    for plan in pb["plans"]:
        trace += oloops.mamdr_epoch(rec or model, theta, phis, g["data"]["train"], plan, shuf, batch, meta_lr)
    secs = time.time() - t0
    aucs = []
    for d in range(pb["D"]):
        model.set_flat(oouter.merge(theta, phis[d], "plus"))
        _, preds = model.evaluate(g["data"]["val"][d], batch)
        aucs.append(float(oauc.auc500(g["data"]["val"][d]["label"], preds, batch)))
    out = dict(trace=trace, aucs=aucs, secs=secs, theta=theta)
    if dump:
        out["dump"] = rec.close()
    return out


# ---------------------------------------------------------------------------------------------- configs[2]: DeepFM + DN
def perm_stream(sizes, base):
    from mamdr_amd import engine
    k = [0]

    def perm_fn(d):
        k[0] += 1
        return engine.shuffle_perm(sizes[d], 10000, base + k[0])
    return perm_fn


def problem_amazon6(batch=1024, steps=160):
    from mamdr_amd import synthetic
    from oracle import tower as otower
    shape = synthetic.SHAPES["amazon6"]
    g = synthetic.generate("amazon6", batch_size=batch, seed=123, row_scale=steps * batch / shape["n_train"],
                           splits=("train", "val"), hot=dict(users=2000, items=1000, share=0.7))
    D = g["n_domain"]
    params = otower.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], D, pretrained=False)
    sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
    seq = [int(d) for d in np.random.RandomState(5).permutation(D)]
    return dict(g=g, params=params, sizes=sizes, seq=seq, D=D)


def job_amazon6_dn(batch=1024):
    from oracle import auc as oauc
    from oracle import loops as oloops
    from oracle import tower as otower
    pb = problem_amazon6(batch)
    g = pb["g"]
    model = otower.OracleModel(pb["params"], emb_trainable=True, dropout=0.5, lr=1e-3, dropout_seed=DROPOUT_SEED,
                               tower="deepfm")
    theta = model.get_flat().copy()
    t0 = time.time()
    trace = oloops.dn_epoch(model, theta, g["data"]["train"], pb["seq"], perm_stream(pb["sizes"], 500), batch, 0.5)
    secs = time.time() - t0
    model.set_flat(theta)
    aucs = []
    for d in range(pb["D"]):
        _, preds = model.evaluate(g["data"]["val"][d], batch)
        aucs.append(float(oauc.auc500(g["data"]["val"][d]["label"], preds, batch)))
    from oracle import bigtable
    return dict(trace=trace, aucs=aucs, secs=secs,
                tables={n: np.ascontiguousarray(bigtable.densify(model.params[n]), F32) for n in ("user_emb", "item_emb")})


# ---------------------------------------------------------------------------------------------- configs[4]: Star + MAMDR
class StarMeta(object):
    """oracle Star model seen through its meta parameters (what the MAMDR loop reads and assigns, maml.py:153-194)."""

    def __init__(self, m):
        self.m = m

    def get_flat(self):
        return self.m.get_flat(meta_only=True)

    def set_flat(self, vec):
        self.m.set_flat(vec, meta_only=True)

    def train_pass(self, data, perm, batch_size, max_steps=0, accumulate_into=None):
        assert accumulate_into is None
        out = self.m.train_pass(data, perm, batch_size, max_steps)
        if self.shake is not None:
            # a perturbed twin (tests/ensemble.py): every tensor but the two 180 MB tables changed by one fp32 rounding after
            # every pass (the tables' touched rows pick the difference up in the next pass; shaking 92 M table elements 28
            # times would cost more than the oracle's epoch)
            rs, rel = self.shake
            for n_ in sorted(self.m.params):
                a = self.m.params[n_]
                if a.dtype == F32 and n_ not in ("user_emb", "item_emb"):
                    a[...] = perturbed(a, rs, rel)
        return out

    shake = None


def problem_amazon13(batch=8192, keras_init=False):
    """keras_init: PartitionedNorm gamma = 1 / beta = 0 (Star/partitioned_norm.py:19-22), zero biases (star_fcn.py:24-25)
    and phi_d = a second random initialisation (mamdr.py:31-33) -- the state `star_meta_mamdr` really starts from.
    Otherwise (round 4's conditioning): PN / biases moved off their special values, phi_d = 0."""
    from mamdr_amd import synthetic
    from oracle import star as ostar
    shape = synthetic.SHAPES["amazon13"]
    g = synthetic.generate("amazon13", batch_size=batch, seed=123, row_scale=90000 * 13 / shape["n_train"] / 3,
                           splits=("train", "val"), hot=dict(users=3000, items=1500, share=0.8))
    D = g["n_domain"]
    all_sizes = [g["data"]["train"][d]["uid"].shape[0] for d in range(D)]
    doms = sorted(range(D), key=lambda d: -all_sizes[d])[:4]
    params = ostar.init_params(np.random.RandomState(1024), g["n_user"], g["n_item"], D)
    if not keras_init:
        irs = np.random.RandomState(7)
        for n_ in ("pn_gamma_shared", "pn_gamma_spec"):
            params[n_] = (params[n_] + irs.standard_normal(params[n_].shape) * 0.2).astype(F32)
        for n_ in ("pn_beta_shared", "pn_beta_spec", "bs0", "bs1", "bs2", "bd0", "bd1", "bd2", "gb"):
            params[n_] = (irs.standard_normal(params[n_].shape) * 0.05).astype(F32)
    prs = np.random.RandomState(3)
    plan = {"seq": [doms[i] for i in prs.permutation(4)], "dr": []}
    for q in [doms[i] for i in prs.permutation(4)]:
        plan["dr"].append((q, [doms[i] for i in prs.permutation(4) if doms[i] != q][:2] + [q]))
    return dict(g=g, params=params, plan=plan, doms=doms, all_sizes=all_sizes, D=D)


def star_phi0(pb, d, n_meta, names_meta=None):
    """phi_d of the keras_init variant: the meta part of a second random initialisation of the model, seed 2000 + d
    (mamdr.py:31-33).  The two big tables are drawn in place of a full second model (2 x 92 M normals per domain)."""
    from oracle import star as ostar
    g = pb["g"]
    p2 = ostar.init_params(np.random.RandomState(2000 + d), g["n_user"], g["n_item"], pb["D"])
    m2 = ostar.OracleStar(p2, emb_trainable=True, lr=1e-3)
    v = m2.get_flat(meta_only=True).copy()
    assert v.size == n_meta
    return v


def job_amazon13_star(batch=8192, keras_init=False, perturb=0.0, phi0="init", pseed=99):
    """oracle/loops.mamdr_epoch on oracle/star.OracleStar (dense Adam over every table row and every per-domain slice
    each step).  perturb > 0: a perturbed TWIN -- every initial tensor multiplied by (1 + perturb * N(0, 1)) elementwise in fp32
    and the small tensors again after every pass (StarMeta.shake); `pseed` draws the twin.  Its distance from the unperturbed
    run measures the oracle's own sensitivity to rounding-level changes (the instrument of tests/test_gpu_parity.py's
    miniature Star test, here at the full table size)."""
    from oracle import auc as oauc
    from oracle import loops as oloops
    from oracle import outer as oouter
    from oracle import star as ostar
    pb = problem_amazon13(batch, keras_init)
    g, doms, plan = pb["g"], pb["doms"], pb["plan"]
    params = pb["params"]
    if perturb > 0:
        prs = np.random.RandomState(pseed)
        for n_ in sorted(params):
            a = params[n_]
            if a.dtype == F32:
                params[n_] = perturbed(a, prs, perturb)
    model = ostar.OracleStar(params, emb_trainable=True, lr=1e-3)
    wrapped = StarMeta(model)
    if perturb > 0:
        wrapped.shake = (np.random.RandomState(pseed + 7919), perturb)
    theta = wrapped.get_flat().copy()
    if keras_init and phi0 == "init":
        phis = {d: star_phi0(pb, d, theta.size) for d in doms}
    else:           # (keras_init with phi0 == "zero": round 4's diagnostic, profiles/r04_star13_phases_keras_init.txt)
        phis = {d: np.zeros_like(theta) for d in doms}
    t0 = time.time()
    trace = oloops.mamdr_epoch(wrapped, theta, phis, g["data"]["train"], plan, perm_stream(pb["all_sizes"], 900), batch, 0.5)
    secs = time.time() - t0
    aucs = {}
    for d in doms:
        wrapped.set_flat(oouter.merge(theta, phis[d], "plus"))
        _, preds = model.evaluate(g["data"]["val"][d], batch)
        aucs[d] = float(oauc.auc500(g["data"]["val"][d]["label"], preds, batch))
    tail = {n: np.array(model.params[n], F32).ravel() for n in ostar.param_names(True)[1]}
    return dict(trace=trace, aucs=aucs, secs=secs, tail=tail, n_meta=theta.size,
                mov_mean={d: model.state["mov_mean"][d].copy() for d in doms},
                mov_var={d: model.state["mov_var"][d].copy() for d in doms}, steps=model.state["steps"].copy())


# ---------------------------------------------------------------------------------------------- run.py end to end
class Recorder(object):
    """what the run.py pipeline decided, epoch by epoch: attached to a built model (cli.main's `on_model`), it wraps
    `val_and_test` and `early_stop_step` of the outermost wrapper and reads the traces / finetune log afterwards.
    Everything it keeps is plain python (it travels back from the worker process)."""

    def __init__(self):
        self.events = []
        self.model = None

    def attach(self, model):
        self.model = model
        rec = self.events
        base = getattr(model, "base_model", model)
        depth = [0]
        # the outermost wrapper AND the tower below it: a wrapper that leaves the training loop to the tower (uncertainty
        # weighting: train() = the base model's alternate loop) validates through the tower's methods.  Only the outermost of
        # nested calls is recorded.
        for target in ([model] if base is model else [model, base]):
            def wrap(target=target):
                vt, es = target.val_and_test, target.early_stop_step

                def val_and_test(mode):
                    depth[0] += 1
                    try:
                        out = vt(mode)
                    finally:
                        depth[0] -= 1
                    if depth[0] == 0:
                        rec.append(("eval", mode, float(out[0]), float(out[1]), {int(k): float(v) for k, v in out[2].items()},
                                    {int(k): float(v) for k, v in out[3].items()}))
                    return out

                def early_stop_step(metric):
                    depth[0] += 1
                    try:
                        stop = es(metric)
                    finally:
                        depth[0] -= 1
                    if depth[0] == 0:
                        rec.append(("early_stop", float(metric), float(base.best_metric), int(base.counter), bool(stop)))
                    return stop
                target.val_and_test = val_and_test
                target.early_stop_step = early_stop_step
            wrap()

    def summary(self, result):
        model = self.model
        base = getattr(model, "base_model", model)
        return dict(events=self.events, trace=[tuple(t) for t in getattr(model, "trace", [])],
                    finetune_log={int(d): dict(v) for d, v in getattr(base, "finetune_log", {}).items()},
                    result=(float(result[0]), float(result[1]), {int(k): float(v) for k, v in result[2].items()},
                            {int(k): float(v) for k, v in result[3].items()}))


def pipeline_config(cfg_file, name, tmp, train=None, dataset=None, model=None):
    import copy
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", cfg_file)) as f:
        cfg = copy.deepcopy(json.load(f))
    if name:
        cfg["model"]["name"] = name
    cfg["train"].update(result_save_path=os.path.join(tmp, "result"), checkpoint_path=os.path.join(tmp, "ckpt"))
    # (job keys are hashable: list-valued settings travel as tuples)
    cfg["train"].update({k: (list(v) if isinstance(v, tuple) else v) for k, v in (train or {}).items()})
    cfg["dataset"].update(dataset or {})
    cfg["model"].update(model or {})
    return cfg


def run_pipeline(cfg, engine_factory=None, perturb=0.0, pseed=99):
    """mamdr_amd.cli.main (= the reference's run.py:71-89: train -> val / early stop -> best state -> test -> finetune ->
    save_result) with a Recorder attached.  -> the Recorder's summary + result.json as written.
    perturb > 0: the built model's initial weights (what theta starts from) are changed at rounding level before
    training starts -- a run of the self-divergence instrument; `pseed` draws the perturbation (the K twins of an ensemble
    differ in it)."""
    import json
    from mamdr_amd import cli
    from mamdr_amd import parallel
    recs = {}

    def on_model(model):
        # (train.lanes > 1: called once per lane, on the lane's thread; every lane sees the same gathered results and
        # takes the same decisions -- lane 0's record is the run's, the lanes' traces are kept beside it)
        recs[parallel.world()[0]] = r = Recorder()
        r.attach(model)
        if perturb > 0:
            # a twin = an implementation that ROUNDS DIFFERENTLY: its initial weights and the live weights after every pass
            # (every train_steps call) are changed by one fp32 rounding, relative `perturb` = 2e-7 per element.  (Round 5's
            # twins differed in the initial weights only; the HIP engine differs from the oracle in every contraction's
            # summation order and in exp / log -- the teacher-forced tests measure 1e-7 relative in a step's loss and a
            # median of ~1e-3 lr per step in the weights after ONE pass -- so a twin that re-injects rounding noise per pass
            # is the fair model of "another fp32 evaluation of this training", and still a conservative one.)
            import torch
            eng = model.model
            # (the initial draw is the same on every lane of a lane run -- the lanes must start from ONE model --, the
            # per-pass draws are the lane's own)
            rs0, rs = np.random.RandomState(pseed), np.random.RandomState(pseed + 7919 * (parallel.world()[0] + 1))

            def shake(r):
                w = eng.get_weights().clone()
                noise = r.standard_normal(w.numel()).astype(F32)
                eng.set_weights(w * (1 + perturb * torch.from_numpy(noise).to(w.device)))
            shake(rs0)
            inner = eng.train_steps

            def train_steps(*a, **k):
                out = inner(*a, **k)
                shake(rs)
                return out
            eng.train_steps = train_steps
    out = cli.main(cfg, engine_factory, on_model=on_model)
    s = recs[0].summary(out)
    s["lane_traces"] = {r: [tuple(t) for t in getattr(rec.model, "trace", [])] for r, rec in sorted(recs.items())}
    s["lane_events"] = {r: rec.events for r, rec in sorted(recs.items())}
    rdir = cfg["train"]["result_save_path"]
    found = [os.path.join(r, "result.json") for r, _, fs in os.walk(rdir) if "result.json" in fs]
    with open(found[0]) as f:
        s["result_json"] = json.load(f)
    return s


def job_pipeline(cfg_file, model_name, train, dataset, perturb=0.0, model=(), pseed=99):
    """the oracle twin of a whole run: the SAME host code (cli.main, model_zoo/*, meta.py) on tests/fake_engine.FakeEngine,
    i.e. every numeric call answered by the numpy oracle.  train / dataset: tuples of (key, value) overrides."""
    import contextlib
    import io
    import tempfile
    from fake_engine import fake_factory as FakeEngine      # (FakeEngine, or FakeStarEngine for the Star tower)
    tmp = tempfile.mkdtemp(prefix="mamdr_twin_")
    cfg = pipeline_config(cfg_file, model_name, tmp, dict(train), dict(dataset), dict(model))
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        s = run_pipeline(cfg, FakeEngine, perturb, pseed)
    s["secs"] = time.time() - t0
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return s


JOBS = {"fullsize_mamdr": job_fullsize_mamdr, "amazon6_dn": job_amazon6_dn, "amazon13_star": job_amazon13_star,
        "pipeline": job_pipeline}
COST = {"fullsize_mamdr": 1.0, "amazon6_dn": 1.5, "amazon13_star": 2.0, "pipeline": 6.0}
