"""The teacher-forced harness checked against itself on the CPU (tests/teacher.py, tests/oracle_jobs.PassDump): a SECOND oracle
model started from every dumped pass state must reproduce the dumped losses and end states bit for bit -- the dump's layout,
the slot hand-over between passes, the counters (Adam t with TF's running beta powers, dropout position) and the shuffle
stream are what the GPU tests (tests/test_gpu_teacher.py) rely on."""
import shutil

import numpy as np

import oracle_jobs
import teacher

SHAPE = {"name": "Taobao", "split": "mini", "n_domain": 3, "n_user": 400, "n_item": 150, "n_train": 2600, "n_val": 300,
         "n_test": 300, "pretrained": True}


def test_dumped_passes_replay_bit_for_bit():
    from mamdr_amd import plan as mplan
    from oracle import tower as otower
    batch = 256
    ora = oracle_jobs.job_fullsize_mamdr(SHAPE, batch, 0.5, 1, dump=True)
    try:
        pb = oracle_jobs.problem_fullsize(SHAPE, batch, 1)
        g, params = pb["g"], pb["params"]
        dump = ora["dump"]
        assert len(dump["meta"]) == len(ora["trace"]) == 3 + 2 * sum(len(s) for _, s in pb["plans"][0]["dr"])
        recs, tail = oracle_jobs.read_dump(dump)
        # the slots are handed from pass to pass untouched; weights are re-assigned between passes (DN -> DR, support -> query)
        t = 0
        for k, (t0, s0, n) in enumerate(dump["meta"]):
            assert t0 == s0 == t
            t += n
        model = otower.OracleModel({k: (v if k in ("user_emb", "item_emb") else v.copy()) for k, v in params.items()},
                                   dropout=0.5, lr=1e-3, dropout_seed=oracle_jobs.DROPOUT_SEED)
        side = teacher.OracleSide(model, g["data"]["train"])
        shuf = mplan.PassShuffler(pb["sizes"], 10000, oracle_jobs.SHUFFLE_SEED)
        bars = dict(loss_first=0.0, loss_rel=0.0, frac=0.0, max_klr=0.0, med_klr=0.0, m_rel=0.0, v_rel=0.0)
        out = teacher.run_teacher_forced(side, dump, ora["trace"], shuf, batch, 1e-3, bars, variants=(1, 1), exact=True)
        assert out["passes"] == len(ora["trace"]) and out["steps"] == t and out["ragged_passes"] >= 1
        assert out["loss_rel"] == 0.0 and out["max_klr"] == 0.0 and out["m_rel"] == 0.0 and not out["violations"]
        # ... and a harness that could not tell a wrong state from a right one would be worthless: one weight off by 0.05
        # at the start of a pass must trip the bars of the GPU test
        from teacher_bars import BARS
        class Bent(teacher.OracleSide):
            def load(self, w, m, v, t_, step):
                w = np.array(w, np.float32)
                w[w.size // 2] += np.float32(0.05)
                teacher.OracleSide.load(self, w, m, v, t_, step)
        bent = Bent(model, g["data"]["train"])
        shuf = mplan.PassShuffler(pb["sizes"], 10000, oracle_jobs.SHUFFLE_SEED)
        out2 = teacher.run_teacher_forced(bent, dump, ora["trace"], shuf, batch, 1e-3, BARS, variants=(1,))
        assert out2["violations"], "a perturbed start state passed the teacher-forced bars"
    finally:
        shutil.rmtree(ora["dump"]["dir"], ignore_errors=True)


def test_ensemble_bars_accept_a_member_and_reject_an_offset():
    """tests/ensemble.Ensemble on synthetic AUCs: runs drawn like the members pass (20 of 20 draws); a run with a
    systematic offset of two sigma fails the mean-distance bar; a single comparison twelve sigma out fails the
    largest-distance bar."""
    import pytest
    pytest.importorskip("torch")
    from ensemble import Ensemble
    rs = np.random.RandomState(0)
    D, E, K1 = 10, 6, 6
    sigma = 8e-4 * (1 + rs.rand(1, D))                       # (domains differ in size, hence in spread)
    base = 0.7 + 0.1 * rs.rand(E, D)
    members = [base + sigma * rs.standard_normal((E, D)) for _ in range(K1)]

    def run(h):
        ens = Ensemble([None] * K1)
        for e in range(E):
            ens.check(("val", e), {d: h[e, d] for d in range(D)}, [{d: m[e, d] for d in range(D)} for m in members])
        ens.aggregate("synthetic")
        return ens
    for _ in range(20):
        ens = run(base + sigma * rs.standard_normal((E, D)))
    assert ens.n_cmp == E * D and ens.beyond >= 1            # (at this sigma some comparisons exceed the plain 1e-3)
    with pytest.raises(AssertionError, match="mean distance"):
        run(base + 2 * sigma + sigma * rs.standard_normal((E, D)))
    h = base + sigma * rs.standard_normal((E, D))
    h[2, 3] += 12 * sigma[0, 3]
    with pytest.raises(AssertionError, match="largest distance"):
        run(h)
