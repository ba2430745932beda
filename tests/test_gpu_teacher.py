"""Teacher-forced full-size epochs on the GPU: every pass of one MAMDR meta-epoch of BASELINE.json configs[1] (Taobao-10 bs
1,024, the k_wgrad_adam path) and configs[3] (Taobao-30 bs 4,096, the slab path) is started from the ORACLE's state at that
point and compared pass by pass -- per-step losses and the end state (weights, Adam m / v) -- with tight bars and no
self-divergence term (tests/teacher.py).  The oracle run is the one tests/test_gpu_fullsize.py's
full-epoch AUC test reads (tests/oracle_jobs.job_fullsize_mamdr with dump=True: one oracle epoch, two tests); the trainable
FULL-table configs[2] / configs[4] run their oracle in lock-step inside this process (further down).

Reference call sites of what a pass is: model_zoo/mamdr.py:44-108 (`model.fit(.., steps_per_epoch=train_step)` /
`train_on_batch` loops over a re-initialised iterator), DeepCTR/deepctr.py:54-60 (tf.train.AdamOptimizer: one slot set for the
whole run), utils/dataset.py:20-38 (shuffle, final partial batch kept).

Bars (per pass of n steps, lr = 1e-3; `k lr` = n * lr):
  * loss of the pass's FIRST step (identical weights on both sides): relative 2e-5 -- one step's forward pass;
  * loss of every later step: relative 2e-4 (the weights of the two sides have taken up to n - 1 Adam steps apart: Adam turns
    a rounding-level gradient difference of an element whose gradient is ~0 into a step of up to lr);
  * end weights: the measures of tests/test_gpu_parity.assert_adam_close at its bars (<= 1e-3 of the elements beyond 5 % of
    k lr, none beyond 2.02 k lr, median <= 0.002 k lr);
  * Adam slots at the end, relative L2 distance per pass: v <= 5e-3; m <= 0.1 -- a gross-error bar: the first moment is the
    chaotic quantity of a pass (the ORACLE restarted from its own state with the weights changed by one fp32 rounding ends a
    31-step pass 3.5e-2 away in m, 2.8e-4 in v, 6.5 % of k lr in its worst weight: profiles/r06_teacher_probe.txt), which is
    exactly why the passes are restarted from the oracle's state;
  * the two launch paths of a pass (announced to mamdr_pregather_passes or not, losses written or not) end in the same bits.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import oracle_jobs      # noqa: E402
import teacher          # noqa: E402

from teacher_bars import BARS      # noqa: E402


def _frozen_case(shape, batch, meta_lr):
    from mamdr_amd import engine, plan as mplan
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    pb = oracle_jobs.problem_fullsize(shape, batch, 1)
    g, params, names, sizes, D = (pb[k] for k in ("g", "params", "names", "sizes", "D"))
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5)
    assert eng.dropout_seed == oracle_jobs.DROPOUT_SEED
    eng.bind_table("user_emb", params["user_emb"])
    eng.bind_table("item_emb", params["item_emb"])
    for d in range(D):
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    side = teacher.HipSide(eng, names, {n: params[n].size for n in names})
    ora = oracle_jobs.result("fullsize_mamdr", shape=shape, batch=batch, meta_lr=meta_lr, epochs=1, perturb=0.0, dump=True)
    shuf = mplan.PassShuffler(sizes, 10000, oracle_jobs.SHUFFLE_SEED)       # the oracle's stream, pass by pass
    by_phase = {}

    def report(k, phase, d, n_st, frac, mx, med, mr, vr):
        w = by_phase.setdefault(phase, [0, 0, 0.0, 0.0, 0.0])
        w[0] += 1
        w[1] += n_st
        w[2], w[3], w[4] = max(w[2], frac), max(w[3], mx), max(w[4], max(mr, vr))
    t0 = time.time()
    segs, o = [], 0
    for n in names:
        segs.append((n, o, params[n].size))
        o += params[n].size
    out = teacher.run_teacher_forced(side, ora["dump"], ora["trace"], shuf, batch, 1e-3, BARS, report=report, segments=segs)
    secs = time.time() - t0
    fused = batch <= 1024
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == (1 if fused else 0)
    if fused:       # variant 0 of every pass was served by its k_pass_prep_multi launch
        assert int(eng.lib.mamdr_pregather_launches(eng.ctx)) == out["passes"] == int(eng.lib.mamdr_pregather_hits(eng.ctx))
        # (variant 1 of every pass forgot the hint and ran k_pass_prep inside its call)
    print("%s bs %d teacher-forced: %d passes / %d domain-steps (%d passes end in a partial batch), hip %.1f s x 2 launch paths; "
          "oracle epoch %.1f s (waited %.1f s)" % (shape, batch, out["passes"], out["steps"], out["ragged_passes"], secs,
                                                  ora["secs"], ora.get("waited_seconds", 0.0)))
    print("  worst over the passes: first-step loss rel %.1e (bar %.0e), any-step loss rel %.1e (%.0e); end weights: fraction "
          "beyond 5 %% of k lr %.1e (%.0e), max %.3f k lr (%.2f), median %.5f k lr (%.3f); Adam m / v relative L2 %.1e / %.1e (%.0e / %.0e)" % (
              out["loss_first"], BARS["loss_first"], out["loss_rel"], BARS["loss_rel"], out["frac"], BARS["frac"], out["max_klr"],
              BARS["max_klr"], out["med_klr"], BARS["med_klr"], out["m_rel"], out["v_rel"], BARS["m_rel"], BARS["v_rel"]))
    for phase, (n, st, frac, mx, mv) in sorted(by_phase.items()):
        print("    %-10s %3d passes %5d steps: worst fraction %.1e, max %.3f k lr, slots %.1e" % (phase, n, st, frac, mx, mv))
    for v in out["violations"][:12]:
        print("  VIOLATION", v)
    assert not out["violations"], "%d violations in %d passes (first: %r)" % (len(out["violations"]), out["passes"], out["violations"][0])
    assert out["ragged_passes"] >= D and out["steps"] == sum(t[2] for t in ora["trace"])
    eng.close()
    return out


@pytest.mark.oracle_job("fullsize_mamdr", shape="taobao10", batch=1024, meta_lr=0.5, epochs=1, perturb=0.0, dump=True)
def test_taobao10_bs1024_epoch_teacher_forced():
    """BASELINE.json configs[1]: 130 passes / 1,2xx steps of k_pass_prep(_multi) + k_tower4<.., W1L, PRE(, W2D)> + k_wgrad_adam
    with the domain table's pending step, each from the oracle's state."""
    out = _frozen_case("taobao10", 1024, 0.5)
    assert out["passes"] >= 120


@pytest.mark.oracle_job("fullsize_mamdr", shape="taobao30", batch=4096, meta_lr=0.5, epochs=1, perturb=0.0, dump=True)
def test_taobao30_bs4096_epoch_teacher_forced():
    """BASELINE.json configs[3] / north_star's 1-GPU target: 390 passes / 1,5xx steps of k_tower (16-row tiles, and the
    four-row tower on the short last batches) + k_wgrad(_pf) + k_update, each from the oracle's state."""
    out = _frozen_case("taobao30", 4096, 0.5)
    assert out["passes"] >= 380


# ------------------------------------------------------------------ configs[2] / configs[4]: trainable FULL-size tables
def _report_lockstep(title, ls, secs, bars):
    out = ls.summary()
    print("%s teacher-forced (oracle in lock-step): %d passes in %d chunks of <= %d steps / %d domain-steps (%d passes end in a "
          "partial batch), %.1f s" % (title, out["passes"], out["chunks"], ls.chunk, out["steps"], out["ragged_passes"], secs))
    print("  worst over the passes and tensors: first-step loss rel %.1e (bar %.0e), any-step loss rel %.1e (%.0e); end weights: "
          "fraction beyond 5 %% of k lr %.1e (%.0e), max %.3f k lr (%.2f), median %.5f k lr (%.3f); Adam m / v relative L2 %.1e / %.1e "
          "(%.0e / %.0e)%s" % (out["loss_first"], bars["loss_first"], out["loss_rel"], bars["loss_rel"], out["frac"], bars["frac"],
                        out["max_klr"], bars["max_klr"], out["med_klr"], bars["med_klr"], out["m_rel"], out["v_rel"], bars["m_rel"], bars["v_rel"],
                        "; moving statistics relative L2 %.1e" % out["aux_rel"] if "aux_rel" in out else ""))
    for r in ls.rows:
        print("    chunk %3d domain %2d steps %3d..%3d (%6d rows): loss %.1e / %.1e, weights frac %.1e max %.3f med %.5f, slots %.1e / %.1e" % (
            r["k"], r["d"], r["first"], r["first"] + r["n"] - 1, r["rows"], r["loss_first"], r["loss_rel"], r["frac"], r["max_klr"],
            r["med_klr"], r["m_rel"], r["v_rel"]))
    for v in ls.bad[:12]:
        print("  VIOLATION", v)
    assert not ls.bad, "%d violations in %d chunks (first: %r)" % (len(ls.bad), out["chunks"], ls.bad[0])
    return out


def test_amazon6_deepfm_dn_epoch_teacher_forced():
    """BASELINE.json configs[2] on Amazon-6's FULL trainable tables (79.2 M parameters with their Adam slots): every pass of the
    Domain Negotiation epoch of tests/test_gpu_fullsize.py (domain_negotiation.py:53-88; 6 passes, 160+ steps) from the oracle's
    state -- k_tower4<DX, FM> + [k_wgrad + k_emb_reduce + k_emb_rows] + [k_update + k_lin_sweep + k_emb_catchup] and the LAZY
    table Adam with its forced flushes, whose replayed rows (all 618 K of them move every step under TF1's dense Adam,
    deepctr.py:54-60,118-126) are part of the compared end state."""
    from mamdr_amd import engine
    from oracle import loops as oloops
    from oracle import tower as otower
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    batch = 1024
    pb = oracle_jobs.problem_amazon6(batch)
    g, params, sizes, seq, D = (pb[k] for k in ("g", "params", "sizes", "seq", "D"))
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5, emb_trainable=True, tower="deepfm")
    for d in range(D):
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    model = otower.OracleModel(params, emb_trainable=True, dropout=0.5, lr=1e-3, dropout_seed=oracle_jobs.DROPOUT_SEED,
                               tower="deepfm")
    assert sorted(model.names) == sorted(eng.segments)
    theta = model.get_flat().copy()
    # 8-step chunks (teacher.LockStep.train_pass: this configuration amplifies a rounding-level difference ~1.15x per step).
    # Even so single chunks show EVENTS: Adam is sign-like on elements whose second moment is tiny (the tables start at
    # N(0, 1e-4^2), so do the rows of W0 they feed), and an element whose gradient is rounding residue moves +- lr whichever way
    # the residue points.  The oracle against copies of its own state perturbed by one rounding, same chunks: up to 7.8e-3 of
    # W0's elements beyond 5 % of k lr (profiles/r06_teacher_probe.txt) -- hence 1e-2 here instead of 1e-3; the other bars
    # are the frozen-table cases'.
    bars = dict(BARS, frac=1e-2)
    ls = teacher.LockStep(model, model, eng, g["data"]["train"], 1e-3, bars, chunk=8)
    t0 = time.time()
    trace = oloops.dn_epoch(ls, theta, g["data"]["train"], seq, oracle_jobs.perm_stream(sizes, 500), batch, 0.5)
    out = _report_lockstep("amazon6 deepfm DN bs 1024, full tables", ls, time.time() - t0, bars)
    assert out["passes"] == len(trace) == D and out["steps"] >= 150
    # (8-step chunks never reach the lazy table Adam's forced flush -- 32 steps; the free-running epoch of
    # tests/test_gpu_fullsize.py asserts >= 4 of them -- every chunk END replays every lagging row of both tables, which is
    # what the end-state comparison covers)
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 0
    eng.close()


def test_amazon13_star_mamdr_epoch_teacher_forced():
    """BASELINE.json configs[4] on Amazon-13's FULL trainable tables (91.9 M parameters inside theta / phi) from the reference's own
    initial values (PartitionedNorm gamma = 1 / beta = 0, zero biases, phi_d = a second random initialisation): every pass of the
    DN + DR meta-epoch of tests/test_gpu_fullsize.py (mamdr.py:41-108; 28 passes, 220 steps at bs 8,192) from the oracle's state,
    PartitionedNorm's moving statistics and zero-debias slots included (partitioned_norm.py:177-193) -- k_star_stats / k_star_prep
    + k_tower<train, 384> + PartitionedNorm backward + k_wgrad_reduce + k_star_update_catchup, lazy table Adam, lazy slices."""
    from mamdr_amd import engine
    from oracle import loops as oloops
    from oracle import star as ostar
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    batch = 8192
    pb = oracle_jobs.problem_amazon13(batch, True)
    g, params, plan, doms, all_sizes, D = (pb[k] for k in ("g", "params", "plan", "doms", "all_sizes", "D"))
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.0, emb_trainable=True, tower="star")
    for d in doms:
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    model = ostar.OracleStar(params, emb_trainable=True, lr=1e-3)
    assert sorted(model.names) == sorted(eng.segments)
    wrapped = oracle_jobs.StarMeta(model)
    theta = wrapped.get_flat().copy()
    phis = {d: oracle_jobs.star_phi0(pb, d, theta.size) for d in doms}

    def aux_of(m):
        st = m.state
        return np.concatenate([st[k].ravel() for k in ("mov_mean", "mov_var", "biased_mean", "biased_var", "steps")]).astype(np.float32)
    assert 0 <= eng.aux.numel() - aux_of(model).size < 4
    bars = dict(BARS, aux_rel=1e-4)
    # (whole passes: this tower is well conditioned -- 15 steps at most; the domain table's Adam slots average a gradient that
    # is rounding residue on both sides, teacher.LockStep)
    ls = teacher.LockStep(wrapped, model, eng, g["data"]["train"], 1e-3, bars, aux_of=aux_of, chunk=1 << 20,
                          noise_slots=("domain_emb",))
    t0 = time.time()
    trace = oloops.mamdr_epoch(ls, theta, phis, g["data"]["train"], plan, oracle_jobs.perm_stream(all_sizes, 900), batch, 0.5)
    out = _report_lockstep("amazon13 star MAMDR bs 8192, full tables, Keras init", ls, time.time() - t0, bars)
    assert out["passes"] == out["chunks"] == len(trace) == 28 and out["steps"] >= 150
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 0
    eng.close()
