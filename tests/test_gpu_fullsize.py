"""GPU parity at the metric's OWN workloads: one whole MAMDR meta-epoch, per-domain AUC within 1e-3.

BASELINE.json `metric` is quoted on Taobao-10 bs 1024 (configs[1]) and north_star's target on Taobao-30 bs 4096
(configs[3]).  tests/test_gpu_parity.py::test_mamdr_epoch_auc_parity shows the 1e-3 bar in miniature (4 domains,
256 rows); here the HIP path runs the workloads themselves -- every domain, full tables, full splits, the config's
sample_num / add_query_domain, the kernels those batch sizes select (bs 1024: k_pass_prep + k_tower4<.., W1L, PRE> +
k_wgrad_adam + the pending domain-table step; bs 4096: k_tower + k_wgrad + k_update) -- against the numpy oracle
(oracle/loops.mamdr_epoch, the restatement of model_zoo/mamdr.py:41-108 + specific_base_model.py:64-97) on the same
plan, shuffles and dropout masks.  Asserted: the traces are equal and |AUC_hip - AUC_oracle| <= 1e-3 on EVERY
domain's validation split with the merged weights theta + phi_d (north_star: "per-domain AUC matches the reference
within 1e-3") -- plain, no self-divergence term (round 6); the two two-epoch cases at the configs' own meta learning
rate, where a rounding-level twin of the ORACLE is already up to 8.5e-4 from it, are asserted against an ensemble of six
oracle runs instead (tests/ensemble.py).  Every pass of the same epochs is also compared chaos-free: tests/test_gpu_teacher.py.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_jobs                      # noqa: E402  (tests/oracle_jobs.py: the oracle side, in worker processes)
from ensemble import Ensemble, TWIN_SEEDS      # noqa: E402
from oracle import tower as otower      # noqa: E402

F32 = np.float32


def run_case(shape, batch, meta_lr, epochs=1, dump=False, ensemble=False):
    """HIP side on the host path bench.py times (bench.py:439-482): plan.EpochShuffles (every permutation of an epoch from
    one C call, one upload, the NEXT epoch's drawn on the prefetch thread) + parallel.BalancedMAMDR(world 1).epoch, hence
    meta.PassWindow with `peek` -> mamdr_pregather_passes -> k_pass_prep_multi on the fused path.  The oracle (worker
    process, tests/oracle_jobs.job_fullsize_mamdr) draws the same permutations pass by pass from plan.PassShuffler."""
    from mamdr_amd import engine, meta, parallel, plan as mplan
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    pb = oracle_jobs.problem_fullsize(shape, batch, epochs)
    g, params, plans, names, phis0, sizes, D = (pb[k] for k in ("g", "params", "plans", "names", "phis0", "sizes", "D"))
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5)
    assert eng.dropout_seed == oracle_jobs.DROPOUT_SEED
    eng.bind_table("user_emb", params["user_emb"])
    eng.bind_table("item_emb", params["item_emb"])
    for split in ("train", "val"):
        for d in range(D):
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
    assert sizes == [eng.n_rows(d, "train") for d in range(D)]

    def to_dev(flat):
        named, o = {}, 0
        for nme in names:
            sz = params[nme].size
            named[nme] = flat[o:o + sz]
            o += sz
        return eng.pack(named)

    theta_g = to_dev(otower.flatten(params, names))
    steps_per_domain = [-(-n // batch) for n in sizes]
    balanced = parallel.BalancedMAMDR(eng, meta, theta_g, {d: to_dev(p) for d, p in enumerate(phis0)}, steps_per_domain)
    shuffles = mplan.EpochShuffles(mplan.PassShuffler(sizes, 10000, oracle_jobs.SHUFFLE_SEED), eng.device)
    trace_g = []
    t0 = time.time()
    for k, plan in enumerate(plans):
        trace_g += balanced.epoch(plan, shuffles.prepare, shuffles, batch, 1e-3, meta_lr, "plus")
        if k + 1 < len(plans):
            shuffles.prefetch(balanced.local_passes(plans[k + 1]))
    torch.cuda.synchronize()
    gsecs = time.time() - t0
    fused = batch <= 1024
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == (1 if fused else 0)
    hits, launches = int(eng.lib.mamdr_pregather_hits(eng.ctx)), int(eng.lib.mamdr_pregather_launches(eng.ctx))
    if fused:          # the pass windows of the bench's host path ran: <= 16 passes per k_pass_prep_multi launch
        assert launches >= len(trace_g) // 16 and hits >= len(trace_g) - launches, (hits, launches, len(trace_g))
        assert launches < len(trace_g) // 4
    else:              # (the slab path gathers inside the tower: the hint is a no-op there)
        assert hits == 0 and launches == 0
    ora = oracle_jobs.result("fullsize_mamdr", shape=shape, batch=batch, meta_lr=meta_lr, epochs=epochs, perturb=0.0,
                             **({"dump": True} if dump else {}))
    # perturbed oracle twins (tests/ensemble.py): runs whose initial tensors and live weights after every pass differ at
    # rounding level -- what ANY fp32 evaluation of this training is distributed like.  One twin is printed beside every
    # case; the two-epoch cases at the configs' own meta learning rate (the chaotic ones: a twin of theirs is up to 8.5e-4
    # from the oracle) are asserted against the ensemble of K = 5, the others against north_star's PLAIN 1e-3.
    twins = [oracle_jobs.result("fullsize_mamdr", shape=shape, batch=batch, meta_lr=meta_lr, epochs=epochs, perturb=PERTURB,
                                pseed=sd) for sd in (TWIN_SEEDS if ensemble else TWIN_SEEDS[:1])]
    orb = twins[0]
    assert all(trace_g == m["trace"] for m in [ora] + twins)
    n_steps = sum(t[2] for t in trace_g)
    print("%s bs %d: %d domain-steps in %d passes (%d pregather launches, %d calls served); oracle %.1f s (waited %.1f s), "
          "hip %.2f s" % (shape, batch, n_steps, len(trace_g), launches, hits, ora["secs"], ora.get("waited_seconds", 0.0),
                          gsecs))
    merged = eng.new_vector()
    worst, worst_self, beyond, aucs, auc_hip = 0.0, 0.0, 0, [], {}
    for d in range(D):
        eng.merge(merged, theta_g, balanced.phis[d], "plus")
        eng.set_weights(merged)
        _, auc_g = eng.evaluate(d, "val")
        auc_hip[d] = auc_g
        auc_o, self_div = ora["aucs"][d], abs(ora["aucs"][d] - orb["aucs"][d])
        print("  domain %2d: val rows %6d  AUC hip %.5f oracle %.5f  diff %+.1e   (oracle vs its first twin %.1e)" % (
            d, eng.n_rows(d, "val"), auc_g, auc_o, auc_g - auc_o, self_div))
        worst, worst_self = max(worst, abs(auc_g - auc_o)), max(worst_self, self_div)
        beyond += abs(auc_g - auc_o) > 1e-3
        aucs.append(auc_o)
        if not ensemble:
            assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)           # north_star's bar, plain
    if ensemble:        # no factor on a single draw, no allowed share of misses: not an outlier of six oracle runs
        ens = Ensemble([ora] + twins)
        ens.check("val", auc_hip, [{d: m["aucs"][d] for d in range(D)} for m in [ora] + twins])
        ens.aggregate("%s bs %d, %d epochs" % (shape, batch, epochs))
    # theta itself (0.56 MB): the outer updates of both sides applied to inner passes that agree to rounding
    th_g, th_o = eng.unpack(theta_g), ora["theta"]
    o, worst_th = 0, 0.0
    for nme in names:
        sz = params[nme].size
        a, b = np.asarray(th_g[nme]).ravel(), th_o[o:o + sz]
        o += sz
        rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        worst_th = max(worst_th, rel)
        assert rel < 0.1, (nme, rel)           # (gross-error bar; the measure of closeness is the AUC bar above)
    # the comparison is made on a model that has learnt (predictions spread over the threshold grid)
    assert float(np.mean(aucs)) > 0.6, aucs
    print("  worst |dAUC| %.2e (oracle vs its perturbed twin: %.2e; %d of %d domains beyond the plain 1e-3), mean oracle AUC "
          "%.4f; theta: worst per-tensor relative L2 distance %.1e" % (worst, worst_self, beyond, D, float(np.mean(aucs)), worst_th))
    eng.close()


PERTURB = 2e-7


def _job(shape, batch, meta_lr, epochs, dump=False, ensemble=False):
    """dump: the unperturbed oracle run also writes its per-pass states -- the SAME run serves the teacher-forced epoch of
    tests/test_gpu_teacher.py (one oracle epoch, two tests).  ensemble: K = 5 perturbed twins instead of one."""
    def deco(fn):
        fn = pytest.mark.oracle_job("fullsize_mamdr", shape=shape, batch=batch, meta_lr=meta_lr, epochs=epochs, perturb=0.0,
                                    **({"dump": True} if dump else {}))(fn)
        for sd in (TWIN_SEEDS if ensemble else TWIN_SEEDS[:1]):
            fn = pytest.mark.oracle_job("fullsize_mamdr", shape=shape, batch=batch, meta_lr=meta_lr, epochs=epochs,
                                        perturb=PERTURB, pseed=sd)(fn)
        return fn
    return deco


@_job("taobao10", 1024, 0.1, 2, ensemble=True)
def test_taobao10_bs1024_two_epochs_config_meta_lr():
    """BASELINE.json configs[1] as configured (config/Taobao-10/deepctr_DN+DR.json: meta_learning_rate 0.1): 10
    domains, 92,137 train rows, bs 1024 -> 2 x 1,2xx inner steps on the fused path (oracle AUC ~0.6-0.75 by then)."""
    run_case("taobao10", 1024, meta_lr=0.1, epochs=2, ensemble=True)


@_job("taobao10", 1024, 0.5, 1, dump=True)
def test_taobao10_bs1024_full_epoch_auc_parity():
    """the same workload with meta lr 0.5, so that ONE epoch already gives a trained model (oracle AUC 0.76-0.82:
    predictions spread over the 500 thresholds) -- the setting of the miniature test in tests/test_gpu_parity.py."""
    run_case("taobao10", 1024, meta_lr=0.5, dump=True)


@_job("taobao30", 4096, 0.5, 1, dump=True)
def test_taobao30_bs4096_full_epoch_auc_parity():
    """BASELINE.json configs[3] / north_star target: 30 domains, 394,805 train rows, bs 4096 (slab path), 1,501
    inner steps; meta lr 0.5 as above (at the config's 0.1 one epoch leaves the oracle at AUC ~0.57)."""
    run_case("taobao30", 4096, meta_lr=0.5, dump=True)


@_job("taobao30", 4096, 0.1, 2, ensemble=True)
def test_taobao30_bs4096_two_epochs_config_meta_lr():
    """BASELINE.json configs[3] as configured (config/Taobao_30/deepctr_DN+DR_bs4096.json: meta_learning_rate 0.1): two
    meta-epochs = 3,0xx inner steps on the slab path (the one-epoch case above runs at 0.5)."""
    run_case("taobao30", 4096, meta_lr=0.1, epochs=2, ensemble=True)


# ------------------------------------------------------------------ configs[2] and configs[4]: trainable FULL-SIZE tables
def _bind_splits(eng, g, domains, splits=("train", "val")):
    for split in splits:
        for d in domains:
            c = g["data"][split][d]
            eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])


@pytest.mark.oracle_job("amazon6_dn", batch=1024)
def test_amazon6_deepfm_dn_epoch_trainable_full_tables():
    """BASELINE.json configs[2] on its own kernel path: deepfm_meta_domain_negotiation, Amazon-6's FULL tables
    (445,789 + 172,653 rows x 128, trainable, N(0, 1e-4^2) as deepctr initialises them: 79.2 M parameters with their
    Adam slots), bs 1,024 -> k_tower4<DX, FM> + [k_wgrad + k_emb_reduce + k_emb_rows] + [k_update + k_lin_sweep +
    k_emb_catchup], the lazy table Adam with its forced flushes.  One Domain Negotiation epoch
    (model_zoo/domain_negotiation.py:53-88: sequential passes over all 6 domains from theta, theta += beta (theta~ - theta))
    of 160+ inner steps on a row sample of the config's data in which 70 % of the rows come from 2,000 users / 1,000
    items per domain (learnable in one epoch) and the rest from the whole per-domain subsets (rows seen once or never:
    long replay gaps; most of the 618 K rows are only ever moved by the regulariser, as TF1's dense Adam moves them).
    Oracle (tests/oracle_jobs.job_amazon6_dn, a worker process): oracle/loops.dn_epoch on oracle/tower.OracleModel --
    dense Adam over every row of both tables every step (oracle/bigtable.py).  Asserted: equal traces, >= 4 forced
    flushes inside the passes, the slab step path, |AUC_hip - AUC_oracle| <= 1e-3 on every domain's validation split,
    oracle mean AUC > 0.6."""
    from mamdr_amd import engine, meta
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    batch = 1024
    pb = oracle_jobs.problem_amazon6(batch)
    g, params, sizes, seq, D = (pb[k] for k in ("g", "params", "sizes", "seq", "D"))
    assert (g["n_user"], g["n_item"]) == (445789, 172653)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5, emb_trainable=True, tower="deepfm")
    assert eng.dropout_seed == oracle_jobs.DROPOUT_SEED
    _bind_splits(eng, g, range(D))
    eng.set_weights(eng.pack(params))
    assert sizes == [eng.n_rows(d, "train") for d in range(D)]
    theta_g = eng.get_weights()
    t0 = time.time()
    trace_g = meta.dn_epoch(eng, theta_g, seq, oracle_jobs.perm_stream(sizes, 500), batch, lr=1e-3, meta_lr=0.5)
    torch.cuda.synchronize()
    gsecs = time.time() - t0
    ora = oracle_jobs.result("amazon6_dn", batch=batch)
    assert trace_g == ora["trace"]
    n_steps = sum(t[2] for t in trace_g)
    forced = int(eng.lib.mamdr_table_flushes(eng.ctx, 1))
    print("amazon6 deepfm DN bs %d: %d domain-steps (%d forced flushes, %d in all); oracle %.1f s (waited %.1f s), hip %.2f s" % (
        batch, n_steps, forced, int(eng.lib.mamdr_table_flushes(eng.ctx, 0)), ora["secs"], ora.get("waited_seconds", 0.0), gsecs))
    assert n_steps >= 150 and forced >= 4
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 0 and int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == n_steps
    eng.set_weights(theta_g)
    worst, aucs = 0.0, []
    for d in range(D):
        _, auc_g = eng.evaluate(d, "val")
        auc_o = ora["aucs"][d]
        print("  domain %d: val rows %6d  AUC hip %.5f oracle %.5f  diff %+.1e" % (d, eng.n_rows(d, "val"), auc_g, auc_o, auc_g - auc_o))
        worst = max(worst, abs(auc_g - auc_o))
        aucs.append(auc_o)
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)
    assert float(np.mean(aucs)) > 0.6, aucs
    # the trained tables themselves (theta after the outer update), all 618 K rows: TF1's dense Adam moved every one
    w_g = eng.unpack(theta_g)
    for name in ("user_emb", "item_emb"):
        a, o = w_g[name].reshape(-1, 128), ora["tables"][name]
        diff = np.abs(a - o)
        print("  %s: |hip - oracle| median %.1e, 99.9 %% %.1e, max %.1e (|oracle| median %.1e)" % (
            name, float(np.median(diff)), float(np.quantile(diff[::7], 0.999)), float(diff.max()),
            float(np.median(np.abs(o)))))
        assert np.isfinite(a).all() and float(np.median(diff)) < 1e-6
    print("  worst |dAUC| %.2e, mean oracle AUC %.4f" % (worst, float(np.mean(aucs))))
    eng.close()


def _star_case(keras_init, phi0="init"):
    from mamdr_amd import engine, meta
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    batch = 8192
    pb = oracle_jobs.problem_amazon13(batch, keras_init)
    g, params, plan, doms, all_sizes, D = (pb[k] for k in ("g", "params", "plan", "doms", "all_sizes", "D"))
    assert (g["n_user"], g["n_item"], D) == (502222, 215403, 13)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.0, emb_trainable=True, tower="star")
    _bind_splits(eng, g, doms)
    full = eng.pack(params)
    eng.set_weights(full)
    theta_g = full[:eng.n_meta].clone()
    assert eng.n_meta < eng.n_params
    if keras_init and phi0 == "init":
        phis_g = {d: torch.from_numpy(oracle_jobs.star_phi0(pb, d, eng.n_meta)).to(eng.device) for d in doms}
    else:
        phis_g = {d: eng.new_vector(meta=True) for d in doms}
    t0 = time.time()
    trace_g = meta.mamdr_epoch(eng, theta_g, phis_g, plan, oracle_jobs.perm_stream(all_sizes, 900), batch, lr=1e-3, meta_lr=0.5)
    torch.cuda.synchronize()
    gsecs = time.time() - t0
    n_steps = sum(t[2] for t in trace_g)
    # (a pass of this plan has at most 15 steps and the DR loop reads / replaces the live rows after every support
    # step, so the replays happen there -- mamdr_sync_tables -- before the 32-step period can force one)
    flushes = int(eng.lib.mamdr_table_flushes(eng.ctx, 0))
    assert n_steps >= 150 and flushes >= 8 and int(eng.lib.mamdr_optimizer_steps(eng.ctx)) == n_steps
    assert int(eng.lib.mamdr_step_path(eng.ctx, batch)) == 0
    merged = eng.new_vector(meta=True)
    auc_g = {}
    for d in doms:
        eng.merge(merged, theta_g, phis_g[d], "plus")
        eng.set_weights(merged)
        auc_g[d] = eng.evaluate(d, "val")[1]
    from oracle import star as ostar
    named = eng.unpack(eng.get_weights())
    tail_g = {n: np.asarray(named[n], np.float32).ravel() for n in ostar.param_names(True)[1]}
    aux = eng.aux_state()
    eng.close()
    return dict(trace=trace_g, aucs=auc_g, tail=tail_g, aux=aux, n_steps=n_steps, flushes=flushes, secs=gsecs, doms=doms)


@pytest.mark.oracle_job("amazon13_star", batch=8192, keras_init=False, perturb=0.0, phi0="init")
def test_amazon13_star_mamdr_epoch_trainable_full_tables():
    """BASELINE.json configs[4] on its own kernel path: star_meta_mamdr, Amazon-13's FULL tables (502,222 + 215,403
    rows x 128, trainable: 91.9 M parameters inside theta / phi_d), bs 8,192 -> k_star_stats / k_star_prep +
    k_tower<train, 384> + PartitionedNorm backward + [k_wgrad + k_emb_reduce + k_emb_rows] + [k_star_update +
    k_emb_catchup], lazy table Adam, lazy per-domain slices (k_star_catchup).  One DN + DR meta-epoch
    (model_zoo/mamdr.py:41-108) on the four largest domains of a row sample (80 % of the rows from 3,000 users / 1,500
    items per domain), theta / phi over the reference's meta filter ["emb", "kernel_shared", "bias_shared"]
    (config/Taobao-10/star_taobao.json:37-41, maml.py:153-179; PartitionedNorm's gamma / beta, the specific kernels and
    the output unit stay live in the model, Star/partitioned_norm.py:102-203), 2 sampled support domains + the query.
    This variant: PartitionedNorm's gamma / beta and the biases moved OFF their special initial values, phi_d = 0 (the
    well-conditioned problem; the Keras-init variant follows).  Oracle (worker process): oracle/loops.mamdr_epoch on
    oracle/star.OracleStar with dense Adam over every table row and every per-domain slice each step.  Asserted: equal
    traces, the table replays between the passes, per-domain val AUC of theta + phi_d within the plain 1e-3, oracle mean
    AUC > 0.6, PartitionedNorm's moving statistics."""
    h = _star_case(False)
    ora = oracle_jobs.result("amazon13_star", batch=8192, keras_init=False, perturb=0.0, phi0="init")
    assert h["trace"] == ora["trace"]
    print("amazon13 star MAMDR bs 8192: %d domain-steps in %d passes (%d table flushes); oracle %.1f s (waited %.1f s), hip %.2f s" % (
        h["n_steps"], len(h["trace"]), h["flushes"], ora["secs"], ora.get("waited_seconds", 0.0), h["secs"]))
    worst, aucs = 0.0, []
    for d in h["doms"]:
        auc_g, auc_o = h["aucs"][d], ora["aucs"][d]
        print("  domain %2d: AUC hip %.5f oracle %.5f  diff %+.1e" % (d, auc_g, auc_o, auc_g - auc_o))
        worst = max(worst, abs(auc_g - auc_o))
        aucs.append(auc_o)
        assert abs(auc_g - auc_o) <= 1e-3, (d, auc_g, auc_o)
    assert float(np.mean(aucs)) > 0.6, aucs
    # PartitionedNorm's moving statistics of the trained domains (non-trainable state, outside theta / phi)
    for d in h["doms"]:
        # (batch means of trained table rows: the rows themselves differ by ~1e-4 after 200 Adam steps at lr 1e-3 --
        # Adam normalises rounding-level gradient differences to steps of that size -- while the means are ~5e-4)
        np.testing.assert_allclose(h["aux"]["mov_mean"][d], ora["mov_mean"][d], rtol=1e-3, atol=3e-4)
        np.testing.assert_allclose(h["aux"]["mov_var"][d], ora["mov_var"][d], rtol=2e-2, atol=1e-6)
    np.testing.assert_array_equal(h["aux"]["steps"], ora["steps"])
    print("  worst |dAUC| %.2e, mean oracle AUC %.4f" % (worst, float(np.mean(aucs))))


KERAS_PERTURB = 2e-7


KERAS_ZERO_SEEDS = TWIN_SEEDS[:3]        # the phi0 = 0 diagnostic: an ensemble of four oracle runs (57 s of CPU each)


def _keras_seeds(phi0):
    return KERAS_ZERO_SEEDS if phi0 == "zero" else TWIN_SEEDS[:1]


def _keras_jobs(phi0):
    return [pytest.mark.oracle_job("amazon13_star", batch=8192, keras_init=True, perturb=0.0, phi0=phi0)] + [
        pytest.mark.oracle_job("amazon13_star", batch=8192, keras_init=True, perturb=KERAS_PERTURB, phi0=phi0, pseed=sd)
        for sd in _keras_seeds(phi0)]


@pytest.mark.parametrize("phi0", [pytest.param("init", marks=_keras_jobs("init")), pytest.param("zero", marks=_keras_jobs("zero"))])
def test_amazon13_star_mamdr_epoch_at_keras_initial_values(phi0):
    """configs[4] from the state the reference really starts from: PartitionedNorm gamma = 1, beta = 0
    (Star/partitioned_norm.py:19-22), zero biases (star_fcn.py:24-25), phi_d = a second random initialisation of the
    model (mamdr.py:31-33).  At beta = 0 the normalised domain columns of a single-domain batch are rounding residue
    of (x - mean), their kernel rows' gradients are noise and Adam turns noise into +- lr steps: the problem is
    ill-conditioned for ANY fp32 evaluation.  The instrument that shows it: a SECOND ORACLE run
    whose initial tensors are perturbed by 2e-7 relative (one fp32 rounding) -- its distance from the first oracle run
    is the oracle's self-divergence, printed beside every domain.  Bar per domain, phi0 "init" (the reference's state):
    north_star's PLAIN |AUC_hip - AUC_oracle| <= 1e-3 (round 6; the measured distances are 3e-5 - 6e-5, the self-divergence
    term of rounds 4 - 5 was never needed).  phi0 "zero" (not a state the reference starts from; one domain 2e-3 off with an
    oracle twin of its own 7.5e-4 off): not an outlier of an ensemble of four oracle runs (tests/ensemble.py).  The tail tensors (outside theta / phi) are reported the same way: relative L2 distance hip-oracle
    next to oracle-oracle'.
    phi0 "init": phi_d as the reference draws it.  phi0 "zero": the starting point of round 4's diagnostic
    (profiles/r04_star13_phases_keras_init.txt: one domain 2.0e-3 off, then unexplained) under the same instrument."""
    h = _star_case(True, phi0)
    ora = oracle_jobs.result("amazon13_star", batch=8192, keras_init=True, perturb=0.0, phi0=phi0)
    twins = [oracle_jobs.result("amazon13_star", batch=8192, keras_init=True, perturb=KERAS_PERTURB, phi0=phi0, pseed=sd)
             for sd in _keras_seeds(phi0)]
    orb = twins[0]
    assert all(h["trace"] == m["trace"] for m in [ora] + twins)
    print("amazon13 star MAMDR at Keras init (phi0 %s), bs 8192: %d domain-steps; oracle %.1f s + perturbed oracle %.1f s, hip %.2f s" % (
        phi0, h["n_steps"], ora["secs"], orb["secs"], h["secs"]))
    worst, aucs = 0.0, []
    for d in h["doms"]:
        a_h, a_o, a_p = h["aucs"][d], ora["aucs"][d], orb["aucs"][d]
        self_div = abs(a_o - a_p)
        print("  domain %2d: AUC hip %.5f oracle %.5f oracle' %.5f | |hip - oracle| %.1e, oracle self-divergence %.1e" % (
            d, a_h, a_o, a_p, abs(a_h - a_o), self_div))
        aucs.append(a_o)
        worst = max(worst, abs(a_h - a_o))
        if phi0 == "init":
            assert abs(a_h - a_o) <= 1e-3, (d, a_h, a_o, a_p)
    if phi0 != "init":
        ens = Ensemble([ora] + twins)
        ens.check("val", h["aucs"], [m["aucs"] for m in [ora] + twins])
        ens.aggregate("amazon13 star at Keras init, phi0 = 0")
    names = sorted(ora["tail"])
    t_h, t_o, t_p = (np.concatenate([x["tail"][n_] for n_ in names]) for x in (h, ora, orb))
    rel_h = float(np.linalg.norm(t_h - t_o) / np.linalg.norm(t_o))
    rel_p = float(np.linalg.norm(t_p - t_o) / np.linalg.norm(t_o))
    print("  tensors outside theta / phi: relative L2 distance hip-oracle %.2e, oracle-oracle' %.2e" % (rel_h, rel_p))
    assert rel_h <= 3 * rel_p + 1e-4, (rel_h, rel_p)
    assert float(np.mean(aucs)) > 0.55, aucs
    np.testing.assert_array_equal(h["aux"]["steps"], ora["steps"])
    print("  largest |hip - oracle|: %.1e (bar 1e-3, plain)" % worst)
