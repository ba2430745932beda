"""Teacher-forced epochs (test infrastructure; VERDICT r05 "Next round" item 1a).

The end-of-training AUC comparisons of tests/test_gpu_fullsize.py / test_gpu_e2e.py are chaotic: after ~1,200 Adam steps two
fp32 evaluations of the same training differ by 1e-4 .. 1e-3 in per-domain AUC whatever the kernels do, so those tests
cannot bound the error of a single step.  Here EVERY pass of a meta-epoch is run on the HIP engine from the ORACLE's state
at that point -- weights, Adam m / v, Adam step count (with TF's running beta powers), position of the dropout stream; a
domain-table step the fused path left pending is materialised first -- and the pass's per-step losses and end state are
compared with the oracle's: the two sides never drift apart by more than one pass (<= ~30 steps on Taobao-10, <= ~20 on
Taobao-30), there is no self-divergence term anywhere, and every kernel instance the epoch's shapes select -- every
domain, every ragged last batch -- is held to the tight bar of the one-step tests.

`Side` is what the harness needs from an engine; `HipSide` drives mamdr_amd.engine.TowerEngine through the C ABI (bound
vectors + mamdr_set_counters), `OracleSide` a second oracle model (the CPU suite's self-check of the harness).
"""
import numpy as np

F32 = np.float32


class OracleSide(object):
    """the harness checked against itself: a second oracle model started from the dumped states must reproduce them."""

    def __init__(self, model, data):
        self.model, self.data = model, data

    def load(self, w, m, v, t, step):
        from oracle import tower as otower
        self.model.set_flat(np.array(w, F32))
        otower.unflatten(np.array(m, F32), self.model.opt.m, self.model.names)
        otower.unflatten(np.array(v, F32), self.model.opt.v, self.model.names)
        self.model.opt.t = int(t)
        self.model.opt.b1p, self.model.opt.b2p = otower.beta_powers(int(t))
        self.model.step = int(step)

    def run(self, d, perm, batch, variant):
        return np.array(self.model.train_pass(self.data[d], perm, batch), F32)

    def state(self):
        from oracle import tower as otower
        m = self.model
        return m.get_flat(), otower.flatten(m.opt.m, m.names), otower.flatten(m.opt.v, m.names)

    def counters(self):
        return int(self.model.opt.t), int(self.model.step)


class HipSide(object):
    """mamdr_amd.engine.TowerEngine: the oracle's flat order <-> the library's flat vector (segments are 16-B aligned,
    padding stays zero) through one index vector; states travel as device tensors."""

    def __init__(self, eng, names, sizes_of):
        import torch
        self.torch, self.eng = torch, eng
        idx = []
        for n in names:
            off, cnt = eng.segments[n]
            assert cnt == sizes_of[n], (n, cnt, sizes_of[n])
            idx.append(np.arange(off, off + cnt, dtype=np.int64))
        self.idx = torch.from_numpy(np.concatenate(idx)).to(eng.device)
        self.perms = {}

    def _put(self, dst, flat):
        t = self.torch.from_numpy(np.ascontiguousarray(flat, F32)).to(self.eng.device)
        dst.index_copy_(0, self.idx, t)

    def load(self, w, m, v, t, step):
        eng = self.eng
        eng.sync()                              # pending domain-table step / lagging rows belong to the state being replaced
        self._put(eng._weights, w)
        self._put(eng._adam_m, m)
        self._put(eng._adam_v, v)
        eng.set_counters(t, step)

    def run(self, d, perm, batch, variant):
        """variant 0: the bench's path -- the pass announced by mamdr_pregather_passes (k_pass_prep_multi), no loss output;
        variant 1: no hint (k_pass_prep inside the call), per-step losses written."""
        torch, eng = self.torch, self.eng
        key = (d, perm.ctypes.data, perm.shape[0])
        if key not in self.perms:
            self.perms = {key: torch.from_numpy(perm).to(eng.device)}
        pd = self.perms[key]
        n_steps = -(-perm.shape[0] // batch)
        if variant == 0:
            eng.pregather([(d, pd)], batch)
            eng.train_steps(d, pd, batch_size=batch)
            return None
        loss = torch.zeros(n_steps, dtype=torch.float32, device=eng.device)
        eng.train_steps(d, pd, batch_size=batch, loss_out=loss)
        return loss.cpu().numpy()

    def state(self):
        eng = self.eng
        eng.sync()
        return tuple(x.index_select(0, self.idx).cpu().numpy() for x in (eng._weights, eng._adam_m, eng._adam_v))

    def counters(self):
        lib, ctx = self.eng.lib, self.eng.ctx
        return int(lib.mamdr_optimizer_steps(ctx)), int(lib.mamdr_dropout_steps(ctx))


def adam_stats(got, want, n_steps, lr):
    """the measures of tests/test_gpu_parity.assert_adam_close: fraction of elements beyond 5 % of k lr, the largest and the
    median difference in units of k lr."""
    diff = np.abs(np.asarray(got, F32).ravel() - np.asarray(want, F32).ravel())
    klr = n_steps * lr
    return float(np.mean(diff > 0.05 * klr)), float(diff.max() / klr), float(np.median(diff) / klr)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run_teacher_forced(side, dump, trace, perm_fn, batch, lr, bars, variants=(0, 1), report=None, exact=False):
    """every pass of `trace` ((phase, domain, n_steps) tuples, the oracle's) on `side` from the dumped start state.
    bars: dict(loss_first, loss_rel, frac, max_klr, med_klr, mv_rel) -- see tests/test_gpu_teacher.py.  -> summary dict."""
    import oracle_jobs
    recs, tail = oracle_jobs.read_dump(dump)
    assert len(trace) == len(dump["meta"]) == recs.shape[0]
    worst = dict(loss_first=0.0, loss_rel=0.0, frac=0.0, max_klr=0.0, med_klr=0.0, m_rel=0.0, v_rel=0.0)
    n_steps_total, ragged = 0, 0
    for k, ((phase, d, n_tr), (t0, s0, n_st)) in enumerate(zip(trace, dump["meta"])):
        assert n_tr == n_st
        perm = perm_fn(d)
        assert perm.shape[0] > (n_st - 1) * batch and perm.shape[0] <= n_st * batch
        ragged += perm.shape[0] % batch != 0
        w0, m0, v0, w1 = recs[k]
        m1, v1 = (recs[k + 1][1], recs[k + 1][2]) if k + 1 < len(trace) else (tail[0], tail[1])
        ends = []
        for variant in variants:
            side.load(w0, m0, v0, t0, s0)
            losses = side.run(d, perm, batch, variant)
            assert side.counters() == (t0 + n_st, s0 + n_st), (k, side.counters(), t0, s0, n_st)
            ends.append(side.state())
            if losses is not None:
                lo = dump["losses"][k]
                assert losses.shape == lo.shape and np.isfinite(losses).all()
                rel = np.abs(losses - lo) / np.maximum(np.abs(lo), 1e-6)
                worst["loss_first"] = max(worst["loss_first"], float(rel[0]))
                worst["loss_rel"] = max(worst["loss_rel"], float(rel.max()))
                assert rel[0] <= bars["loss_first"], ("first step's loss (identical weights)", k, phase, d, losses[0], lo[0])
                assert rel.max() <= bars["loss_rel"], ("loss", k, phase, d, int(rel.argmax()), losses, lo)
        for e in ends[1:]:          # the two launch paths of a pass give the same bits
            for a, b in zip(ends[0], e):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("variants differ", k, phase, d)
        wg, mg, vg = ends[0]
        if exact:
            for a, b in ((wg, w1), (mg, m1), (vg, v1)):
                assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)), (k, phase, d)
        frac, mx, med = adam_stats(wg, w1, n_st, lr)
        mr, vr = rel_l2(mg, m1), rel_l2(vg, v1)
        for key, val in (("frac", frac), ("max_klr", mx), ("med_klr", med), ("m_rel", mr), ("v_rel", vr)):
            worst[key] = max(worst[key], val)
        assert frac <= bars["frac"] and mx <= bars["max_klr"] and med <= bars["med_klr"], ("end weights", k, phase, d, n_st, frac, mx, med)
        assert mr <= bars["mv_rel"] and vr <= bars["mv_rel"], ("Adam slots", k, phase, d, mr, vr)
        n_steps_total += n_st
        if report is not None:
            report(k, phase, d, n_st, frac, mx, med, mr, vr)
    return dict(worst, passes=len(trace), steps=n_steps_total, ragged_passes=int(ragged))
