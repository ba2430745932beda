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
        t = self.torch.from_numpy(np.array(flat, F32)).to(self.eng.device)          # (a copy: the dump is a read-only map)
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
        if self.perms.get("host") is not perm:          # (the host array is kept alive with its device copy)
            self.perms = {"host": perm, "dev": torch.from_numpy(perm).to(eng.device)}
        pd = self.perms["dev"]
        n_steps = -(-perm.shape[0] // batch)
        if variant == 0:
            eng.pregather([(d, pd)], batch)
            eng.train_steps(d, pd, batch_size=batch)
            return None
        eng.pregather([], batch)                # forget variant 0's hint: this call resolves and gathers its pass itself
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


def run_teacher_forced(side, dump, trace, perm_fn, batch, lr, bars, variants=(0, 1), report=None, exact=False, segments=None):
    """every pass of `trace` ((phase, domain, n_steps) tuples, the oracle's) on `side` from the dumped start state.
    bars: dict(loss_first, loss_rel, frac, max_klr, med_klr, m_rel, v_rel) -- see tests/test_gpu_teacher.py.  Violations are
    collected over the WHOLE epoch and raised together (a GPU run is too dear to stop at the first one); `segments`
    ([(name, offset, count)] in the oracle's flat order) adds a per-tensor breakdown to a violation's record.
    -> summary dict."""
    import oracle_jobs
    recs, tail = oracle_jobs.read_dump(dump)
    assert len(trace) == len(dump["meta"]) == recs.shape[0]
    worst = dict(loss_first=0.0, loss_rel=0.0, frac=0.0, max_klr=0.0, med_klr=0.0, m_rel=0.0, v_rel=0.0)
    n_steps_total, ragged, bad = 0, 0, []

    def breakdown(got, want):
        if not segments:
            return ""
        rows = [(rel_l2(got[o:o + c], want[o:o + c]), n) for n, o, c in segments]
        return " ".join("%s %.1e" % (n, r) for r, n in sorted(rows, reverse=True)[:4])
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
                if rel[0] > bars["loss_first"]:
                    bad.append(("first step's loss (identical weights)", k, phase, d, float(losses[0]), float(lo[0])))
                if rel.max() > bars["loss_rel"]:
                    bad.append(("loss", k, phase, d, int(rel.argmax()), float(rel.max())))
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
        if not (frac <= bars["frac"] and mx <= bars["max_klr"] and med <= bars["med_klr"]):
            bad.append(("end weights", k, phase, d, n_st, perm.shape[0], frac, mx, med, breakdown(wg, w1)))
        if not (mr <= bars["m_rel"] and vr <= bars["v_rel"]):
            bad.append(("Adam slots", k, phase, d, n_st, perm.shape[0], mr, vr, "m: " + breakdown(mg, m1), "v: " + breakdown(vg, v1)))
        n_steps_total += n_st
        if report is not None:
            report(k, phase, d, n_st, frac, mx, med, mr, vr)
    out = dict(worst, passes=len(trace), steps=n_steps_total, ragged_passes=int(ragged), violations=bad)
    return out


# ---------------------------------------------------------------------------------------------- trainable FULL-size tables
class LockStep(object):
    """Teacher forcing without a dump: the oracle model runs INSIDE the test process and every `train_pass` of the loop is
    mirrored on the HIP engine from the oracle's state at that moment (configs[2] / configs[4]: the state is 1 - 1.1 GB per
    pass with the trainable tables' Adam slots -- uploaded, not stored).  Per pass:
        oracle state (every tensor, Adam m / v, counters, Star: PartitionedNorm's moving statistics) -> HIP engine
        HIP pass queued (per-step losses into a device buffer)  ||  oracle pass on the CPU
        oracle end state -> device, compared tensor by tensor with the HIP engine's (mamdr_sync_tables first: lagging rows
        of the lazy table Adam are replayed, which is part of what is being checked).
    `inner`: the object the loop calls (the model, or its meta view for Star); `model`: the oracle model itself."""

    def __init__(self, inner, model, eng, data, lr, bars, aux_of=None, chunk=8, noise_slots=(), slots_of=None, t_of=None,
                 oracle_pass=None):
        """noise_slots: tensors whose GRADIENT is rounding residue by construction (Star: the domain table -- under
        PartitionedNorm a single-domain batch's domain columns are constant, their normalised values and hence the row's
        gradient are what (x - mean) leaves of equal numbers): their Adam slots are averages of noise on both sides and are
        not compared; their weights are (Adam moves them by at most lr per step whatever the noise)."""
        import torch
        self.torch, self.chunk, self.noise_slots = torch, int(chunk), tuple(noise_slots)
        # (the model's Adam slots / step count / pass call: oracle/tower.OracleModel's by default; oracle/mtl.OracleMTL keeps
        # them on the model itself and takes the domain as an argument of its pass)
        self.slots_of = slots_of or (lambda m: (m.opt.m, m.opt.v))
        self.t_of = t_of or (lambda m: int(m.opt.t))
        self.oracle_pass = oracle_pass or (lambda d, data, perm, bs: self.inner.train_pass(data, perm, bs))
        self.inner, self.model, self.eng, self.lr, self.bars, self.aux_of = inner, model, eng, lr, bars, aux_of
        self.dom_of = {id(cols): d for d, cols in data.items()}
        self.rows, self.bad = [], []

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def _dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, F32).ravel()).to(self.eng.device)

    def _load(self):
        eng, m = self.eng, self.model
        self._sync()
        om, ov = self.slots_of(m)
        for n in m.names:
            off, cnt = eng.segments[n]
            assert cnt == m.params[n].size, (n, cnt, m.params[n].size)
            eng._weights[off:off + cnt].copy_(self._dev(m.params[n]))
            eng._adam_m[off:off + cnt].copy_(self._dev(om[n]))
            eng._adam_v[off:off + cnt].copy_(self._dev(ov[n]))
        if self.aux_of is not None:
            a = self._dev(self.aux_of(m))
            eng.aux[:a.numel()].copy_(a)            # (the library pads its state vector to 16 B)
        eng.set_counters(self.t_of(m), int(m.step))

    def _sync(self):
        if hasattr(self.eng, "sync"):
            self.eng.sync()

    def _counters(self):
        eng, lib = self.eng, self.eng.lib
        if hasattr(lib, "mamdr_graph_optimizer_steps") and type(eng).__name__ == "GraphEngine":
            return int(lib.mamdr_graph_optimizer_steps(eng.ctx)), int(lib.mamdr_graph_dropout_steps(eng.ctx))
        return int(lib.mamdr_optimizer_steps(eng.ctx)), int(lib.mamdr_dropout_steps(eng.ctx))

    def train_pass(self, data, perm, batch_size, max_steps=0, accumulate_into=None):
        """the pass in CHUNKS of `self.chunk` steps, each restarted from the oracle's state: with trainable tables that start
        at N(0, 1e-4^2) under Adam at lr 1e-3 a pass amplifies a rounding-level difference by ~1.15 per step (the oracle
        against its own copy perturbed by one rounding: 4e-7 -> 1.5e-3 in the loss over the 66 steps of Amazon-6's longest
        pass, profiles/r06_teacher_probe.txt) -- per-pass forcing cannot carry a tight bar there, 8-step chunks can."""
        assert accumulate_into is None
        d = self.dom_of[id(data)]
        n_steps = -(-perm.shape[0] // batch_size)
        if max_steps > 0:
            n_steps = min(n_steps, max_steps)
        perm_d = self.torch.from_numpy(perm).to(self.eng.device)
        losses = []
        for s0 in range(0, n_steps, self.chunk):
            c = min(self.chunk, n_steps - s0)
            losses += self._chunk(data, d, perm, perm_d, batch_size, s0, c)
        return losses

    def run_pass(self, d, data, perm, batch_size):
        """a pass over domain d outside oracle/loops.py (the test drives the loop itself)."""
        self.dom_of[id(data)] = d
        return self.train_pass(data, perm, batch_size)

    def _chunk(self, data, d, perm, perm_d, batch_size, s0, n_steps):
        torch, eng, m = self.torch, self.eng, self.model
        aux = getattr(eng, "aux", None)
        t0, s0c = self.t_of(m), int(m.step)
        # launch path A -- what a training run executes: no loss output, i.e. with trainable tables the LAZY table Adam (a
        # non-null loss buffer makes every step synchronise the tables first: include/mamdr_hip.h)
        self._load()
        eng.train_steps(d, perm_d, first_step=s0, n_steps=n_steps, batch_size=batch_size, lr=self.lr)
        self._sync()
        snap = [x.clone() for x in (eng._weights, eng._adam_m, eng._adam_v)] + ([aux.clone()] if aux is not None else [])
        # launch path B -- the same steps again from the same state with the per-step losses written: same bits at the end
        self._load()
        loss_g = torch.zeros(n_steps, dtype=torch.float32, device=eng.device)
        eng.train_steps(d, perm_d, first_step=s0, n_steps=n_steps, batch_size=batch_size, loss_out=loss_g, lr=self.lr)
        self._sync()
        live = [eng._weights, eng._adam_m, eng._adam_v] + ([aux] if aux is not None else [])
        for a, b in zip(snap, live):
            if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
                self.bad.append(("launch paths differ", len(self.rows), d, n_steps, int((a.view(torch.int32) != b.view(torch.int32)).sum())))
        del snap
        sub = perm[s0 * batch_size:(s0 + n_steps) * batch_size]
        losses = self.oracle_pass(d, data, sub, batch_size)             # the oracle's steps
        assert len(losses) == n_steps
        assert self._counters() == (t0 + n_steps, s0c + n_steps) == (self.t_of(m), int(m.step))
        lg, lo = loss_g.cpu().numpy(), np.array(losses, F32)
        rel = np.abs(lg - lo) / np.maximum(np.abs(lo), 1e-6)
        k = len(self.rows)
        if rel[0] > self.bars["loss_first"] or rel.max() > self.bars["loss_rel"]:
            self.bad.append(("loss", k, d, n_steps, float(rel[0]), float(rel.max())))
        klr = n_steps * self.lr
        worst = dict(frac=0.0, max_klr=0.0, med_klr=0.0, m_rel=0.0, v_rel=0.0)
        for n in m.names:
            off, cnt = eng.segments[n]
            diff = (eng._weights[off:off + cnt] - self._dev(m.params[n])).abs_()
            frac, mx = float((diff > 0.05 * klr).float().mean()), float(diff.max()) / klr
            med = float(diff.median()) / klr
            rels = []
            om, ov = self.slots_of(m)
            for got, want in ((eng._adam_m, om[n]), (eng._adam_v, ov[n])):
                w = self._dev(want)
                rels.append(float((got[off:off + cnt] - w).double().norm() / max(float(w.double().norm()), 1e-30)))
            del diff
            if not (frac <= self.bars["frac"] and mx <= self.bars["max_klr"] and med <= self.bars["med_klr"]):
                self.bad.append(("end weights", k, d, n_steps, n, frac, mx, med))
            if n in self.noise_slots:
                rels = [0.0, 0.0]
            if not (rels[0] <= self.bars["m_rel"] and rels[1] <= self.bars["v_rel"]):
                self.bad.append(("Adam slots", k, d, n_steps, n, rels[0], rels[1]))
            for key, val in (("frac", frac), ("max_klr", mx), ("med_klr", med), ("m_rel", rels[0]), ("v_rel", rels[1])):
                worst[key] = max(worst[key], val)
        if self.aux_of is not None:
            a_o = self._dev(self.aux_of(m))
            worst["aux_rel"] = float((eng.aux[:a_o.numel()] - a_o).double().norm() / max(float(a_o.double().norm()), 1e-30))
            if worst["aux_rel"] > self.bars.get("aux_rel", 1e-4):
                self.bad.append(("moving statistics", k, d, n_steps, worst["aux_rel"]))
        last = s0 + n_steps >= -(-perm.shape[0] // batch_size)
        self.rows.append(dict(worst, k=k, d=d, n=n_steps, first=s0, rows=int(min(perm.shape[0], (s0 + n_steps) * batch_size) - s0 * batch_size),
                              ragged=bool(last and perm.shape[0] % batch_size), loss_first=float(rel[0]), loss_rel=float(rel.max())))
        return losses

    def summary(self):
        keys = ("loss_first", "loss_rel", "frac", "max_klr", "med_klr", "m_rel", "v_rel") + (("aux_rel",) if self.aux_of else ())
        return dict({key: max(r[key] for r in self.rows) for key in keys}, chunks=len(self.rows),
                    passes=sum(1 for r in self.rows if r["first"] == 0), steps=sum(r["n"] for r in self.rows),
                    ragged_passes=sum(1 for r in self.rows if r["ragged"]))
