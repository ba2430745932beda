"""Meta-loop epochs on the device-resident tower (host control flow only).

These are the numeric cores of the reference's wrapper `train()` loops with the
python-RNG decisions factored out: the caller supplies the domain order, the DR
support domains and the per-pass shuffle, so that the loop is reproducible and
shardable.  Loop structure follows

    alternate_epoch  model_zoo/DeepCTR/deepctr.py:70-78
    dn_epoch         model_zoo/domain_negotiation.py:49-88
    reptile_epoch    model_zoo/reptile.py:45-99
    mamdr_epoch      model_zoo/mamdr.py:44-108   (DN phase then DR phase)
    maml_epoch       model_zoo/maml.py:62-116    (first-order MAML, outer Adam)

Weights never leave the GPU: the reference's K.batch_get_value -> numpy ->
SetVarOp round trips (maml.py:189-194, utils/tool.py:36-45) become device copies
and elementwise kernels on flat vectors.  Every function returns the trace of
(phase, domain, n_steps) it executed.
"""
import torch


def _upload_perm(eng, perm):
    if perm is None:
        return None
    if isinstance(perm, torch.Tensor):
        return perm
    return torch.from_numpy(perm).to(eng.device, non_blocking=True)


class PassWindow(object):
    """The passes a loop is about to run, announced ahead: where the engine would resolve and gather every pass's rows
    at the start of its own call (frozen tables on the fused step path: one launch per call, 7 us of Taobao-10's
    ~210 us passes) they are gathered for a whole window of passes in ONE launch (engine.pregather ->
    mamdr_pregather_passes).  Needs a perm_fn that can show its next permutations without consuming them
    (plan.EpochShuffles.peek); anything else -- and any engine without the hint -- runs as before.  A window holds at
    most `max_passes` passes and `budget_rows` rows (1 KB each: the gathered rows should still sit in the 256 MB
    infinity cache when their steps read them).  Results do not depend on it (same rows, same bits)."""

    def __init__(self, eng, perm_fn, batch_size, budget_rows=96 * 1024, max_passes=16):
        self.eng, self.perm_fn, self.batch_size = eng, perm_fn, batch_size
        self.budget, self.max_passes = budget_rows, max_passes
        import os
        self.on = hasattr(perm_fn, "peek") and hasattr(eng, "pregather") and not os.environ.get("MAMDR_NO_PASS_WINDOW")
        if os.environ.get("MAMDR_PASS_WINDOW_ROWS"):          # (A/B measurements of the window size)
            self.budget = int(os.environ["MAMDR_PASS_WINDOW_ROWS"])
        self.todo, self.left = [], 0

    def announce(self, domains):
        self.todo, self.left = list(domains), 0

    def step(self):
        """right before every announced pass"""
        if not self.on:
            return
        if self.left == 0 and self.todo:
            take, rows = 0, 0
            while take < len(self.todo) and take < self.max_passes:
                n = self.eng.n_rows(self.todo[take], "train")
                if take and rows + n > self.budget:
                    break
                rows += n
                take += 1
            chunk, self.todo = self.todo[:take], self.todo[take:]
            perms = self.perm_fn.peek(chunk)
            if perms is None:          # the loop is not following the announced order: no hint
                self.on = False
                return
            # (a window of one pass is what the call does by itself: an empty hint, so that no entry of an earlier
            # window -- same device buffer, other contents two epochs later -- can ever be taken for this pass)
            self.eng.pregather(list(zip(chunk, perms)) if take > 1 else [], self.batch_size)
            self.left = take
        self.left = max(self.left - 1, 0)


def run_pass(eng, d, perm_fn, batch_size, lr, trace, phase, max_steps=0, optimizer="adam", window=None):
    """one pass over domain d's train split = re-initialised iterator + n_step x train_on_batch.
    window = (begin, end): the pass covers that file-order slice only (meta-train / meta-val split)."""
    if window is None:
        perm = _upload_perm(eng, perm_fn(d) if perm_fn is not None else None)
        n, pass_rows = eng.n_rows(d, "train"), None
    else:
        import numpy as np
        perm = perm_fn(d, window) if perm_fn is not None else None
        if perm is None:
            perm = np.arange(window[0], window[1], dtype=np.int32)
        perm = _upload_perm(eng, perm)
        n = pass_rows = window[1] - window[0]
    n_steps = -(-n // batch_size)
    if max_steps and max_steps > 0:
        n_steps = min(n_steps, max_steps)
    if pass_rows is None:
        eng.train_steps(d, perm=perm, first_step=0, n_steps=n_steps, lr=lr, optimizer=optimizer,
                        batch_size=batch_size)
    else:
        eng.train_steps(d, perm=perm, first_step=0, n_steps=n_steps, lr=lr, optimizer=optimizer,
                        batch_size=batch_size, pass_rows=pass_rows)
    trace.append((phase, d, n_steps))
    return n_steps


def alternate_epoch(eng, seq, perm_fn, batch_size, lr):
    trace = []
    for d in seq:
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "alt")
    return trace


def dn_epoch(eng, theta, seq, perm_fn, batch_size, lr, meta_lr, meta_train_step=0, target=-1):
    """target >= 0 (domain_negotiation.py:44-45,67,89-93): the target domain closes the inner sequence with an
    uncapped pass, and after the outer update the model (not theta) takes one more full pass over it."""
    trace = []
    eng.assign_meta(theta)
    for d in seq:
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn", meta_train_step)
    if target >= 0:
        run_pass(eng, target, perm_fn, batch_size, lr, trace, "dn")
    eng.interp(theta, eng.meta_weights, theta, meta_lr)     # theta += (theta~ - theta) * beta
    eng.assign_meta(theta)
    if target >= 0:
        run_pass(eng, target, perm_fn, batch_size, lr, trace, "target")
    return trace


def reptile_epoch(eng, theta, seq, perm_fn, batch_size, lr, meta_lr, batch_variant=False, meta_train_step=0,
                  acc=None, target=-1):
    """target >= 0 (reptile.py:47-48,82-85,98-102): the target domain is skipped in the sequence; every
    domain's pass is followed by ONE step on the target domain's freshly shuffled iterator before the outer
    update, and the epoch ends with a full pass of the model (not theta) over the target domain."""
    trace = []
    if batch_variant and acc is None:
        acc = torch.zeros_like(theta)
    for d in seq:
        if target >= 0 and d == target:
            continue
        eng.assign_meta(theta)
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "reptile", meta_train_step)
        if target >= 0:
            run_pass(eng, target, perm_fn, batch_size, lr, trace, "target_step", 1)
        if batch_variant:
            eng.accumulate(acc, eng.meta_weights, theta)
        else:
            eng.interp(theta, eng.meta_weights, theta, meta_lr)
    if batch_variant:
        eng.apply_accumulated(theta, acc, 0.0, meta_lr)
    eng.assign_meta(theta)
    if target >= 0:
        run_pass(eng, target, perm_fn, batch_size, lr, trace, "target")
    return trace


class OuterAdamState(object):
    """slots of MAML's separate outer tf.train.AdamOptimizer (maml.py:201): device m, v and the
    fp32 running beta powers TF keeps as variables."""

    def __init__(self, eng):
        import numpy as np
        self.m = eng.new_vector()
        self.v = eng.new_vector()
        self.b1p = np.float32(1.0)
        self.b2p = np.float32(1.0)

    def apply(self, eng, theta, acc, lr, grad_scale=1.0, clear=True):
        """theta (n_meta or n_params floats) takes one outer Adam step with the gradient `acc`; a full-size accumulator
        under a meta range contributes its slice of that range (the slots are indexed like the full vector)."""
        import numpy as np
        self.b1p = np.float32(self.b1p * np.float32(0.9))
        self.b2p = np.float32(self.b2p * np.float32(0.999))
        n = theta.numel()
        off = getattr(eng, "meta_off", 0) if (n != acc.numel() or n != self.m.numel()) else 0
        g = acc if acc.numel() == n else acc[off:off + n]
        m = self.m if self.m.numel() == n else self.m[off:off + n]
        v = self.v if self.v.numel() == n else self.v[off:off + n]
        eng.adam_apply(theta, m, v, g, lr, float(self.b1p), float(self.b2p), grad_scale)
        if clear:
            acc.zero_()


def maml_epoch(eng, theta, outer, acc, seq, perm_fn, batch_size, lr, meta_lr, batch_variant=False,
               meta_train_step=0, grad_scale=1.0, windows=None, meta_domain=-1):
    """acc must be bound with eng.bind_accumulator(acc) and zero on entry.
    windows: {domain: (train window, meta window)} for the meta-train / meta-val split, None = train-train.
    meta_domain >= 0 (train.target_domain, maml.py:336-338): every meta pass runs over that domain's whole
    train split instead of the domain's own meta set."""
    trace = []
    for d in seq:
        wt, wm = windows[d] if windows else (None, None)
        dm = d
        if meta_domain >= 0:
            dm, wm = meta_domain, None
        eng.assign_meta(theta)
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "maml_train", meta_train_step, window=wt)
        run_pass(eng, dm, perm_fn, batch_size, lr, trace, "maml_meta", meta_train_step, optimizer="accumulate",
                 window=wm)
        if not batch_variant:
            outer.apply(eng, theta, acc, meta_lr, grad_scale)
    if batch_variant:
        outer.apply(eng, theta, acc, meta_lr, grad_scale)
    eng.assign_meta(theta)
    return trace


def mldg_epoch(eng, theta, outer, acc, seq, perm_fn, batch_size, lr, meta_lr, batch_variant=False,
               meta_train_step=0, grad_scale=1.0, windows=None, meta_domain=-1):
    """MLDG as the reference implements it (model_zoo/mldg.py:62-125): per domain the model is reset to
    theta, the meta-train pass only ACCUMULATES d total_loss / d theta (no inner optimiser step), the outer
    Adam moves the live model by that gradient (accumulator kept), the meta-val pass adds the gradients at
    the moved weights, then the model is reset to theta and the outer Adam applies the sum -> new theta
    (per domain, or once per epoch for "batch" names).  One outer Adam (meta_learning_rate) for both applies.
    `lr` is unused by the passes (accumulate mode) and only kept for symmetry."""
    trace = []
    for d in seq:
        wt, wm = windows[d] if windows else (None, None)
        eng.assign_meta(theta)
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "mldg_train", meta_train_step, optimizer="accumulate",
                 window=wt)
        live = eng.meta_weights.clone()
        outer.apply(eng, live, acc, meta_lr, grad_scale, clear=False)
        eng.assign_meta(live)
        dm = d
        if meta_domain >= 0:            # train.target_domain (mldg.py:339-341): the meta pass runs over the target domain
            dm, wm = meta_domain, None
        run_pass(eng, dm, perm_fn, batch_size, lr, trace, "mldg_meta", meta_train_step, optimizer="accumulate",
                 window=wm)
        if not batch_variant:
            eng.assign_meta(theta)
            outer.apply(eng, theta, acc, meta_lr, grad_scale)
    if batch_variant:
        eng.assign_meta(theta)
        outer.apply(eng, theta, acc, meta_lr, grad_scale)
    eng.assign_meta(theta)
    return trace


def pcgrad_epoch(eng, outer, cur, aux, seq, aux_plan, perm_fn, batch_size, lr, meta_lr, meta_train_step=0,
                 grad_scale=1.0, windows=None):
    """PCGrad as the reference implements it (model_zoo/pcgrad.py:62-124): the model is NOT reset between
    domains (theta is the live model).  Per domain: accumulate d total_loss / d theta over its (meta-)train
    pass at the current weights; for every sampled auxiliary domain accumulate its gradient at the same weights
    and project it onto the running gradient (mamdr_pcgrad_project: pcgrad.py:152-160 with final is current);
    one outer-Adam step of the live model with the result.  cur / aux: two full-size flat vectors.
    aux_plan = {domain: [auxiliary domains]}."""
    trace = []
    # the projection and the outer step cover the meta parameters only (pcgrad.py:152-160 walks `self.model_meta_parms`)
    tensors = None
    if eng.n_meta != eng.n_params and hasattr(eng, "segment_shapes"):
        shapes = eng.segment_shapes()
        lo, hi = eng.meta_off, eng.meta_off + eng.n_meta
        tensors = [(off, shapes[n][0], shapes[n][1]) for n, (off, cnt) in eng.segments.items() if off >= lo and off + cnt <= hi]
    for d in seq:
        eng.bind_accumulator(cur)
        cur.zero_()
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "pcgrad_query", meta_train_step, optimizer="accumulate",
                 window=windows[d][0] if windows else None)
        for a in aux_plan[d]:
            eng.bind_accumulator(aux)
            aux.zero_()
            run_pass(eng, a, perm_fn, batch_size, lr, trace, "pcgrad_aux", 0, optimizer="accumulate",
                     window=windows[a][0] if windows else None)
            eng.pcgrad_project(cur, aux, tensors)
        live = eng.meta_weights.clone()
        outer.apply(eng, live, cur, meta_lr, grad_scale)
        eng.assign_meta(live)
    return trace


def dr_query(eng, theta, phi, query, support, perm_fn, batch_size, lr, meta_lr, trace, merged,
             merged_method="plus", domain_regulation_step=0, batch_variant=False, sample_num=None, acc=None, pw=None):
    """DR for one query domain (mamdr.py:60-108): phi is updated in place.  Reads theta
    (fixed during DR) and writes only phi -- the unit that shards across GPUs."""
    eng.merge(merged, theta, phi, merged_method)
    if batch_variant:
        acc.zero_()
    assigned = False
    if pw is None:
        pw = PassWindow(eng, perm_fn, batch_size)
    # (a query pass capped to `domain_regulation_step` steps runs over the first few batches of its shuffle only: announced,
    # ALL of the query domain's rows would be gathered once per support domain and evict useful rows from the window --
    # ADVICE r04; no hint then, every call gathers exactly the rows its steps read)
    capped = bool(domain_regulation_step and domain_regulation_step > 0)
    pw.announce([] if capped else [x for j in support for x in (j, query)])
    for k, j in enumerate(support):
        if not assigned:
            eng.assign_meta(merged)
        pw.step()
        run_pass(eng, j, perm_fn, batch_size, lr, trace, "dr_support")
        pw.step()
        run_pass(eng, query, perm_fn, batch_size, lr, trace, "dr_query", domain_regulation_step)
        if batch_variant:
            shared = theta if merged_method == "times" else None
            eng.accumulate(acc, eng.meta_weights, merged, shared, 1.0)
        else:
            # phi += (theta~ - merged) * gamma; merged = theta (+|*) phi; model := merged for the next support --
            # one pass over the vectors instead of three launches (bit-identical to interp + merge + set_weights)
            assigned = k + 1 < len(support)
            eng.dr_advance(phi, merged, theta, meta_lr, merged_method, assign_model=assigned)
    if batch_variant:
        eng.apply_accumulated(phi, acc, float(sample_num), meta_lr)


def finetune_query(eng, theta, phi, query, perm_fn, batch_size, lr, trace, merged, merged_method="plus"):
    """train.finetune_every_epoch (mamdr.py:110-143): after a query domain's DR the merged model takes one
    full pass over that domain and phi := theta~ - merged (`_update_domain_weights`, mamdr.py:168-171)."""
    eng.merge(merged, theta, phi, merged_method)
    eng.assign_meta(merged)
    run_pass(eng, query, perm_fn, batch_size, lr, trace, "dr_finetune")
    eng.sub(phi, eng.meta_weights, merged)


def mamdr_epoch(eng, theta, phis, plan, perm_fn, batch_size, lr, meta_lr, merged_method="plus",
                domain_regulation_step=0, batch_variant=False, sample_num=None, scratch=None,
                finetune_every_epoch=False):
    """plan = {"seq": [...], "dr": [(query, [support...]), ...]}."""
    trace = []
    pw = PassWindow(eng, perm_fn, batch_size)
    # DN phase (mamdr.py:48-57)
    eng.assign_meta(theta)
    pw.announce(plan["seq"])
    for d in plan["seq"]:
        pw.step()
        run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn")
    eng.interp(theta, eng.meta_weights, theta, meta_lr)
    # DR phase (mamdr.py:59-108)
    merged = scratch if scratch is not None else torch.empty_like(theta)
    acc = torch.zeros_like(theta) if batch_variant else None
    for query, support in plan["dr"]:
        dr_query(eng, theta, phis[query], query, support, perm_fn, batch_size, lr, meta_lr, trace, merged,
                 merged_method, domain_regulation_step, batch_variant, sample_num, acc, pw)
        if finetune_every_epoch:
            finetune_query(eng, theta, phis[query], query, perm_fn, batch_size, lr, trace, merged, merged_method)
    return trace
