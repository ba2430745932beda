"""Domain sharding of the meta loops over the GPUs of one node (one process per GPU).

SURVEY.md section 8e: the reference is single-process; the paper's PS-worker
scheme (slides p.22) sums per-worker displacements.  Here

* DR shards by query domain: phi_i updates for different i read theta (fixed
  during DR) and write only phi_i -> no collective in the DR phase.  A query
  domain has a fixed owner rank for the whole run (LPT on its train rows), so
  phi_i and its Adam-free state never move.
* DN splits the shuffled sequence into per-rank sub-sequences run from the same
  theta; the only data-path collective is ONE all-reduce (sum, fp32) of the
  displacement theta~_g - theta per epoch over RCCL/xGMI, then
  theta += beta * sum_g (theta~_g - theta).  With one rank this is exactly the
  reference's DN update (domain_negotiation.py:118-123).
* eval is per domain: owners evaluate, scalars are all-gathered.

torch.distributed (backend nccl = RCCL on ROCm, gloo in CPU tests) is plumbing.
"""
import torch
import torch.distributed as dist


def lpt_partition(costs, n_parts):
    """longest-processing-time assignment: returns owner[i] for each item."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_parts
    owner = [0] * len(costs)
    for i in order:
        r = min(range(n_parts), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_plan(plan, owner, rank):
    """this rank's part of an epoch plan: its DN sub-sequence (order preserved) and the DR
    entries of the query domains it owns."""
    return {"seq": [d for d in plan["seq"] if owner[d] == rank],
            "dr": [(q, s) for (q, s) in plan["dr"] if owner[q] == rank]}


def allreduce_delta(eng, theta, delta_buf):
    """delta = live - theta on every rank; sum over ranks (ONE collective)."""
    eng.sub(delta_buf, eng.weights, theta)
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(delta_buf, op=dist.ReduceOp.SUM)
    return delta_buf


def dn_phase_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, trace, delta_buf, zero_buf,
                     meta_train_step=0, target=-1):
    """DN phase of one epoch on this rank's sub-sequence + the all-reduce outer update.
    theta += (sum_g delta_g) * beta, evaluated as interp(theta, delta, 0, beta).
    meta_train_step caps every pass (domain_negotiation.py:67); a target domain (:44-45,89-93) closes EVERY rank's
    sub-sequence with an uncapped pass -- each displacement then ends adapted to the target, as the single
    sequence's does -- and the caller runs the closing target pass on the updated model (identical on every rank)."""
    eng.set_weights(theta)
    for d in seq_local:
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn", meta_train_step)
    if target >= 0:
        meta.run_pass(eng, target, perm_fn, batch_size, lr, trace, "dn")
    rank, ws = world()
    if ws == 1:
        eng.interp(theta, eng.weights, theta, meta_lr)
    else:
        allreduce_delta(eng, theta, delta_buf)
        eng.interp(theta, delta_buf, zero_buf, meta_lr)


def reptile_epoch_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, delta_buf, zero_buf,
                          meta_train_step=0):
    """Reptile, per-domain variant (reptile.py:45-99): the reference interpolates theta after EVERY domain, a
    sequential recurrence.  Sharded: every rank runs that recurrence over its own domains on a private copy
    starting from the epoch's theta, then the ranks' total displacements are summed (ONE all-reduce) and applied:
    theta += sum_g (theta_g - theta).  One rank: exactly the reference's epoch."""
    trace = []
    local = theta.clone()
    for d in seq_local:
        eng.set_weights(local)
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "reptile", meta_train_step)
        eng.interp(local, eng.weights, local, meta_lr)
    rank, ws = world()
    if ws == 1:
        theta.copy_(local)
    else:
        eng.sub(delta_buf, local, theta)
        dist.all_reduce(delta_buf, op=dist.ReduceOp.SUM)
        eng.interp(theta, delta_buf, zero_buf, 1.0)
    eng.set_weights(theta)
    return trace


def reptile_batch_epoch_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, acc, meta_train_step=0):
    """Reptile, batch variant (reptile.py:87-96,134-142): every domain starts from theta and adds its
    displacement theta~ - theta to `acc`; the epoch applies theta += beta * sum.  The sum over domains is a sum
    over ranks of per-rank sums: ONE all-reduce of `acc` per epoch, no other change to the algorithm.
    Only the optimiser slots differ from the single-process run: each rank's Adam moments see its own domains."""
    trace = []
    acc.zero_()
    for d in seq_local:
        eng.set_weights(theta)
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "reptile", meta_train_step)
        eng.accumulate(acc, eng.weights, theta)
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    eng.apply_accumulated(theta, acc, 0.0, meta_lr)
    eng.set_weights(theta)
    return trace


def mamdr_epoch_sharded(eng, meta, theta, phis, plan, owner, perm_fn, batch_size, lr, meta_lr, bufs,
                        merged_method="plus", domain_regulation_step=0):
    """one DN+DR epoch; `phis` holds only the vectors this rank owns (dict domain -> vector)."""
    rank, ws = world()
    local = shard_plan(plan, owner, rank)
    trace = []
    dn_phase_sharded(eng, meta, theta, local["seq"], perm_fn, batch_size, lr, meta_lr, trace, bufs["delta"],
                     bufs["zero"])
    for query, support in local["dr"]:
        meta.dr_query(eng, theta, phis[query], query, support, perm_fn, batch_size, lr, meta_lr, trace,
                      bufs["merged"], merged_method, domain_regulation_step)
    return trace


def epoch_assignment(plan, steps_per_domain, n_parts, domain_regulation_step=0):
    """per-epoch balance of one DN + DR epoch over the ranks (hard part 6): the cost of query domain
    i is what its DR will execute with THIS epoch's sampled supports, sum_j (steps_j + query steps_i)
    (mamdr.py:72-108); queries go to ranks by longest-processing-time, then the DN passes (cost steps_d) fill up
    the least-loaded ranks.  Every rank computes the same assignment from the same plan.
    -> (dr_owner {query: rank}, dn_owner {domain: rank}, load per rank)"""
    def qsteps(i):
        s = steps_per_domain[i]
        return min(s, domain_regulation_step) if domain_regulation_step and domain_regulation_step > 0 else s
    cost = {q: sum(steps_per_domain[j] + qsteps(q) for j in support) for q, support in plan["dr"]}
    load = [0.0] * n_parts
    dr_owner, dn_owner = {}, {}
    for q in sorted(cost, key=lambda i: (-cost[i], i)):
        r = min(range(n_parts), key=lambda k: (load[k], k))
        dr_owner[q] = r
        load[r] += cost[q]
    for d in sorted(plan["seq"], key=lambda i: (-steps_per_domain[i], i)):
        r = min(range(n_parts), key=lambda k: (load[k], k))
        dn_owner[d] = r
        load[r] += steps_per_domain[d]
    return dr_owner, dn_owner, load


class BalancedMAMDR(object):
    """DN + DR epochs sharded over the ranks with a per-epoch assignment and ONE collective per epoch.

    Every rank keeps theta and ALL phi_d in one packed buffer [delta | phi_0 | ... | phi_{D-1}].  An epoch:
      1. the plan (same seed everywhere) -> epoch_assignment -> this rank's DN sub-sequence and DR queries;
      2. DN passes from theta; delta = theta~ - theta;
      3. the phi slots this rank did NOT update in the previous epoch are zeroed and the whole buffer is
         all-reduced (sum): delta becomes sum_g delta_g and every phi slot its last owner's value -- the phi
         hand-over to this epoch's owners rides in the DN collective, so an epoch still has exactly one;
      4. theta += beta * sum_g delta_g;  5. DR of the owned queries (meta.dr_query), phi slots updated in place.
    With one rank no collective runs and the epoch is meta.mamdr_epoch's (the reference's loop).
    `sync_phis()` makes every slot current everywhere (before validation / checkpoints)."""

    def __init__(self, eng, meta, theta, phis, steps_per_domain):
        """phis: {domain: vector}, the SAME initial values on every rank (every rank draws all D initialisations)."""
        self.eng, self.meta, self.theta = eng, meta, theta
        self.steps = list(steps_per_domain)
        self.domains = sorted(phis)
        P = theta.numel()
        self.P = P
        self.pack = torch.zeros((1 + len(self.domains)) * P, dtype=torch.float32, device=theta.device)
        self.delta = self.pack[:P]
        self.phis = {}
        for k, d in enumerate(self.domains):
            v = self.pack[(1 + k) * P:(2 + k) * P]
            v.copy_(phis[d])
            self.phis[d] = v
        self.zero = torch.zeros_like(theta)
        self.merged = torch.empty_like(theta)
        self.mine = None            # queries whose phi this rank updated last (None: every slot is current)
        self.last_queries = set()   # queries ANY rank updated in that epoch (the plan is global)
        self.last_load = None

    def _keep_only_current(self):
        """zero every phi slot another rank holds the current value of; a slot nobody updated is identical
        everywhere and stays on rank 0 only, so that the sum over ranks returns it unchanged."""
        rank, _ = world()
        for d in self.domains:
            if d in self.mine or (d not in self.last_queries and rank == 0):
                continue
            self.phis[d].zero_()

    def sync_phis(self):
        rank, ws = world()
        if ws > 1 and self.mine is not None:
            self._keep_only_current()
            dist.all_reduce(self.pack[self.P:], op=dist.ReduceOp.SUM)
        self.mine = None

    def epoch(self, plan, perm_prepare, perm_fn, batch_size, lr, meta_lr, merged_method="plus",
              domain_regulation_step=0, batch_variant=False, sample_num=None, finetune_every_epoch=False):
        """perm_prepare(passes) (optional) is told this rank's passes in execution order before they run
        (plan.EpochShuffles.prepare).  Returns the trace of (phase, domain, n_steps)."""
        from . import plan as mplan
        rank, ws = world()
        eng, meta, theta = self.eng, self.meta, self.theta
        if ws == 1:
            if perm_prepare is not None:
                perm_prepare(mplan.epoch_passes(plan, domain_regulation_step))
            return meta.mamdr_epoch(eng, theta, self.phis, plan, perm_fn, batch_size, lr, meta_lr, merged_method,
                                    domain_regulation_step, batch_variant, sample_num, scratch=self.merged,
                                    finetune_every_epoch=finetune_every_epoch)
        dr_owner, dn_owner, load = epoch_assignment(plan, self.steps, ws, domain_regulation_step)
        self.last_load = load
        local = {"seq": [d for d in plan["seq"] if dn_owner[d] == rank],
                 "dr": [(q, s) for (q, s) in plan["dr"] if dr_owner[q] == rank]}
        if perm_prepare is not None:
            if finetune_every_epoch:
                raise ValueError("pre-drawn epoch shuffles do not cover the per-query finetune passes")
            perm_prepare(mplan.epoch_passes(local, domain_regulation_step))
        trace = []
        acc = torch.zeros_like(theta) if batch_variant else None
        eng.set_weights(theta)
        for d in local["seq"]:
            meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn")
        eng.sub(self.delta, eng.weights, theta)
        if self.mine is None:                       # every slot is current everywhere: only delta travels
            dist.all_reduce(self.delta, op=dist.ReduceOp.SUM)
        else:
            self._keep_only_current()
            dist.all_reduce(self.pack, op=dist.ReduceOp.SUM)
        eng.interp(theta, self.delta, self.zero, meta_lr)
        for query, support in local["dr"]:
            meta.dr_query(eng, theta, self.phis[query], query, support, perm_fn, batch_size, lr, meta_lr, trace,
                          self.merged, merged_method, domain_regulation_step, batch_variant, sample_num, acc)
            if finetune_every_epoch:
                meta.finetune_query(eng, theta, self.phis[query], query, perm_fn, batch_size, lr, trace, self.merged,
                                    merged_method)
        self.mine = set(q for q, _ in local["dr"])
        self.last_queries = set(q for q, _ in plan["dr"])
        return trace


def gather_domain_scalars(local, n_domain, device):
    """local: dict domain -> (loss, auc) for owned domains; returns full dicts on every rank."""
    rank, ws = world()
    t = torch.zeros(n_domain, 3, dtype=torch.float64, device=device)
    for d, (loss, auc) in local.items():
        t[d, 0], t[d, 1], t[d, 2] = loss, auc, 1.0
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu()
    return ({d: float(t[d, 0]) for d in range(n_domain) if t[d, 2] > 0},
            {d: float(t[d, 1]) for d in range(n_domain) if t[d, 2] > 0})
