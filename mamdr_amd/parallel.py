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


def dn_phase_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, trace, delta_buf, zero_buf):
    """DN phase of one epoch on this rank's sub-sequence + the all-reduce outer update.
    theta += (sum_g delta_g) * beta, evaluated as interp(theta, delta, 0, beta)."""
    eng.set_weights(theta)
    for d in seq_local:
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn")
    rank, ws = world()
    if ws == 1:
        eng.interp(theta, eng.weights, theta, meta_lr)
    else:
        allreduce_delta(eng, theta, delta_buf)
        eng.interp(theta, delta_buf, zero_buf, meta_lr)


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
