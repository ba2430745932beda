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

LANES (round 5): the same sharding inside ONE process.  One dependent chain of 1,024-row steps cannot fill 256 CUs
(each step is two launches whose workgroups wait for each other's results), but the units that shard
over ranks are just as independent on one GPU: `LaneGroup(L).run(fn)` runs `fn(lane)` on L host threads, each lane with
its own engine on its own HIP stream, and inside those threads `world()` answers (lane, L) and the collectives of this
module (`all_reduce`, `broadcast`, the phi hand-over) become stream-ordered copies / sums between the lanes' buffers --
every function below runs unchanged, and a lane run IS the L-rank run (same assignment, same arithmetic; sums in lane order,
which is the order of a 2-rank gloo / RCCL all-reduce -- tested bit for bit for 2 lanes; rings of more ranks add chunk by
chunk in orders of their own), with the kernels of different lanes overlapping on the device.

RANKS x LANES (round 6): the two compose.  Under N processes each LaneGroup of L lanes is a slice of ONE world of N * L
participants -- `world()` answers (rank * L + lane, N * L) -- so a rank's share of an epoch (its DR queries, its DN
sub-sequence) is dealt on to its lanes by the same assignment code, and a collective is the lane step on the device
followed by ONE inter-rank collective per process: all_reduce = sum over the process's lanes in lane order, lane 0's
`dist.all_reduce` of that sum, lane 0's result copied to the other lanes; broadcast = the source lane's thread broadcasts
between the processes, then between the lanes; a phi slot that changes hands travels lane to lane inside a process and
through lane 0's batched send / recv between processes.  Exactly one thread of a process is inside torch.distributed at
any time (the lanes meet at a host barrier around every collective) and every process issues the same sequence.
"""
import threading

import torch
import torch.distributed as dist

_lane = threading.local()


class LaneGroup(object):
    """L lanes = L host threads of one process standing in for L ranks (module docstring).  Collectives between lanes:
    every lane publishes its tensor and an event recorded on ITS stream, a host barrier, every lane makes its stream wait
    for the publishers' events and reads; a second round of events keeps a publisher from overwriting what a reader has
    not read yet.  No host synchronisation with the device anywhere: the lanes' streams stay asynchronous."""

    def __init__(self, n, outer=None, sum_block=None):
        """outer = (rank, world size) of the process group this process is a rank of (default: read from
        torch.distributed; (0, 1) without one).  sum_block (tests): the all-reduce adds the lanes in blocks of that many --
        the order in which `sum_block` lanes x n / sum_block processes add up -- instead of one run in lane order."""
        self.n = int(n)
        if outer is None:
            outer = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
        self.outer_rank, self.outer_ws = int(outer[0]), int(outer[1])
        self.sum_block = int(sum_block) if sum_block else self.n
        if self.n % self.sum_block:
            raise ValueError("sum_block %d does not divide %d lanes" % (self.sum_block, self.n))
        self.barrier = threading.Barrier(self.n)
        self.slots = [None] * self.n
        self.ev = [None] * self.n
        self.ev2 = [None] * self.n
        self.adders = [None] * self.n

    def run(self, fn):
        """fn(lane) on every lane (lane 0 on the calling thread's device); returns the list of results.  The first
        exception of any lane is re-raised after every lane has ended (a failing lane breaks the barrier: no lane waits
        for it for ever)."""
        results, errors = [None] * self.n, [None] * self.n
        device = torch.cuda.current_device() if torch.cuda.is_available() else None
        self.barrier = threading.Barrier(self.n)        # (a barrier broken by a failed run stays broken: ADVICE r05)

        def body(lane):
            _lane.group, _lane.rank = self, lane
            try:
                if device is not None:
                    torch.cuda.set_device(device)
                    with torch.cuda.stream(_lane_stream(device, lane)):
                        results[lane] = fn(lane)
                        torch.cuda.current_stream().synchronize()
                else:
                    results[lane] = fn(lane)
            except BaseException as e:      # noqa: B902 -- reported below, on the caller's thread
                errors[lane] = e
                self.barrier.abort()
            finally:
                _lane.group, _lane.rank = None, 0
        threads = [threading.Thread(target=body, args=(l,), name="mamdr-lane-%d" % l) for l in range(self.n)]
        _lane_stdout(+1)                    # every lane prints the same progress lines: lane 0's are the run's
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            _lane_stdout(-1)
        first = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)] or \
                [e for e in errors if e is not None]
        if first:
            raise first[0]
        return results

    # -- the collectives (called through the module functions below, from lane threads only)
    def wait(self):
        self.barrier.wait()

    def _publish(self, rank, obj, cuda):
        self.slots[rank] = obj
        if cuda:
            e = torch.cuda.Event()
            e.record()
            self.ev[rank] = e
        self.wait()

    def _read_done(self, rank, cuda, publishers):
        """readers have ENQUEUED their reads: publishers make their streams wait for them before going on."""
        if cuda:
            e = torch.cuda.Event()
            e.record()
            self.ev2[rank] = e
        self.wait()
        if cuda and rank in publishers:
            s = torch.cuda.current_stream()
            for k in range(self.n):
                if k != rank:
                    s.wait_event(self.ev2[k])
        self.wait()             # (the slots and events may be reused from here on)

    def all_reduce(self, rank, t, op):
        cuda = t.is_cuda
        self._publish(rank, t, cuda)
        if cuda:
            s = torch.cuda.current_stream()
            for k in range(self.n):
                if k != rank:
                    s.wait_event(self.ev[k])
        add = self.adders[rank] if (op == "sum" and t.dtype == torch.float32 and t.dim() == 1) else None

        def fold(acc, other):
            if add is not None:
                add(acc, other)
            elif op == "sum":
                torch.add(acc, other, out=acc)
            elif op == "max":
                torch.maximum(acc, other, out=acc)
            else:
                torch.minimum(acc, other, out=acc)
        # lane order on every lane: the same bits everywhere.  (sum_block < n: block sums first, then the blocks in order --
        # what n / sum_block processes of sum_block lanes each compute; one block = one run in lane order)
        acc = None
        for b0 in range(0, self.n, self.sum_block):
            blk = self.slots[b0].clone()
            for k in range(b0 + 1, b0 + self.sum_block):
                fold(blk, self.slots[k])
            if acc is None:
                acc = blk
            else:
                fold(acc, blk)
        self._read_done(rank, cuda, range(self.n))
        if self.outer_ws > 1:
            # ONE inter-rank collective per process: lane 0 reduces the lanes' sum over the ranks, the others take its result
            if rank == 0:
                dist.all_reduce(acc, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
            self._publish(rank, acc, cuda)
            if rank != 0:
                if cuda:
                    torch.cuda.current_stream().wait_event(self.ev[0])
                acc.copy_(self.slots[0])
            self._read_done(rank, cuda, (0,))
        t.copy_(acc)

    def broadcast(self, rank, t, src):
        """src: a participant of the whole world (process src // n, lane src % n)."""
        cuda = t.is_cuda
        src_proc, src = divmod(int(src), self.n)
        if self.outer_ws > 1:
            if rank == src:             # the source lane's thread moves it between the processes (one thread inside dist)
                dist.broadcast(t, src=src_proc)
        elif src_proc != 0:
            raise ValueError("broadcast from participant %d of a world of %d" % (src_proc * self.n + src, self.n))
        self._publish(rank, t, cuda)
        if rank != src:
            if cuda:
                torch.cuda.current_stream().wait_event(self.ev[src])
            t.copy_(self.slots[src])
        self._read_done(rank, cuda, (src,))

    def transfer(self, rank, vectors, moves):
        """moves [(key, src, dst)] between participants of the whole world (process p = x // n, lane x % n): vectors[key] of
        src -> vectors[key] of dst.  Inside this process a stream-ordered device copy; between processes lane 0 sends /
        receives all of this process's slots in ONE batch (the tensors are the lanes' own: lane 0's stream waits for the
        source lanes' events first, the destination lanes wait for lane 0's before they go on).  -> bytes this lane's
        slots put on the wire."""
        cuda = any(v.is_cuda for v in vectors.values())
        me = self.outer_rank
        self._publish(rank, vectors, cuda)
        touched, sent = set(), 0
        remote = []
        for key, src, dst in moves:
            (sp, sl), (dp, dl) = divmod(int(src), self.n), divmod(int(dst), self.n)
            if sp == me and dp == me:
                touched.add(sl)
                if rank == dl and sl != dl:
                    if cuda:
                        torch.cuda.current_stream().wait_event(self.ev[sl])
                    vectors[key].copy_(self.slots[sl][key])
            elif sp == me or dp == me:
                remote.append((key, sp, sl, dp, dl))
                touched.add(sl if sp == me else dl)
                if sp == me and rank == sl:
                    sent += vectors[key].numel() * vectors[key].element_size()
        if remote and self.outer_ws > 1:
            touched.add(0)
            if rank == 0:
                ops = []
                for key, sp, sl, dp, dl in remote:
                    if sp == me:
                        if cuda and sl != 0:
                            torch.cuda.current_stream().wait_event(self.ev[sl])
                        ops.append(dist.P2POp(dist.isend, self.slots[sl][key], dp))
                    else:
                        if cuda and dl != 0:
                            torch.cuda.current_stream().wait_event(self.ev[dl])
                        ops.append(dist.P2POp(dist.irecv, self.slots[dl][key], sp))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        self._read_done(rank, cuda, touched)
        return sent


_lane_streams = {}


def _lane_stream(device, lane):
    """lane k of every LaneGroup of this process runs on the SAME stream: streams map to hardware queues in creation order, and
    a second group on fresh streams can land two lanes on one queue (Taobao-30, 4 lanes: 28 K instead of 33 K domain-steps/s
    for the second group of a process)."""
    key = (device, lane)
    if key not in _lane_streams:
        _lane_streams[key] = torch.cuda.Stream(device=device)
    return _lane_streams[key]


_stdout_lock = threading.Lock()
_stdout_state = {"count": 0, "saved": None}


def _lane_stdout(delta):
    """the filter that drops the prints of lanes > 0 is installed while ANY LaneGroup runs (counted: two groups running side by
    side, or nested, install it once and the last one out restores the stream -- ADVICE r05); threads that are no lanes
    print as before."""
    import sys
    with _stdout_lock:
        st = _stdout_state
        if delta > 0:
            if st["count"] == 0:
                st["saved"] = sys.stdout
                sys.stdout = _Lane0Stdout(sys.stdout)
            st["count"] += 1
        else:
            st["count"] -= 1
            if st["count"] == 0:
                if isinstance(sys.stdout, _Lane0Stdout):
                    sys.stdout = st["saved"]
                st["saved"] = None


class _Lane0Stdout(object):
    def __init__(self, out):
        self._out = out

    def write(self, text):
        if getattr(_lane, "rank", 0) == 0:
            return self._out.write(text)
        return len(text)

    def __getattr__(self, name):
        return getattr(self._out, name)


def lanes():
    """the LaneGroup this thread is a lane of, or None."""
    return getattr(_lane, "group", None)


def lane_adder(eng):
    """the lanes' all-reduce of fp32 vectors sums with this engine's elementwise kernel (mamdr_merge, mode plus) on the
    lane's stream instead of a torch op.  No-op outside a lane."""
    g = lanes()
    if g is not None and hasattr(eng, "merge"):
        g.adders[_lane.rank] = lambda dst, src: eng.merge(dst, dst, src, "plus")


def all_reduce(t, op="sum"):
    g = lanes()
    if g is not None:
        g.all_reduce(_lane.rank, t, op)
    else:
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])


def broadcast(t, src):
    g = lanes()
    if g is not None:
        g.broadcast(_lane.rank, t, src)
    else:
        dist.broadcast(t, src=src)


def barrier():
    g = lanes()
    if g is not None:
        g.wait()
        if g.outer_ws > 1:
            if _lane.rank == 0:
                dist.barrier()
            g.wait()
    elif dist.is_available() and dist.is_initialized():
        dist.barrier()


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
    """(this participant, participants): a lane of a LaneGroup under N processes is participant rank * L + lane of N * L."""
    g = lanes()
    if g is not None:
        return g.outer_rank * g.n + _lane.rank, g.outer_ws * g.n
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# phi slots travel owner -> next owner point to point (BalancedMAMDR._transfer); False: one broadcast per slot instead
# (what `preflight` switches to when the ring of sends / receives fails on this machine)
P2P_ENABLED = True


def p2p_possible(device):
    """send / recv of a tensor on `device` exists on the initialised backend (gloo cannot send device tensors here)."""
    return dist.get_backend() == "nccl" or torch.device(device).type == "cpu"


def preflight(device, payload=4096):
    """First contact with the communicator, before any epoch runs (bench.py / run.py under torch.distributed.run):
    one tiny all-reduce, one ring of batch_isend_irecv (rank r -> r + 1: the pattern of the phi hand-over), one
    broadcast -- each checked for its VALUES on every rank.  A hang is cut by the process group's timeout (bench.py and
    cli.init_distributed both create the group with an explicit one, 300 s by default; the rank exits non-zero).  A send /
    recv that RAISES PROMPTLY ON EVERY RANK (a backend without the operation, a symmetric failure) switches every rank to
    the per-slot broadcast path (P2P_ENABLED = False, agreed by an all-reduce) with a warning; a failure on ONE rank leaves
    its peers waiting in `req.wait()`, which the timeout ends -- that case is a failed run, not a fallback.  Returns a
    record for the bench line."""
    import time
    import warnings
    global P2P_ENABLED
    rank, ws = world()
    rec = {"ranks": ws, "backend": ("lanes" if lanes() is not None else dist.get_backend()) if ws > 1 else None, "all_reduce": None, "p2p": None, "broadcast": None}
    if ws == 1 or lanes() is not None:
        return rec
    t0 = time.perf_counter()
    dev = torch.device(device)
    x = torch.full((payload,), float(rank + 1), dtype=torch.float32, device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    want = ws * (ws + 1) / 2.0
    if not bool((x == want).all().item()):
        raise RuntimeError("preflight: all-reduce over %d ranks gave %r, not %r" % (ws, float(x[0].item()), want))
    rec["all_reduce"] = True
    ok = 1.0
    if p2p_possible(dev):
        try:
            out = torch.arange(payload, dtype=torch.float32, device=dev) + 1000.0 * rank
            got = torch.empty_like(out)
            ops = [dist.P2POp(dist.isend, out, (rank + 1) % ws), dist.P2POp(dist.irecv, got, (rank - 1) % ws)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            src = (rank - 1) % ws
            if not bool((got == torch.arange(payload, dtype=torch.float32, device=dev) + 1000.0 * src).all().item()):
                raise RuntimeError("wrong payload from rank %d" % src)
        except Exception as e:                       # (a hang is the timeout's business)
            warnings.warn("preflight: point-to-point ring failed on rank %d (%s): phi slots will move by broadcast" % (rank, e))
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        P2P_ENABLED = bool(flag.item() > 0.5)
        rec["p2p"] = P2P_ENABLED
    else:
        rec["p2p"] = "n/a (%s has no send / recv of device tensors)" % dist.get_backend()
    y = torch.full((payload,), 7.0 if rank == ws - 1 else -1.0, dtype=torch.float32, device=dev)
    dist.broadcast(y, src=ws - 1)
    if not bool((y == 7.0).all().item()):
        raise RuntimeError("preflight: broadcast from rank %d did not arrive on rank %d" % (ws - 1, rank))
    rec["broadcast"] = True
    rec["seconds"] = time.perf_counter() - t0
    return rec


def shard_plan(plan, owner, rank):
    """this rank's part of an epoch plan: its DN sub-sequence (order preserved) and the DR
    entries of the query domains it owns."""
    return {"seq": [d for d in plan["seq"] if owner[d] == rank],
            "dr": [(q, s) for (q, s) in plan["dr"] if owner[q] == rank]}


def allreduce_delta(eng, theta, delta_buf):
    """delta = live - theta on every rank; sum over ranks (ONE collective)."""
    eng.sub(delta_buf, eng.meta_weights, theta)
    rank, ws = world()
    if ws > 1:
        all_reduce(delta_buf)
    return delta_buf


def dn_phase_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, trace, delta_buf, zero_buf,
                     meta_train_step=0, target=-1):
    """DN phase of one epoch on this rank's sub-sequence + the all-reduce outer update.
    theta += (sum_g delta_g) * beta, evaluated as interp(theta, delta, 0, beta).
    meta_train_step caps every pass (domain_negotiation.py:67); a target domain (:44-45,89-93) closes EVERY rank's
    sub-sequence with an uncapped pass -- each displacement then ends adapted to the target, as the single
    sequence's does -- and the caller runs the closing target pass on the updated model (identical on every rank)."""
    lane_adder(eng)
    eng.assign_meta(theta)
    for d in seq_local:
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn", meta_train_step)
    if target >= 0:
        meta.run_pass(eng, target, perm_fn, batch_size, lr, trace, "dn")
    rank, ws = world()
    if ws == 1:
        eng.interp(theta, eng.meta_weights, theta, meta_lr)
    else:
        allreduce_delta(eng, theta, delta_buf)
        eng.interp(theta, delta_buf, zero_buf, meta_lr)


def reptile_epoch_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, delta_buf, zero_buf,
                          meta_train_step=0, target=-1):
    """Reptile, per-domain variant (reptile.py:45-99): the reference interpolates theta after EVERY domain, a
    sequential recurrence.  Sharded: every rank runs that recurrence over its own domains on a private copy
    starting from the epoch's theta, then the ranks' total displacements are summed (ONE all-reduce) and applied:
    theta += sum_g (theta_g - theta).  One rank: exactly the reference's epoch.
    target >= 0 (reptile.py:47-48,82-85,98-102): the target domain is no task; every domain's pass is followed by ONE
    step on the target domain before its interpolation (on whichever rank runs that domain), and the caller closes the
    epoch with a full pass of the updated model over the target domain (on every rank: the model is the same)."""
    lane_adder(eng)
    trace = []
    local = theta.clone()
    for d in seq_local:
        if target >= 0 and d == target:
            continue
        eng.assign_meta(local)
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "reptile", meta_train_step)
        if target >= 0:
            meta.run_pass(eng, target, perm_fn, batch_size, lr, trace, "target_step", 1)
        eng.interp(local, eng.meta_weights, local, meta_lr)
    rank, ws = world()
    if ws == 1:
        theta.copy_(local)
    else:
        eng.sub(delta_buf, local, theta)
        all_reduce(delta_buf)
        eng.interp(theta, delta_buf, zero_buf, 1.0)
    eng.assign_meta(theta)
    return trace


def reptile_batch_epoch_sharded(eng, meta, theta, seq_local, perm_fn, batch_size, lr, meta_lr, acc, meta_train_step=0,
                                target=-1):
    """Reptile, batch variant (reptile.py:87-96,134-142): every domain starts from theta and adds its
    displacement theta~ - theta to `acc`; the epoch applies theta += beta * sum.  The sum over domains is a sum
    over ranks of per-rank sums: ONE all-reduce of `acc` per epoch, no other change to the algorithm.
    Only the optimiser slots differ from the single-process run: each rank's Adam moments see its own domains."""
    lane_adder(eng)
    trace = []
    acc.zero_()
    for d in seq_local:
        if target >= 0 and d == target:
            continue
        eng.assign_meta(theta)
        meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "reptile", meta_train_step)
        if target >= 0:           # reptile.py:82-85: one step on the target domain after every domain's pass
            meta.run_pass(eng, target, perm_fn, batch_size, lr, trace, "target_step", 1)
        eng.accumulate(acc, eng.meta_weights, theta)
    rank, ws = world()
    if ws > 1:
        all_reduce(acc)
    eng.apply_accumulated(theta, acc, 0.0, meta_lr)
    eng.assign_meta(theta)
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


class TailSync(object):
    """The tensors OUTSIDE theta / phi (Star tower: PartitionedNorm gamma / beta, the per-domain kernels and biases, the
    output unit -- `weights[n_meta:]` -- and the non-trainable moving statistics `aux`) are trained by the inner steps
    only and live in each rank's engine.  The reference updates them in ONE sequence of passes
    (model_zoo/Star/star.py:70-127 under mamdr.py:41-108); sharded, every rank applies its own passes to its own
    copy.  `sync()` makes them one model again: since the last common value every rank moved its copy by delta_g in
    k_g steps of its own, the new common value is common + the STEP-WEIGHTED MEAN of the displacements -- a per-domain
    slice (specific kernels / biases, PartitionedNorm's per-domain gamma / beta) weighted by the steps each rank took ON
    THAT DOMAIN (a slice only one rank trained keeps that rank's value), shared tensors (shared gamma / beta, the output
    unit) by each rank's total steps.  Not the sum the DN displacement of theta takes (that one is damped by the meta
    learning rate): these tensors are stepped by Adam with no outer rate, every rank walks the whole way to where its
    gradient vanishes, and N such walks added up overshoot N-fold -- measured with the Star tower on Taobao-10
    (tools/dist_auc_star.sh, profiles/r03u_dist_auc_star.jsonl): summed displacements 0.8116 (N = 1) -> 0.8079 (2) ->
    0.7703 (4) average test AUC.  MAMDR_TAIL_SYNC=sum keeps the sum for comparison.  The moving statistics of domain d are exponential averages of batch
    statistics, not sums: ranks are combined weighted by the number of steps k_g each took on d since the last sync,
    mov_d = sum_g k_g mov_d,g / sum_g k_g, and the zero-debias slots are rebuilt for the summed step count
    (partitioned_norm.py:177-193: biased = mov * (1 - 0.99^steps)).  Adam's moments stay per rank, like every
    other optimiser slot.  No-op for towers whose flat vector is all meta (mlp / deepfm / wdl)."""

    def __init__(self, eng):
        self.eng = eng
        if getattr(eng, "meta_holes", ()) and world()[1] > 1:
            raise NotImplementedError("multi-process runs do not support scattered meta_parms lists (tensors that are no "
                                      "neighbours in the flat vector)")
        if getattr(eng, "meta_off", 0) and eng.n_meta != eng.n_params and world()[1] > 1:
            raise NotImplementedError("multi-process runs support meta parameters that form a prefix of the flat vector "
                                      "(\"all\", the Star filter); got the range [%d, %d)" % (eng.meta_off, eng.meta_off + eng.n_meta))
        self.n_meta, self.n_tail = eng.n_meta, eng.n_params - eng.n_meta
        if getattr(eng, "meta_off", 0):
            self.n_tail = 0
        self.aux = getattr(eng, "aux", None)
        self.active = self.n_tail > 0
        if not self.active:
            return
        self.common = eng.weights[self.n_meta:].clone()
        self.aux_common = self.aux.clone() if self.aux is not None else None
        import os
        self.mean = os.environ.get("MAMDR_TAIL_SYNC", "mean") != "sum"
        # domain of every tail element (-1: shared): the per-domain tensors are [D][...] blocks of the flat vector
        self.elem_dom = torch.full((self.n_tail,), -1, dtype=torch.int64, device=self.common.device)
        D = eng.n_domain
        for name, (off, cnt) in getattr(eng, "segments", {}).items():
            if off >= self.n_meta and cnt % D == 0 and (name[:2] in ("Wd", "bd") or name in ("pn_gamma_spec", "pn_beta_spec")):
                per = cnt // D
                self.elem_dom[off - self.n_meta:off - self.n_meta + cnt] = torch.arange(cnt, device=self.common.device) // per
        self.is_dom = self.elem_dom >= 0
        self.dom_idx = self.elem_dom.clamp(min=0)

    EPS = 1e-3      # weight of a rank's total steps inside a per-domain slice: decides only where NO rank trained the domain

    def _weights(self, k):
        """per-element weights from the per-domain step counts k [D] (one rank's own, or the sum over the ranks)."""
        tot = k.sum()
        return torch.where(self.is_dom, k[self.dom_idx] + self.EPS * tot, tot.expand(self.n_tail))

    def floats(self):
        """payload of one sync (floats all-reduced)."""
        if not self.active:
            return 0
        D = self.eng.n_domain
        return self.n_tail + ((self.aux.numel() - D) // 2 + D if self.aux is not None else 0)

    def _aux_views(self, a):
        D = self.eng.n_domain
        X = (a.numel() - D) // (4 * D)
        dx = D * X
        return a[0:dx].view(D, X), a[dx:2 * dx].view(D, X), a[2 * dx:3 * dx].view(D, X), a[3 * dx:4 * dx].view(D, X), \
            a[4 * dx:4 * dx + D]

    def rebase(self):
        """the live tail IS the common value (every rank holds the same one: after a broadcast / at start)."""
        if self.active:
            self.common.copy_(self.eng.weights[self.n_meta:])
            if self.aux is not None:
                self.aux_common.copy_(self.aux)

    def fill(self, buf):
        """write this rank's payload (floats() elements) into `buf`: [tail - common | k * mov_mean | k * mov_var | k]."""
        live = self.eng.weights[self.n_meta:]
        torch.sub(live, self.common, out=buf[:self.n_tail])
        if self.aux is not None:
            mm, mv, _, _, steps = self._aux_views(self.aux)
            D, X = mm.shape
            k = (steps - self._aux_views(self.aux_common)[4]).clamp_(min=0.0)
            if self.mean:
                buf[:self.n_tail].mul_(self._weights(k))
            o = self.n_tail
            buf[o:o + D * X].view(D, X).copy_(mm * k[:, None])
            buf[o + D * X:o + 2 * D * X].view(D, X).copy_(mv * k[:, None])
            buf[o + 2 * D * X:o + 2 * D * X + D].copy_(k)

    def apply(self, buf):
        """`buf` summed over the ranks -> the new common value, on every rank."""
        live = self.eng.weights[self.n_meta:]
        if self.aux is not None and self.mean:
            D = self.eng.n_domain
            X = (self.aux.numel() - D) // (4 * D)
            W = self._weights(buf[self.n_tail + 2 * D * X:self.n_tail + 2 * D * X + D])
            live.copy_(self.common + torch.where(W > 0, buf[:self.n_tail] / W.clamp(min=1e-30), torch.zeros_like(W)))
        else:
            live.copy_(self.common + buf[:self.n_tail])
        if self.aux is not None:
            mm, mv, bm, bv, steps = self._aux_views(self.aux)
            D, X = mm.shape
            o = self.n_tail
            smm, smv = buf[o:o + D * X].view(D, X), buf[o + D * X:o + 2 * D * X].view(D, X)
            sk = buf[o + 2 * D * X:o + 2 * D * X + D]
            cm, cv, _, _, csteps = self._aux_views(self.aux_common)
            moved = sk > 0
            w = torch.where(moved, sk, torch.ones_like(sk))[:, None]
            mm.copy_(torch.where(moved[:, None], smm / w, cm))
            mv.copy_(torch.where(moved[:, None], smv / w, cv))
            steps.copy_(csteps + sk)
            factor = (1.0 - torch.pow(torch.full_like(steps, 0.99), steps))[:, None]
            bm.copy_(mm * factor)
            bv.copy_(mv * factor)
        self.rebase()

    def sync(self):
        """a collective of its own (before validation / checkpoints; inside an epoch the payload rides in the DN
        all-reduce: BalancedMAMDR.epoch)."""
        rank, ws = world()
        if not self.active or ws == 1:
            return
        buf = torch.empty(self.floats(), dtype=torch.float32, device=self.common.device)
        self.fill(buf)
        all_reduce(buf)
        self.apply(buf)


class BalancedMAMDR(object):
    """DN + DR epochs sharded over the ranks with a per-epoch assignment and ONE data-path collective per epoch.

    Every rank keeps theta and a slot for every phi_d; `where[d]` is the rank whose slot holds the CURRENT phi_d
    (None: identical everywhere).  An epoch:
      1. the plan (same seed everywhere) -> epoch_assignment -> this rank's DN sub-sequence and DR queries;
      2. phi hand-over: a slot whose next owner differs from `where` travels owner -> next owner point to point
         (RCCL send / recv over the xGMI link of that pair; every rank computes the same move list) -- with trainable
         tables a phi is 0.3 GB, and only the slots that change hands move (round 2 all-reduced all D of them);
      3. DN passes from theta; ONE all-reduce (sum) of [theta~ - theta | tail displacement] (TailSync);
      4. theta += beta * sum_g delta_g;  5. DR of the owned queries (meta.dr_query), phi slots updated in place.
    dn_mode "replicated" (SURVEY 8e's fallback): every rank runs the WHOLE DN sequence -- one sequential chain,
    the reference's update (domain_negotiation.py:53-88) -- and rank 0's theta~ (+ tail) is broadcast so that all
    ranks continue from the same bits (their Adam step counts differ); DR stays sharded.
    With one rank no collective runs and the epoch is meta.mamdr_epoch's (the reference's loop).
    `sync_phis()` makes every slot current everywhere; `owner(d)` says who evaluates / finetunes domain d."""

    def __init__(self, eng, meta, theta, phis, steps_per_domain, dn_mode="sharded"):
        """phis: {domain: vector}, the SAME initial values on every rank (every rank draws all D initialisations)."""
        if dn_mode not in ("sharded", "replicated"):
            raise ValueError("dn_mode must be 'sharded' or 'replicated', not: {}".format(dn_mode))
        self.eng, self.meta, self.theta = eng, meta, theta
        lane_adder(eng)
        self.dn_mode = dn_mode
        self.steps = list(steps_per_domain)
        self.domains = sorted(phis)
        P = theta.numel()
        self.P = P
        self.tail = TailSync(eng)
        T = self.tail.floats()
        T = (T + 3) // 4 * 4
        # [delta | tail payload | phi_0 ... phi_{D-1}]: the DN collective reduces the first P + T floats in place
        self.pack = torch.zeros((1 + len(self.domains)) * P + T, dtype=torch.float32, device=theta.device)
        self.delta = self.pack[:P]
        self.tail_buf = self.pack[P:P + T]
        self.phis = {}
        for k, d in enumerate(self.domains):
            v = self.pack[T + (1 + k) * P:T + (2 + k) * P]
            v.copy_(phis[d])
            self.phis[d] = v
        self.zero = torch.zeros_like(theta)
        self.merged = torch.empty_like(theta)
        self.where = {d: None for d in self.domains}
        self.mine = None            # queries this rank ran in the last epoch
        self.last_load = None
        self.wire_bytes = []        # per epoch: payload bytes this rank put into collectives + point-to-point sends
        self.host_prep_s = 0.0      # host work that no rank count divides: assignment + drawing / uploading the shuffles

    def owner(self, d, fallback=0):
        """rank holding the current phi_d (`fallback` when every rank does)."""
        w = self.where.get(d)
        return fallback if w is None else w

    def _transfer(self, moves):
        """moves: [(domain, src rank, dst rank)] in the same order on every rank."""
        rank, ws = world()
        sent = 0
        if not moves:
            return sent
        g = lanes()
        if g is not None and (g.outer_ws == 1 or (P2P_ENABLED and p2p_possible(self.pack.device))):
            # lane to lane inside the process: a stream-ordered device copy; between processes: lane 0's batched send / recv
            g.transfer(_lane.rank, self.phis, moves)
            return sum(self.P * 4 for _, src, _ in moves if src == rank)
        p2p = g is None and P2P_ENABLED and p2p_possible(self.pack.device)
        if p2p:
            ops = []
            for d, src, dst in moves:
                if rank == src:
                    ops.append(dist.P2POp(dist.isend, self.phis[d], dst))
                    sent += self.P * 4
                elif rank == dst:
                    ops.append(dist.P2POp(dist.irecv, self.phis[d], src))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        else:                       # gloo with device tensors (ranks sharing one GPU in tests): no send / recv there
            for d, src, dst in moves:
                broadcast(self.phis[d], src)
                sent += self.P * 4 if rank == src else 0
        return sent

    def sync_phis(self):
        rank, ws = world()
        if ws > 1:
            for d in self.domains:
                if self.where[d] is not None:
                    broadcast(self.phis[d], self.where[d])
        self.where = {d: None for d in self.domains}

    def sync_tail(self):
        """tensors outside theta / phi -> one model on every rank (before validation / checkpoints)."""
        self.tail.sync()

    def _assign(self, plan, domain_regulation_step=0):
        """this epoch's owners (a pure function of the plan, the same on every rank) and this rank's part of it."""
        rank, ws = world()
        if self.dn_mode == "replicated":    # DN is not dealt out: the DR queries alone are balanced
            dr_owner, _, load = epoch_assignment({"seq": [], "dr": plan["dr"]}, self.steps, ws, domain_regulation_step)
            dn_owner = {d: rank for d in plan["seq"]}
            load = [l + sum(self.steps[d] for d in plan["seq"]) for l in load]
        else:
            dr_owner, dn_owner, load = epoch_assignment(plan, self.steps, ws, domain_regulation_step)
        local = {"seq": [d for d in plan["seq"] if dn_owner[d] == rank],
                 "dr": [(q, s) for (q, s) in plan["dr"] if dr_owner[q] == rank]}
        return dr_owner, dn_owner, load, local

    def local_passes(self, plan, domain_regulation_step=0):
        """the passes `epoch(plan, ...)` will run on THIS rank, in execution order (what plan.EpochShuffles.prefetch wants
        for the next epoch while the current one runs)."""
        from . import plan as mplan
        rank, ws = world()
        if ws == 1:
            return mplan.epoch_passes(plan, domain_regulation_step)
        return mplan.epoch_passes(self._assign(plan, domain_regulation_step)[3], domain_regulation_step)

    def epoch(self, plan, perm_prepare, perm_fn, batch_size, lr, meta_lr, merged_method="plus",
              domain_regulation_step=0, batch_variant=False, sample_num=None, finetune_every_epoch=False):
        """perm_prepare(passes) (optional) is told this rank's passes in execution order before they run
        (plan.EpochShuffles.prepare).  Returns the trace of (phase, domain, n_steps)."""
        from . import plan as mplan
        rank, ws = world()
        eng, meta, theta = self.eng, self.meta, self.theta
        if perm_prepare is not None and finetune_every_epoch:
            raise ValueError("pre-drawn epoch shuffles do not cover the per-query finetune passes")
        import time
        t_host = time.perf_counter()
        if ws == 1:
            if perm_prepare is not None:
                perm_prepare(mplan.epoch_passes(plan, domain_regulation_step))
            self.host_prep_s += time.perf_counter() - t_host
            return meta.mamdr_epoch(eng, theta, self.phis, plan, perm_fn, batch_size, lr, meta_lr, merged_method,
                                    domain_regulation_step, batch_variant, sample_num, scratch=self.merged,
                                    finetune_every_epoch=finetune_every_epoch)
        replicated = self.dn_mode == "replicated"
        dr_owner, dn_owner, load, local = self._assign(plan, domain_regulation_step)
        self.last_load = load
        if perm_prepare is not None:
            perm_prepare(mplan.epoch_passes(local, domain_regulation_step))
        self.host_prep_s += time.perf_counter() - t_host
        wire = self._transfer([(q, self.where[q], dr_owner[q]) for q, _ in sorted(plan["dr"])
                               if self.where[q] is not None and self.where[q] != dr_owner[q]])
        trace = []
        acc = torch.zeros_like(theta) if batch_variant else None
        if replicated and self.tail.active:
            self.tail.sync()            # the previous epoch's DR displacements of the tail, before DN moves it again
            wire += self.tail.floats() * 4
        eng.assign_meta(theta)
        pw = meta.PassWindow(eng, perm_fn, batch_size)
        pw.announce(local["seq"])
        for d in local["seq"]:
            pw.step()
            meta.run_pass(eng, d, perm_fn, batch_size, lr, trace, "dn")
        if replicated:
            # one chain; rank 0's result is everybody's (bit-identical continuation on every rank)
            live = eng.weights
            if self.tail.active:
                broadcast(live, 0)
                if self.tail.aux is not None:
                    broadcast(self.tail.aux, 0)
                self.tail.rebase()
                wire += live.numel() * 4 if rank == 0 else 0
                eng.sub(self.delta, live[:self.P], theta)
            else:
                eng.sub(self.delta, live, theta)
                broadcast(self.delta, 0)
                wire += self.P * 4 if rank == 0 else 0
        else:
            eng.sub(self.delta, eng.meta_weights, theta)
            if self.tail.active:
                # [delta | tail displacement | statistics] in ONE collective
                self.tail.fill(self.tail_buf)
                all_reduce(self.pack[:self.P + self.tail_buf.numel()])
                self.tail.apply(self.tail_buf)
                wire += (self.P + self.tail_buf.numel()) * 4
            else:
                all_reduce(self.delta)
                wire += self.P * 4
        eng.interp(theta, self.delta, self.zero, meta_lr)
        for query, support in local["dr"]:
            meta.dr_query(eng, theta, self.phis[query], query, support, perm_fn, batch_size, lr, meta_lr, trace,
                          self.merged, merged_method, domain_regulation_step, batch_variant, sample_num, acc)
            if finetune_every_epoch:
                meta.finetune_query(eng, theta, self.phis[query], query, perm_fn, batch_size, lr, trace, self.merged,
                                    merged_method)
        self.mine = set(q for q, _ in local["dr"])
        for q, _ in plan["dr"]:
            self.where[q] = dr_owner[q]
        self.wire_bytes.append(wire)
        return trace


def broadcast_live(eng, src=0):
    """every rank continues from rank `src`'s LIVE model (weights + Star's moving statistics).  Needed wherever the ranks
    trained their live models separately after the last collective and the loop goes on as if there were one model --
    the closing pass over the target domain of Reptile / Domain Negotiation (reptile.py:98-102,
    domain_negotiation.py:89-93): the ranks' shuffle streams and Adam slots differ by then (each drew a shuffle and
    stepped its optimiser per OWNED domain), so the same pass from the same theta ends in different weights per rank,
    while validation deals the domains round-robin on the premise of one model and rank 0 alone writes the checkpoint
    (ADVICE r04).  The Adam slots stay per rank, as every optimiser slot does."""
    rank, ws = world()
    if ws <= 1:
        return
    w = eng.weights
    broadcast(w, src)
    if w.data_ptr() != eng.weights.data_ptr():      # (an engine that hands out copies of its weights: the tests' CPU stand-in)
        eng.set_weights(w)
    aux = getattr(eng, "aux", None)
    if aux is not None:
        broadcast(aux, src)


def gather_domain_scalars(local, n_domain, device):
    """local: dict domain -> (loss, auc) for owned domains; returns full dicts on every rank."""
    rank, ws = world()
    t = torch.zeros(n_domain, 3, dtype=torch.float64, device=device)
    for d, (loss, auc) in local.items():
        t[d, 0], t[d, 1], t[d, 2] = loss, auc, 1.0
    if ws > 1:
        all_reduce(t)
    t = t.cpu()
    return ({d: float(t[d, 0]) for d in range(n_domain) if t[d, 2] > 0},
            {d: float(t[d, 1]) for d in range(n_domain) if t[d, 2] > 0})
