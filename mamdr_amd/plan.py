"""Host-side randomness of the meta loops, made explicit and seedable.

The reference draws the domain order and the DR support domains from python's
unseeded `random` module (model_zoo/mamdr.py:45-46,65-70,
domain_negotiation.py:41-42, reptile.py:45) and reshuffles every pass through
tf.data (utils/dataset.py:27-37).  Here the same decisions come from a seeded
`random.Random`, so runs are reproducible and the oracle can replay them.
"""
import random

import numpy as np

from . import engine as _engine


class EpochPlanner(object):
    """per-epoch domain order + DR support samples (mamdr.py:44-70)."""

    def __init__(self, domains, sample_num=5, add_query_domain=True, shuffle_sequence=True, seed=123):
        self.seq = list(domains)
        self.sample_num = int(sample_num)
        self.add_query_domain = bool(add_query_domain)
        self.shuffle_sequence = bool(shuffle_sequence)
        self.rng = random.Random(seed)

    def next_sequence(self):
        """random.shuffle(train_sequence) (mamdr.py:45-46): shuffles the persistent list in place."""
        if self.shuffle_sequence:
            self.rng.shuffle(self.seq)
        return list(self.seq)

    def next_epoch(self, with_dr=True):
        seq = self.next_sequence()
        plan = {"seq": seq, "dr": []}
        if with_dr:
            for idx in seq:
                cand = list(seq)
                cand.remove(idx)                                     # mamdr.py:66-67
                aux = self.rng.sample(cand, k=min(self.sample_num, len(cand)))
                if self.add_query_domain:
                    aux.append(idx)                                  # mamdr.py:69-70
                plan["dr"].append((idx, aux))
        return plan


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


class PassShuffler(object):
    """perm_fn(d): a fresh tf.data-style shuffle of domain d's train rows for every pass
    (iterator re-initialisation, mamdr.py:52-53,81-82,92-93).  Seeds are
    mix(base_seed, pass counter); `shuffle_fn` lets the oracle substitute its own
    restatement of the same stream."""

    def __init__(self, sizes, buffer_size=10000, seed=123, shuffle=True, shuffle_fn=None):
        self.sizes = sizes
        self.buffer_size = int(buffer_size)
        self.seed = int(seed)
        self.counter = 0
        self.shuffle = shuffle
        self.shuffle_fn = shuffle_fn or _engine.shuffle_perm

    def __call__(self, d, window=None):
        """window = (begin, end): a pass over that file-order slice only (dataset.take / dataset.skip of the
        meta-train / meta-val split, maml.py:300-330): the slice is shuffled on its own."""
        begin, end = (window[0], window[1]) if window is not None else (0, self.sizes[d])
        if not self.shuffle:
            return None if window is None else np.arange(begin, end, dtype=np.int32)
        self.counter += 1
        seed = _mix64(self.seed * 0x10001 + self.counter)
        if window is not None and len(window) > 2 and window[2] == "stream":
            # (begin, end, "stream"): positions [begin, end) of a fresh shuffle of the WHOLE split -- dataset.shuffle
            # then take / skip, the non-exclusive meta split (maml.py:316-323): two passes may share rows
            return np.ascontiguousarray(self.shuffle_fn(self.sizes[d], self.buffer_size, seed)[begin:end])
        perm = self.shuffle_fn(end - begin, self.buffer_size, seed)
        return perm if begin == 0 else (perm + np.int32(begin)).astype(np.int32)


def epoch_passes(plan, domain_regulation_step=0):
    """the passes of one MAMDR epoch in execution order (mamdr.py:48-108): DN over the sequence, then for every
    query domain and every support domain a support pass and a query pass.  -> [(domain, max_steps)],
    max_steps 0 = the whole split."""
    out = [(d, 0) for d in plan["seq"]]
    for q, support in plan["dr"]:
        for j in support:
            out.append((j, 0))
            out.append((q, int(domain_regulation_step) if domain_regulation_step and domain_regulation_step > 0 else 0))
    return out


class EpochShuffles(object):
    """perm_fn for the meta loops that draws EVERY permutation of an epoch up front -- the same PassShuffler
    stream in the same order, so the passes see the same shuffles as with one upload per pass -- into one pinned
    staging buffer (one C call: mamdr_shuffle_perms) and uploads them in ONE copy; the passes then take device
    slices.  Two staging / device buffers alternate, and `prefetch(passes)` draws the NEXT epoch's permutations on a
    worker thread while the caller enqueues / the stream runs the current one (the C call releases the GIL): on
    Amazon-13 drawing an epoch's 39 K shuffles takes 0.3 s, ten times what the launch queue can hide."""

    def __init__(self, shuffler, device):
        import torch
        self.sh = shuffler
        self.device = device
        self.torch = torch
        self.host = [None, None]
        self.dev = [None, None]
        self.done = [None, None]
        self.k = 0
        self.queue = []
        self.pending = None         # (passes, job, thread) of a prefetch in flight
        self.lookahead = None       # callable: invoked once per epoch, a few passes in (see __call__)
        self.timing = {"stage": 0.0, "stage_sync": 0.0, "draw": 0.0, "join": 0.0, "upload": 0.0, "queue": 0.0}   # seconds, summed

    def _stage(self, passes):
        """seeds (consumed from the shuffler's stream, in pass order) and the staging buffer of one epoch."""
        import time
        t0 = time.perf_counter()
        torch, sh = self.torch, self.sh
        n = np.array([sh.sizes[d] for d, _ in passes], np.int64)
        seeds = np.empty(len(passes), np.uint64)
        for i in range(len(passes)):
            sh.counter += 1
            seeds[i] = _mix64(sh.seed * 0x10001 + sh.counter)
        total = int(n.sum())
        k = self.k = self.k ^ 1
        if self.host[k] is None or self.host[k].numel() < total:
            cap = max(total, 1) * 5 // 4
            self.host[k] = torch.empty(cap, dtype=torch.int32).pin_memory()
            self.dev[k] = torch.empty(cap, dtype=torch.int32, device=self.device)
            self.done[k] = None
        if self.done[k] is not None:
            t1 = time.perf_counter()
            self.done[k].synchronize()          # the previous upload from this staging buffer has been read
            self.timing["stage_sync"] += time.perf_counter() - t1
        self.timing["stage"] += time.perf_counter() - t0
        return {"k": k, "n": n, "seeds": seeds, "total": total, "error": None, "passes": passes, "queue": None}

    def _draw(self, job):
        import ctypes as C
        import time
        t0 = time.perf_counter()
        try:
            lib = _engine.L.load()
            _engine.L.check(lib.mamdr_shuffle_perms(len(job["n"]), job["n"].ctypes.data_as(C.c_void_p), self.sh.buffer_size,
                                                    job["seeds"].ctypes.data_as(C.c_void_p),
                                                    C.c_void_p(self.host[job["k"]].data_ptr())))
            # the passes' device slices (views of the device buffer the upload will fill): built here, off the caller's
            # critical path -- 250 slices cost 1.8 ms of an epoch boundary whose run-ahead margin is 2 - 3 ms
            k, off, q = job["k"], 0, []
            for (d, _), cnt in zip(job["passes"], job["n"]):
                q.append((d, self.dev[k][off:off + int(cnt)]))
                off += int(cnt)
            job["queue"] = q
        except Exception as e:      # surfaces in prepare(), on the caller's thread
            job["error"] = e
        self.timing["draw"] += time.perf_counter() - t0

    def prefetch(self, passes):
        """start drawing the permutations of the epoch whose passes these are; `prepare` of the SAME passes then only
        uploads.  Nothing else may draw from the shuffler in between (the stream order is the pass order): `prepare` checks
        the shuffler's counter and raises if anything did."""
        import threading
        if not self.sh.shuffle:
            return
        self.cancel()
        passes = list(passes)
        job = self._stage(passes)
        job["counter_after"] = self.sh.counter      # (checked in prepare: nobody else drew from the stream in between)
        t = threading.Thread(target=self._draw, args=(job,))
        t.start()
        self.pending = (passes, job, t)

    def cancel(self):
        """drop a prefetch nobody asked for (training stopped early, another plan came): its seeds go back."""
        if self.pending is not None:
            passes, job, t = self.pending
            t.join()
            self.sh.counter -= len(passes)
            self.k ^= 1
            self.pending = None

    def prepare(self, passes):
        """passes: [(domain, max_steps)] in execution order (epoch_passes) for THIS rank."""
        torch = self.torch
        sh = self.sh
        self.pos = 0
        if not sh.shuffle:
            self.queue = [(d, None) for d, _ in passes]
            return
        passes = list(passes)
        if self.pending is not None and self.pending[0] == passes:
            _, job, t = self.pending
            self.pending = None
            if sh.counter != job["counter_after"]:
                # (a validation / target / closing pass that called the shuffler directly after the prefetch -- the
                # lookahead draws the NEXT epoch's seeds while this one still runs -- would silently reorder the seed stream)
                raise RuntimeError("the shuffle stream was drawn from between prefetch() and prepare(): %d draws since" % (
                    sh.counter - job["counter_after"]))
            import time
            t0 = time.perf_counter()
            t.join()
            self.timing["join"] += time.perf_counter() - t0
        else:
            self.cancel()
            job = self._stage(passes)
            self._draw(job)
        if job["error"] is not None:
            raise job["error"]
        k, n, total = job["k"], job["n"], job["total"]
        import time
        t0 = time.perf_counter()
        self.dev[k][:total].copy_(self.host[k][:total], non_blocking=True)
        self.done[k] = torch.cuda.Event()
        self.done[k].record(torch.cuda.current_stream(self.device))
        t1 = time.perf_counter()
        self.queue = job["queue"]
        self.timing["upload"] += t1 - t0
        self.timing["queue"] += time.perf_counter() - t1
        if self.lookahead is not None and not self.queue:
            self.lookahead()        # (a rank / lane without passes this epoch plans the next one like its peers: ADVICE r05)

    def peek(self, domains):
        """the permutations the next len(domains) calls will hand out, without consuming them (meta.PassWindow gathers
        those passes' rows ahead of their calls); None if the epoch's next passes are not over these domains."""
        nxt = self.queue[self.pos:self.pos + len(domains)]
        if len(nxt) != len(domains) or any(dd != d for (dd, _), d in zip(nxt, domains)):
            return None
        return [perm for _, perm in nxt]

    def __call__(self, d, window=None):
        if window is not None:
            raise ValueError("EpochShuffles serves whole-split passes only")
        dd, perm = self.queue[self.pos]
        if dd != d:
            raise RuntimeError("pass %d of the epoch is over domain %d, not %d" % (self.pos, dd, d))
        self.pos += 1
        if self.lookahead is not None and self.pos == min(8, len(self.queue)):
            # a few passes of this epoch are enqueued (the stream has fresh work): the caller's hook plans the NEXT epoch and
            # calls prefetch() now, so that its draw runs beside the whole of this epoch's enqueueing instead of after it
            self.lookahead()
        return perm


def plan_steps(plan, steps_per_domain, domain_regulation_step=0):
    """number of domain-steps a plan executes (metric accounting)."""
    n = sum(steps_per_domain[d] for d in plan["seq"])
    for q, support in plan["dr"]:
        qs = steps_per_domain[q]
        if domain_regulation_step and domain_regulation_step > 0:
            qs = min(qs, domain_regulation_step)
        n += sum(steps_per_domain[j] + qs for j in support)
    return n
