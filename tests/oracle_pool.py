"""Worker-process pool for the heavy oracle runs of the GPU parity tests (test infrastructure; see tests/oracle_jobs.py
for the jobs).  This module imports nothing heavy at load time ON PURPOSE: a spawned worker unpickles `_init_worker`
by importing this module, and the worker must choose its CPUs BEFORE numpy is imported -- OpenBLAS creates its threads at
import and they keep the affinity mask of that moment.

CPU budget (measured in round 5 on the GPU box): 256 hardware threads are visible, but the container's cgroup grants 16
CPUs of time (`cpu_quota`); 18 - 30 oracle jobs started at once slowed each other 3 - 6x (the whole-pipeline twin 44 s
with 6 jobs, 244 s with 30) and the main process's own in-test oracle runs with them -- throttling, not cache or
memory traffic.  So the pool runs as many workers AT A TIME as the quota pays for (2 BLAS threads each, the main process
4: 6 workers on that box), longest jobs first, the rest queue; workers still take their own physical cores (no migrations,
no SMT sharing with the main process).
"""
import os
import time

MAIN_CORES = 16
WORKER_BLAS_THREADS = 2


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 `cpu.max` / v1 cfs quota), None if unlimited.  The GPU box
    shows 256 hardware threads and grants 16 CPUs of time (`cpu.max` = 1600000 100000): more runnable threads than that
    are THROTTLED -- whole 100 ms periods without a time slice -- however they are pinned.  Measured in round 5
    (tests/diag_oracle_crowd.py): the oracle step of one worker 2.1 ms, of 8 / 16 / 28 concurrent workers of 4 threads
    5.6 / 11.1 / 22.3 ms -- total throughput constant.  (It is also what rounds 3 - 4 called the "CPU cliff".)"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            return max(1.0, float(q) / float(per))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        if q > 0:
            return max(1.0, q / per)
    except Exception:
        pass
    return None


def budget():
    """(BLAS threads of the main process, worker count limit): the main process and the workers together stay inside the
    CPU quota (or the visible cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    allc = os.environ.get("MAMDR_TEST_ALL_CPUS")
    if allc:
        n = max(n, len(allc.split(",")))
    q = cpu_quota()
    cpus = n if q is None else min(float(n), q)
    main = 4 if cpus <= 16 else 8
    workers = max(1, int((cpus - main) // WORKER_BLAS_THREADS))
    return main, workers

_pool = None
_futures = {}


def physical_cores():
    """[[hardware threads of core 0], [of core 1], ...] among the CPUs this process may use."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    groups, seen = [], set()
    for c in allowed:
        if c in seen:
            continue
        sib = [c]
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                if "-" in part:
                    a, b = part.split("-")
                    sib += list(range(int(a), int(b) + 1))
                else:
                    sib.append(int(part))
            sib = [s for s in sib if s in allowed] or [c]
        except Exception:
            pass
        groups.append(sorted(sib))
        seen.update(sib)
    return groups


def pin_main():
    """called by conftest BEFORE numpy / torch are imported, on big hosts only: the main process (and the BLAS / OpenMP
    threads it creates later) keep to the first MAIN_CORES physical cores.  Returns the full core list for the workers."""
    cores = physical_cores()
    if len(cores) < 4 * MAIN_CORES:
        return None
    try:
        os.environ["MAMDR_TEST_ALL_CPUS"] = ",".join(str(c) for g in cores for c in g)
        os.sched_setaffinity(0, [c for g in cores[:MAIN_CORES] for c in g])
    except Exception:
        return None
    return cores


def _init_worker(counter, n_workers, blas_threads):
    try:
        with counter.get_lock():
            idx = counter.value
            counter.value += 1
        allc = os.environ.get("MAMDR_TEST_ALL_CPUS")
        if allc:                                    # the parent pinned itself: widen to the whole machine first
            os.sched_setaffinity(0, [int(c) for c in allc.split(",")])
        cores = physical_cores()
        if len(cores) >= 4 * MAIN_CORES:
            free = cores[MAIN_CORES:]
            per = max(blas_threads, min(4, len(free) // max(1, n_workers)))
            lo = (idx * per) % max(1, len(free) - per + 1)
            os.sched_setaffinity(0, [g[0] for g in free[lo:lo + per]])
    except Exception:
        pass
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(blas_threads))
    os.environ.setdefault("OMP_NUM_THREADS", str(blas_threads))
    try:
        from threadpoolctl import threadpool_limits
        import numpy  # noqa: F401
        globals()["_limit"] = threadpool_limits(limits=blas_threads, user_api="blas")
    except Exception:
        pass


def _key(name, kwargs):
    return (name,) + tuple(sorted(kwargs.items()))


def start(keys, cost=None):
    """keys: [(job name, kwargs)].  As many workers as the CPU budget pays for; the jobs queue, longest first."""
    global _pool
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    uniq = []
    for name, kw in keys:
        if _key(name, kw) not in [_key(*u) for u in uniq]:
            uniq.append((name, kw))
    todo = [u for u in uniq if _key(*u) not in _futures]
    if not todo:
        return
    if _pool is None:
        n = max(1, min(len(uniq), budget()[1]))
        ctx = mp.get_context("spawn")
        _pool = ProcessPoolExecutor(max_workers=n, mp_context=ctx, initializer=_init_worker,
                                    initargs=(ctx.Value("i", 0), n, WORKER_BLAS_THREADS))
    cost = cost or {}
    todo.sort(key=lambda u: -cost.get(u[0], 1.0) * float(u[1].get("epochs", 1)))       # longest first
    for name, kw in todo:
        _futures[_key(name, kw)] = _pool.submit(_run, name, kw)


def _run(name, kw):
    import oracle_jobs
    t0 = time.time()
    out = oracle_jobs.JOBS[name](**kw)
    out["job_seconds"] = time.time() - t0
    return out


def result(job, **kw):
    """the job's result: from the pool if the session started it, else computed here."""
    f = _futures.get(_key(job, kw))
    if f is None:
        return _run(job, kw)
    t0 = time.time()
    out = f.result(timeout=1500)
    out["waited_seconds"] = time.time() - t0
    return out


def shutdown():
    global _pool
    if _pool is not None:
        for f in _futures.values():
            f.cancel()
        procs = list(getattr(_pool, "_processes", {}).values())
        _pool.shutdown(wait=False, cancel_futures=True)
        for p in procs:                  # a job still running when the session ends (a failed -x run) is not waited for
            if p.is_alive():
                p.terminate()
        _pool = None
    _futures.clear()
