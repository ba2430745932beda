"""Worker-process pool for the heavy oracle runs of the GPU parity tests (test infrastructure; see tests/oracle_jobs.py
for the jobs).  This module imports nothing heavy at load time ON PURPOSE: a spawned worker unpickles `_init_worker`
by importing this module, and the worker must choose its CPUs BEFORE numpy is imported -- OpenBLAS creates its threads at
import and they keep the affinity mask of that moment.

CPU layout (measured in round 5 on the 256-thread GPU box): 18 oracle jobs started at once slowed each other 3 - 6x (the
whole-pipeline twin 67 s alone, 395 s in the crowd) and slowed the main process's own in-test oracle runs with them.
So the pool is topology-aware: the main pytest process keeps the first MAIN_CORES physical cores (both hardware threads
of each), every worker gets its own block of physical cores (one hardware thread per core) and caps BLAS at 4 threads
(4 threads: 4.1 / 15.5 ms per oracle step at bs 1,024 / 4,096; 8 threads: 3.8 / 12.7 ms -- profiles/r05_oracle_threads.txt).
"""
import os
import time

MAIN_CORES = 16
WORKER_BLAS_THREADS = 4

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
            per = max(blas_threads, min(8, len(free) // max(1, n_workers)))
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
    """keys: [(job name, kwargs)].  One worker per job, up to 32 (they all start at once)."""
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
        n_cores = len(physical_cores())
        allc = os.environ.get("MAMDR_TEST_ALL_CPUS")
        if allc:
            n_cores = max(n_cores, len(allc.split(",")) // 2)
        n = max(1, min(len(uniq), 32, max(1, (n_cores - MAIN_CORES) // WORKER_BLAS_THREADS)))
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
