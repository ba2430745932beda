import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# On the GPU box (256 hardware threads) the main process keeps to a block of cores of its own and the oracle workers to
# theirs (tests/oracle_pool.py) -- decided BEFORE numpy / torch create their thread pools.
import oracle_pool  # noqa: E402
if (os.cpu_count() or 1) >= 64 and not os.environ.get("MAMDR_TEST_NO_PINNING"):
    oracle_pool.pin_main()

# The numpy oracle's GEMMs are small (1,024 .. 8,192 rows x 384 .. 64 columns): on the GPU box (256 hardware threads,
# OpenBLAS default 64) they ran 4 - 6x SLOWER than on 8 threads -- 14 / 72 ms per oracle step at bs 1,024 / 4,096 against
# 3.8 / 12.7 ms (profiles/r05_oracle_threads.txt, tests/diag_oracle_threads.py) -- the container's CPU quota of 16 CPUs
# throttles what runs beyond it (oracle_pool.cpu_quota).  4 threads where the quota is <= 16 CPUs (the workers take the
# rest), 8 otherwise.  Results do not depend on it beyond BLAS' own blocking (the bars of the parity tests are ~1e-3,
# rounding ~1e-7).
BLAS_THREADS = int(os.environ.get("MAMDR_TEST_BLAS_THREADS", str(oracle_pool.budget()[0])))
try:
    from threadpoolctl import threadpool_limits
    _blas_limit = threadpool_limits(limits=BLAS_THREADS, user_api="blas")      # kept for the whole session
except Exception:                                                              # pragma: no cover
    _blas_limit = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "oracle_job(name, **kwargs): the test reads the result of tests/oracle_jobs.py's "
                                       "job `name`; selected jobs start in worker processes when the session starts")


def pytest_collection_modifyitems(config, items):
    """tests that wait for an oracle job run LAST (stable otherwise): the jobs compute while the rest of the suite runs."""
    items.sort(key=lambda it: 1 if any(True for _ in it.iter_markers("oracle_job")) else 0)


def pytest_collection_finish(session):
    """start the heavy oracle runs of the SELECTED tests in worker processes, all at once, so that they compute
    while the HIP side of the suite runs (tests/oracle_jobs.py).  Only where a GPU is visible: without one the tests
    that would read them skip.  (device_count() does not initialise the GPU; the workers are spawned, not forked.)"""
    keys = []
    for item in session.items:
        for m in item.iter_markers("oracle_job"):
            keys.append((m.args[0], dict(m.kwargs)))
    if not keys or os.environ.get("MAMDR_TEST_NO_ORACLE_POOL"):
        return
    try:
        import torch
        if torch.cuda.device_count() == 0:
            return
    except Exception:
        return
    import oracle_jobs      # (tests/ is on sys.path: rootdir conftest)
    oracle_jobs.start(keys)


def pytest_sessionstart(session):
    """the teacher-forced tests' per-pass oracle states (tests/oracle_jobs.PassDump, up to ~1 GB for a Taobao-30 epoch) live
    in one directory per session, removed when the session ends whatever the tests did."""
    import tempfile
    if "MAMDR_TEST_DUMP_ROOT" not in os.environ:
        os.environ["MAMDR_TEST_DUMP_ROOT"] = session.config._mamdr_dump_root = tempfile.mkdtemp(prefix="mamdr_tf_session_")


def pytest_sessionfinish(session, exitstatus):
    oracle_pool.shutdown()
    root = getattr(session.config, "_mamdr_dump_root", None)
    if root:
        import shutil
        shutil.rmtree(root, ignore_errors=True)
        os.environ.pop("MAMDR_TEST_DUMP_ROOT", None)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
