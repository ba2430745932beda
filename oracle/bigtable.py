"""Dense TF1 Adam over table-sized tensors without table-sized temporaries (test infrastructure).

oracle/tower.py and oracle/star.py form the gradient of a trainable embedding table the way the reference's
graph does: a dense `[n, 128]` tensor `g = scatter_add(d loss / d x rows) + 2 l2 W`, fed to
tf.train.AdamOptimizer's dense kernel, which moves EVERY row every step.  Written down literally that costs a
float64 table (456 MB for Amazon-6's user table), a dozen single-threaded passes over 79 M - 92 M floats and
seconds per step -- too slow to run the hundreds of steps a trained-model comparison at the BASELINE configs'
own table sizes needs (tests/test_gpu_fullsize.py).

`RowGrad` keeps the SAME gradient as (touched rows, their float64-accumulated sums rounded to fp32, the
regulariser coefficient); `adam_rows` / `sgd_rows` apply the SAME per-element expressions as
oracle/tower.Optimizer in row blocks on a thread pool (numpy's ufuncs release the GIL).  Every element sees
the identical sequence of fp32 operations, so the result is bit-identical to the dense formulation whatever
the block size or thread count (tests/test_oracle_tower.py::test_rowgrad_adam_is_bitwise_the_dense_formula).
`table_sumsq` (the regulariser's value in the reported loss) adds per-block float64 sums: its fp32 rounding
can differ from numpy's single pairwise sum in the last place; no gradient depends on it.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

F32 = np.float32

# tables with at least this many elements take the RowGrad path (tests lower it to cross-check the two paths)
MIN_ELEMENTS = 1 << 22

_pool = None


def threads():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(32, n))


def pool():
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(threads())
    return _pool


def _blocks(n_rows):
    per = max(1024, -(-n_rows // (4 * threads())))
    return [(r0, min(n_rows, r0 + per)) for r0 in range(0, n_rows, per)]


def _run(fn, n_rows):
    blocks = _blocks(n_rows)
    if len(blocks) == 1:
        return [fn(*blocks[0])]
    return list(pool().map(lambda b: fn(*b), blocks))


def use_rows(table):
    return table.size >= MIN_ELEMENTS


class RowGrad(object):
    """g = float32(scatter_add in float64 of `contrib` at `ids`) + reg * table, held sparsely.

    np.add.at on a float64 zero table adds a row's contributions in batch order; here the same additions run in
    the same order on a compact [n_unique, E] float64 array."""

    def __init__(self, table, ids, contrib, reg):
        self.table = table
        rows, inv = np.unique(ids, return_inverse=True)
        acc = np.zeros((rows.shape[0], table.shape[1]), np.float64)
        np.add.at(acc, inv, contrib.astype(np.float64))
        self.rows = rows.astype(np.int64)
        self.sums = acc.astype(F32)
        self.reg = F32(reg)

    def block(self, r0, r1):
        """the dense gradient of rows [r0, r1)."""
        p = self.table[r0:r1]
        g = (self.reg * p).astype(F32) if self.reg != 0 else np.zeros_like(p)
        lo, hi = np.searchsorted(self.rows, (r0, r1))
        if hi > lo:
            idx = self.rows[lo:hi] - r0
            g[idx] = (self.sums[lo:hi] + g[idx]).astype(F32)
        return g

    def dense(self):
        return self.block(0, self.table.shape[0])


def densify(g):
    return g.dense() if isinstance(g, RowGrad) else g


def adam_rows(p, m, v, grad, alpha, omb1, omb2, eps):
    """oracle/tower.Optimizer.adam's three statements, per row block."""
    def work(r0, r1):
        g = grad.block(r0, r1)
        pp, mm, vv = p[r0:r1], m[r0:r1], v[r0:r1]
        t = g - mm
        t *= omb1
        mm += t                          # m += ((g - m) * (1 - beta1))
        np.multiply(g, g, out=t)
        t -= vv
        t *= omb2
        vv += t                          # v += ((g * g - v) * (1 - beta2))
        np.multiply(mm, alpha, out=t)
        np.sqrt(vv, out=g)
        g += eps
        t /= g
        pp -= t                          # p -= (m * alpha) / (sqrt(v) + eps)
    _run(work, p.shape[0])


def sgd_rows(p, grad, lr):
    def work(r0, r1):
        g = grad.block(r0, r1)
        g *= lr
        p[r0:r1] -= g
    _run(work, p.shape[0])


def table_sumsq(table):
    """float32(sum of squares accumulated in float64), block-wise."""
    if table.ndim < 2 or not use_rows(table):
        return F32(np.sum(np.square(table, dtype=F32), dtype=np.float64))
    parts = _run(lambda r0, r1: np.sum(np.square(table[r0:r1], dtype=F32), dtype=np.float64), table.shape[0])
    return F32(np.sum(np.array(parts, np.float64)))
