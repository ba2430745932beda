"""Pin oracle/auc.py on the reference's known-answer vector (utils/auc.py:46-55)."""
import numpy as np

from oracle import auc


def test_reference_docstring_example():
    m = auc.AUC(num_thresholds=3)
    np.testing.assert_array_equal(m.thr, np.array([-1e-7, 0.5, 1 + 1e-7], np.float32))
    m.update_state(np.array([0, 0, 1, 1]), np.array([0, 0.5, 0.3, 0.9], np.float32))
    np.testing.assert_array_equal(m.tp, [2, 1, 0])
    np.testing.assert_array_equal(m.fp, [2, 0, 0])
    np.testing.assert_array_equal(m.fn, [0, 1, 2])
    np.testing.assert_array_equal(m.tn, [0, 2, 2])
    assert float(m.result()) == 0.75


def test_thresholds_500():
    t = auc.thresholds(500)
    assert t.shape == (500,) and t.dtype == np.float32
    assert t[0] == np.float32(-1e-7) and t[-1] == np.float32(1 + 1e-7)
    assert t[1] == np.float32(1.0 / 499) and t[498] == np.float32(498.0 / 499)
    assert np.all(np.diff(t) > 0)


def test_batched_equals_unbatched_and_tracks_exact_auc():
    rs = np.random.RandomState(0)
    y = (rs.rand(5000) < 0.3).astype(np.float32)
    p = np.clip(0.3 + 0.2 * (y - 0.3) + 0.2 * rs.randn(5000), 0, 1).astype(np.float32)
    a1 = auc.auc500(y, p)
    a2 = auc.auc500(y, p, batch_size=1024)
    assert a1 == a2
    # exact (rank) AUC within the discretisation error of 500 thresholds
    order = np.argsort(p, kind="mergesort")
    ranks = np.empty(5000)
    ranks[order] = np.arange(1, 5001)
    npos = y.sum()
    exact = (ranks[y == 1].sum() - npos * (npos + 1) / 2) / (npos * (5000 - npos))
    assert abs(float(a1) - exact) < 5e-3


def test_degenerate_all_one_class():
    assert float(auc.auc500(np.zeros(10), np.linspace(0, 1, 10).astype(np.float32))) == 0.0
    assert float(auc.auc500(np.ones(10), np.linspace(0, 1, 10).astype(np.float32))) == 0.0
