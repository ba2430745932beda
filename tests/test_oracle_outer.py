"""Pin oracle/outer.py bit-for-bit against vectors produced by the reference's own
numpy methods (tests/golden/make_outer_goldens.py)."""
import os

import numpy as np
import pytest

from oracle import outer

F32 = np.float32


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "outer_goldens.npz"))


def tensors(G, prefix):
    return [G["%s_%d" % (prefix, i)] for i in range(int(G["n_tensors"]))]


def same_bits(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


LRS = (("lr0p1", 0.1), ("lr1", 1), ("lr0p5", 0.5))


@pytest.mark.parametrize("lr_name,lr", LRS)
def test_dn_reptile_update(G, lr_name, lr):
    for prefix in ("dn_", "reptile_", "mamdr_dn_"):
        for th, nw, want in zip(tensors(G, "theta"), tensors(G, "new"), tensors(G, prefix + lr_name)):
            with np.errstate(all="ignore"):
                got = outer.dn_update(th.copy(), nw, lr)
            assert same_bits(got, want)


@pytest.mark.parametrize("method", ("plus", "times"))
def test_merge(G, method):
    for th, ph, want in zip(tensors(G, "theta"), tensors(G, "phi"), tensors(G, "merged_" + method)):
        with np.errstate(all="ignore"):
            assert same_bits(outer.merge(th, ph, method), want)


@pytest.mark.parametrize("method", ("plus", "times"))
@pytest.mark.parametrize("lr_name,lr", LRS)
def test_mamdr_dr_update(G, method, lr_name, lr):
    for th, ph, nw, want in zip(tensors(G, "theta"), tensors(G, "phi"), tensors(G, "new"),
                                tensors(G, "mamdr_dr_%s_%s" % (method, lr_name))):
        with np.errstate(all="ignore"):
            merged = outer.merge(th, ph, method)
            got = outer.mamdr_update(ph.copy(), nw, merged, lr)
        assert same_bits(got, want)


def test_reptile_batch(G):
    for th, n1, n2, wacc, want, wafter in zip(tensors(G, "theta"), tensors(G, "new"), tensors(G, "new2"),
                                              tensors(G, "reptile_acc"), tensors(G, "reptile_batch"),
                                              tensors(G, "reptile_acc_after")):
        with np.errstate(all="ignore"):
            acc = np.zeros_like(th)
            t = th.copy()
            outer.reptile_accumulate(acc, n1, t)
            outer.reptile_accumulate(acc, n2, t)
            assert same_bits(acc, wacc)
            outer.reptile_apply(t, acc, 0.1)
        assert same_bits(t, want)
        assert same_bits(acc, wafter)


@pytest.mark.parametrize("method", ("plus", "times"))
def test_mamdr_batch_and_domain_weights(G, method):
    for th, ph, n1, n2, wacc, want, wdw in zip(tensors(G, "theta"), tensors(G, "phi"), tensors(G, "new"),
                                               tensors(G, "new2"), tensors(G, "mamdr_acc_" + method),
                                               tensors(G, "mamdr_batch_" + method),
                                               tensors(G, "mamdr_domain_weights_" + method)):
        with np.errstate(all="ignore"):
            merged = outer.merge(th, ph, method)
            acc = np.zeros_like(th)
            outer.mamdr_accumulate(acc, n1, merged, th, method)
            outer.mamdr_accumulate(acc, n2, merged, th, method)
            assert same_bits(acc, wacc)
            p = ph.copy()
            outer.mamdr_apply_grads(p, acc, 5, 0.1)
            assert same_bits(p, want)
            assert same_bits(outer.mamdr_domain_weights(n1, merged), wdw)


def test_pcgrad_projection_matches_reference_goldens(golden_dir):
    """oracle.outer.pcgrad_project against vectors produced by the reference's own PCGrad.PCGrad
    (tests/golden/make_pcgrad_goldens.py): two successive projections, every tensor shape of the tower."""
    import os
    from oracle import outer as oouter
    G = np.load(os.path.join(golden_dir, "pcgrad_goldens.npz"))
    n = int(G["n_tensors"])
    cur = [G["current_%d" % i].copy() for i in range(n)]
    for tag, step in (("aux1", "after1"), ("aux2", "after2")):
        aux = [G["%s_%d" % (tag, i)].copy() for i in range(n)]
        oouter.pcgrad_project(cur, aux)
        for i in range(n):
            assert np.array_equal(cur[i].view(np.uint32), G["%s_final_%d" % (step, i)].view(np.uint32)), (step, i)
            assert np.array_equal(aux[i].view(np.uint32), G["%s_aux_%d" % (step, i)].view(np.uint32)), (step, i)
    # both branches were exercised
    d0 = np.sum(G["current_1"] * G["aux1_1"], axis=-1)
    assert (d0 > 0).any() and (d0 <= 0).any()
