"""Generate tests/golden/pcgrad_goldens.npz from the reference's OWN numpy method
`PCGrad.PCGrad(final_grads, current_grads, aux_grads)` (model_zoo/pcgrad.py:152-160).

Runs only in the build container (needs /root/reference; only the .npz travels).  tensorflow / deepctr are
replaced by MagicMock modules exactly as in make_outer_goldens.py; the method is pure numpy.  As in the
reference's train loop (pcgrad.py:107-124) `final_grads` IS `current_grads` (the same list object), and two
auxiliary gradients are projected one after the other.

Usage:  python tests/golden/make_pcgrad_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_outer_goldens as mog  # noqa: E402  (stub finder)

OUT = os.path.join(HERE, "pcgrad_goldens.npz")
# the variable shapes of the tower (rows of 256 / 128 / 64 / 1 elements, 1-d biases, an embedding table)
SHAPES = [(6, 128), (12, 256), (9, 128), (5, 64), (256,), (128,), (64,), (64, 1), (1,)]


def main():
    sys.meta_path.insert(0, mog._StubFinder())
    sys.path.insert(0, mog.REF)
    from model_zoo.pcgrad import PCGrad
    obj = PCGrad.__new__(PCGrad)
    rs = np.random.RandomState(20231102)

    def rand():
        return [(rs.standard_normal(s) * 0.01).astype(np.float32) for s in SHAPES]

    current, aux1, aux2 = rand(), rand(), rand()
    # make sure both branches occur in every 2-d tensor and in the 1-d ones
    aux1[4] = (np.abs(aux1[4]) * np.sign(current[4])).astype(np.float32)      # dot > 0
    aux1[5] = (-np.abs(aux1[5]) * np.sign(current[5])).astype(np.float32)     # dot < 0
    out = {"n_tensors": np.array(len(SHAPES))}
    for i, (c, a1, a2) in enumerate(zip(current, aux1, aux2)):
        out["current_%d" % i], out["aux1_%d" % i], out["aux2_%d" % i] = c.copy(), a1.copy(), a2.copy()
    final = current                                 # the same list object, as in pcgrad.py:108-109
    with np.errstate(all="ignore"):
        obj.PCGrad(final, current, aux1)
        for i in range(len(SHAPES)):
            out["after1_final_%d" % i], out["after1_aux_%d" % i] = final[i].copy(), aux1[i].copy()
        obj.PCGrad(final, current, aux2)
        for i in range(len(SHAPES)):
            out["after2_final_%d" % i], out["after2_aux_%d" % i] = final[i].copy(), aux2[i].copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
