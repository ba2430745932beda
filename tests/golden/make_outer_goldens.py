"""Generate tests/golden/outer_goldens.npz from the reference's OWN numpy methods.

Runs only in the build container (needs /root/reference; it never travels to the
GPU box -- only the .npz does).  tensorflow / deepctr are not installable, so
they are replaced by MagicMock modules; the methods exercised here are pure
numpy and never touch them:

  DomainNegotiation._update_meta_weight      model_zoo/domain_negotiation.py:118-123
  Reptile._update_meta_weight / _accumulate_grad / _update_meta_weight_by_grads
                                             model_zoo/reptile.py:127-142
  MAMDR._update_meta_weight / _update_domain_weights / _accumulate_grad /
        _update_meta_weight_by_grads         model_zoo/mamdr.py:168-196
  SpecificBase._merge_weights                model_zoo/specific_base_model.py:164-172

Usage:  python tests/golden/make_outer_goldens.py
"""
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "outer_goldens.npz")


class _StubFinder(object):
    """serve MagicMock modules for tensorflow.* / deepctr.* / tqdm / sklearn-free imports."""
    PREFIXES = ("tensorflow", "deepctr", "tqdm")

    def find_module(self, name, path=None):
        return self if name.split(".")[0] in self.PREFIXES else None

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.PREFIXES:
            import importlib.machinery
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        # classes used as base classes must be real types
        m.Metric = type("Metric", (object,), {})
        m.Layer = type("Layer", (object,), {})
        return m

    def exec_module(self, module):
        pass


def _load_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    from model_zoo.domain_negotiation import DomainNegotiation
    from model_zoo.reptile import Reptile
    from model_zoo.mamdr import MAMDR
    return DomainNegotiation, Reptile, MAMDR


def _dummy(cls, new_vars, train_config):
    """an instance without __init__; `_get_meta_weights` returns the 'model' weights."""
    obj = cls.__new__(cls)
    obj.__dict__["base_model"] = types.SimpleNamespace(train_config=train_config)
    obj.__dict__["train_config"] = train_config
    obj.__dict__["_get_meta_weights"] = lambda: [a.copy() for a in new_vars]
    return obj


def main():
    DN, Reptile, MAMDR = _load_reference()
    rs = np.random.RandomState(20230401)
    shapes = [(7, 128), (48, 32), (256,), (64, 1), (1,)]

    def rand(scale=1.0):
        return [(rs.standard_normal(s) * scale).astype(np.float32) for s in shapes]

    theta, new, phi = rand(0.1), rand(0.1), rand(0.01)
    # make some entries awkward: denormal-ish, large, equal
    theta[2][:4] = np.array([1e-30, -3e38, 0.0, 1.0], np.float32)
    new[2][:4] = np.array([-1e-30, 3e38, -0.0, 1.0], np.float32)
    out = {}

    def put(prefix, arrs):
        for i, a in enumerate(arrs):
            out["%s_%d" % (prefix, i)] = a

    put("theta", theta)
    put("new", new)
    put("phi", phi)
    out["n_tensors"] = np.array(len(shapes))

    for lr_name, lr in (("lr0p1", 0.1), ("lr1", 1), ("lr0p5", 0.5)):
        cfg = {"meta_learning_rate": lr, "merged_method": "plus", "sample_num": 5}
        # DN / Reptile: old += (new - old) * lr
        t = [a.copy() for a in theta]
        with np.errstate(all="ignore"):
            _dummy(DN, new, cfg)._update_meta_weight(t)
        put("dn_" + lr_name, t)
        t = [a.copy() for a in theta]
        with np.errstate(all="ignore"):
            _dummy(Reptile, new, cfg)._update_meta_weight(t)
        put("reptile_" + lr_name, t)
        # MAMDR DN phase: update_vars += (new - update_vars) * lr
        t = [a.copy() for a in theta]
        with np.errstate(all="ignore"):
            _dummy(MAMDR, new, cfg)._update_meta_weight(t, meta_lr=lr)
        put("mamdr_dn_" + lr_name, t)
        # MAMDR DR phase: phi += (new - merged) * lr
        for method in ("plus", "times"):
            cfg_m = dict(cfg, merged_method=method)
            m = _dummy(MAMDR, new, cfg_m)
            with np.errstate(all="ignore"):
                merged = m._merge_weights(theta, phi)
                p = [a.copy() for a in phi]
                m._update_meta_weight(p, merged, meta_lr=lr)
            put("merged_%s" % method, merged)
            put("mamdr_dr_%s_%s" % (method, lr_name), p)

    cfg = {"meta_learning_rate": 0.1, "merged_method": "plus", "sample_num": 5}
    # Reptile batch variant: two accumulations then apply
    new2 = rand(0.1)
    put("new2", new2)
    acc = [np.zeros_like(a) for a in theta]
    t = [a.copy() for a in theta]
    with np.errstate(all="ignore"):
        _dummy(Reptile, new, cfg)._accumulate_grad(acc, t)
        _dummy(Reptile, new2, cfg)._accumulate_grad(acc, t)
        put("reptile_acc", [a.copy() for a in acc])
        _dummy(Reptile, new, cfg)._update_meta_weight_by_grads(acc, t)
    put("reptile_batch", t)
    put("reptile_acc_after", acc)

    # MAMDR batch variant (plus / times) and phi = new - merged
    for method in ("plus", "times"):
        cfg_m = dict(cfg, merged_method=method)
        m1, m2 = _dummy(MAMDR, new, cfg_m), _dummy(MAMDR, new2, cfg_m)
        with np.errstate(all="ignore"):
            merged = m1._merge_weights(theta, phi)
            acc = [np.zeros_like(a) for a in theta]
            m1._accumulate_grad(acc, merged, theta)
            m2._accumulate_grad(acc, merged, theta)
            put("mamdr_acc_%s" % method, [a.copy() for a in acc])
            p = [a.copy() for a in phi]
            m1._update_meta_weight_by_grads(acc, p)
            put("mamdr_batch_%s" % method, p)
            dw = [a.copy() for a in phi]
            m1._update_domain_weights(dw, merged)
            put("mamdr_domain_weights_%s" % method, dw)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
