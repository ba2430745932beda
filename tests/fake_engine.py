"""CPU stand-in for mamdr_amd.engine.TowerEngine, built on the oracle (tests only).

Lets the host logic (registry, wrappers, meta loops, sharding, result writing) run in the
`-m "not gpu"` suite.  It is NOT a product fallback: mamdr_amd never imports it.
"""
import numpy as np
import torch

from oracle import auc as oauc
from oracle import outer as oouter
from oracle import tower as otower

F32 = np.float32


class FakeEngine(object):
    def __init__(self, n_user, n_item, n_domain, batch_size, dropout=0.5, emb_trainable=False, tower="mlp",
                 emb_dim=128, hidden=(256, 128, 64), l2_emb=1e-5, device=None, dropout_seed=1024, l2_linear=1e-5,
                 uncertainty_weight=False):
        if tower not in ("mlp", "deepfm", "wdl"):
            raise NotImplementedError(tower)
        self.tower = tower
        self.n_user, self.n_item, self.n_domain = n_user, n_item, n_domain
        self.batch_size = batch_size
        self.device = torch.device("cpu")
        self.dropout_seed = dropout_seed
        rs = np.random.RandomState(0)
        params = otower.init_params(rs, n_user, n_item, n_domain, emb_dim, hidden)
        l2 = {} if (l2_emb == 1e-5 and l2_linear == 1e-5) else dict(l2_emb=l2_emb, l2_linear=l2_linear)
        self.oracle = otower.OracleModel(params, emb_trainable=emb_trainable, dropout=dropout, hidden=hidden,
                                         dropout_seed=dropout_seed, tower=tower, uncertainty=uncertainty_weight, **l2)
        self.segments = {}
        off = 0
        for name in self.oracle.names:
            self.segments[name] = (off, params[name].size)
            off += params[name].size
        self.n_params = off
        self.n_meta = off
        self.data = {}
        self.calls = []
        self._ema = None

    def compile(self, optimizer):
        from mamdr_amd.engine import FlatVectorOps
        FlatVectorOps.compile(self, optimizer)

    # flat vectors
    def keras_name(self, segment):
        return segment

    def new_vector(self, like=None, meta=False):
        return like.clone() if like is not None else torch.zeros(self.n_meta if meta else self.n_params, dtype=torch.float32)

    def pack(self, named):
        return torch.from_numpy(np.concatenate([np.asarray(named[n], F32).ravel() for n in self.segments]))

    def unpack(self, vec):
        h = vec.numpy()
        return {n: h[o:o + c].copy() for n, (o, c) in self.segments.items()}

    @property
    def weights(self):
        return torch.from_numpy(self.oracle.get_flat())

    meta_off = 0

    meta_holes = ()

    def set_meta_range(self, off, count, holes=()):
        self.meta_off, self.n_meta = int(off), int(count)
        self.meta_holes = tuple((int(o), int(c)) for o, c in holes)

    def assign_meta(self, vec):
        assert vec.numel() == self.n_meta
        live = self.meta_weights
        for o, c in self.meta_holes:
            vec[o:o + c] = live[o:o + c]
        full = self.oracle.get_flat()
        full[self.meta_off:self.meta_off + self.n_meta] = vec.numpy()
        self.oracle.set_flat(full)

    @property
    def meta_weights(self):
        return self.weights[self.meta_off:self.meta_off + self.n_meta]

    def set_weights(self, vec):
        n = vec.numel()
        if n == self.n_params:
            self.oracle.set_flat(vec.numpy().copy())
            return
        full = self.oracle.get_flat()
        off = self.meta_off if n == self.n_meta else 0
        full[off:off + n] = vec.numpy()
        self.oracle.set_flat(full)

    def get_weights(self, out=None):
        w = self.weights
        if out is None:
            return w
        out.copy_(w)
        return out

    def interp(self, dst, a, b, scale):
        an, bn = a.numpy().copy(), b.numpy().copy()
        oouter.mamdr_update(dst.numpy(), an, bn, scale)

    def merge(self, dst, theta, phi, method="plus"):
        dst.copy_(torch.from_numpy(oouter.merge(theta.numpy(), phi.numpy(), method)))

    def dr_advance(self, phi, merged, theta, gamma, method="plus", assign_model=True):
        self.interp(phi, self.meta_weights, merged, gamma)
        self.merge(merged, theta, phi, method)
        if assign_model:
            self.assign_meta(merged)

    def sub(self, dst, a, b):
        dst.copy_(torch.from_numpy(oouter.mamdr_domain_weights(a.numpy(), b.numpy())))

    def accumulate(self, acc, a, b, shared=None, divisor=1.0):
        oouter.mamdr_accumulate(acc.numpy(), a.numpy().copy(), b.numpy().copy(),
                                None if shared is None else shared.numpy(), "times" if shared is not None else "plus",
                                divisor)

    def apply_accumulated(self, dst, acc, divisor, scale):
        if divisor > 0:
            oouter.mamdr_apply_grads(dst.numpy(), acc.numpy(), divisor, scale)
        else:
            oouter.reptile_apply(dst.numpy(), acc.numpy(), scale)

    def pcgrad_project(self, final, aux, tensors=None):
        from oracle import loops as oloops
        oouter.pcgrad_project(oloops.tensor_views(self.oracle, final.numpy()), oloops.tensor_views(self.oracle, aux.numpy()))

    # binding
    def bind_table(self, name, rows):
        self.oracle.params[name] = np.ascontiguousarray(rows, F32)

    def bind_domain_data(self, domain, split, uid, pid, dom, label):
        self.data[(domain, split)] = {"uid": np.asarray(uid, np.int32), "pid": np.asarray(pid, np.int32),
                                      "domain": np.asarray(dom, np.int32), "label": np.asarray(label, F32)}

    def n_rows(self, domain, split):
        return int(self.data[(domain, split)]["uid"].shape[0])

    # hot path
    def train_steps(self, domain, perm=None, first_step=0, n_steps=None, lr=1e-3, optimizer="adam", loss_out=None,
                    batch_size=None, pass_rows=None):
        bs = batch_size or self.batch_size
        cols = self.data[(domain, "train")]
        n = cols["uid"].shape[0] if pass_rows is None else pass_rows
        p = np.arange(n, dtype=np.int32) if perm is None else np.asarray(perm)
        if n_steps is None:
            n_steps = -(-n // bs) - first_step
        if optimizer == "adam" and getattr(self, "compiled", ("adam", None))[0] != "adam":
            optimizer, lr = self.compiled
        self.oracle.lr = lr
        self.oracle.use_sgd = optimizer == "sgd"
        for s in range(first_step, first_step + n_steps):
            idx = p[s * bs:(s + 1) * bs]
            if optimizer == "accumulate":
                if self._ema is not None:
                    g = np.zeros(self.n_params, F32)
                    self.oracle.accumulate_on_batch(g, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx],
                                                    cols["label"][idx])
                    self._ema["step"] = oouter.moving_average_update(self._acc.numpy(), self._ema["biased"], g,
                                                                     self._ema["momentum"], self._ema["step"])
                    continue
                self.oracle.accumulate_on_batch(self._acc.numpy(), cols["uid"][idx], cols["pid"][idx],
                                                cols["domain"][idx], cols["label"][idx])
                continue
            self.oracle.train_on_batch(cols["uid"][idx], cols["pid"][idx], cols["domain"][idx], cols["label"][idx])
        self.calls.append((domain, n_steps, optimizer, lr))
        return n_steps

    def evaluate(self, domain, split, want_preds=False):
        cols = self.data[(domain, split)]
        loss, preds = self.oracle.evaluate(cols, self.batch_size)
        return float(loss), float(oauc.auc500(cols["label"], preds, self.batch_size))

    def bind_accumulator(self, acc):
        self._acc = acc

    def set_moving_average(self, momentum):
        self._ema = {"momentum": float(momentum), "step": 0, "biased": np.zeros(self.n_params, F32)}

    def adam_apply(self, p, m, v, g, lr, beta1_power, beta2_power, grad_scale=1.0):
        o = otower.OuterAdam(p.numel())
        o.m, o.v = m.numpy(), v.numpy()
        # powers arrive AFTER this step's update: rewind one step so that OuterAdam.apply reproduces them
        o.b1p, o.b2p = F32(beta1_power), F32(beta2_power)
        alpha = F32(F32(lr) * np.sqrt(F32(1) - o.b2p, dtype=F32) / (F32(1) - o.b1p))
        gg = (g.numpy() * F32(grad_scale)).astype(F32)
        o.m += ((gg - o.m) * F32(F32(1) - otower.BETA1)).astype(F32)
        o.v += ((gg * gg - o.v) * F32(F32(1) - otower.BETA2)).astype(F32)
        p.numpy()[...] -= ((o.m * alpha) / (np.sqrt(o.v, dtype=F32) + otower.ADAM_EPS)).astype(F32)

    def optimizer_reset(self):
        self.oracle.opt = otower.Optimizer(self.oracle.params, self.oracle.names)

    def close(self):
        pass


class FakeGraphEngine(object):
    """CPU stand-in for mamdr_amd.graph_engine.GraphEngine (multi-task towers), built on oracle/mtl.py (tests only)."""
    compiled = ("adam", None)

    def compile(self, optimizer):
        from mamdr_amd.engine import FlatVectorOps
        FlatVectorOps.compile(self, optimizer)

    def __init__(self, kind, n_user, n_item, n_domain, batch_size, expert_hidden, tower_hidden, gate_hidden=(),
                 num_experts=0, shared_expert_num=0, specific_expert_num=0, dropout=0.5, emb_trainable=False, emb_dim=128,
                 l2_emb=1e-5, device=None, dropout_seed=1024):
        from oracle import mtl as omtl
        if emb_trainable:
            raise NotImplementedError("trainable tables")
        self.kind = kind
        self.n_user, self.n_item, self.n_domain = n_user, n_item, n_domain
        self.batch_size = batch_size
        self.device = torch.device("cpu")
        self.dropout_seed = dropout_seed
        self.spec = omtl.Spec(kind, n_domain, expert_hidden, tower_hidden, gate_hidden, num_experts, shared_expert_num,
                              specific_expert_num, emb_dim)
        params = omtl.init_params(np.random.RandomState(0), self.spec, n_user, n_item)
        self.oracle = omtl.OracleMTL(params, self.spec, dropout=dropout, dropout_seed=dropout_seed)
        self.segments, off = {}, 0
        for name in self.oracle.names:
            self.segments[name] = (off, params[name].size)
            off += params[name].size
        self.n_params = self.n_meta = off
        self.aux = None
        self.data, self.calls = {}, []

    def keras_name(self, segment):
        return segment

    def new_vector(self, like=None, meta=False):
        return like.clone() if like is not None else torch.zeros(self.n_params, dtype=torch.float32)

    def pack(self, named):
        return torch.from_numpy(np.concatenate([np.asarray(named[n], F32).ravel() for n in self.segments]))

    def unpack(self, vec):
        h = vec.numpy()
        return {n: h[o:o + c].copy() for n, (o, c) in self.segments.items()}

    @property
    def weights(self):
        return torch.from_numpy(self.oracle.get_flat())

    def set_weights(self, vec):
        self.oracle.set_flat(vec.numpy().copy())

    def get_weights(self, out=None):
        w = self.weights
        if out is None:
            return w
        out.copy_(w)
        return out

    def bind_table(self, name, rows):
        self.oracle.params[name] = np.ascontiguousarray(rows, F32)

    def bind_domain_data(self, domain, split, uid, pid, dom, label):
        self.data[(domain, split)] = {"uid": np.asarray(uid, np.int32), "pid": np.asarray(pid, np.int32),
                                      "domain": np.asarray(dom, np.int32), "label": np.asarray(label, F32)}

    def n_rows(self, domain, split):
        return int(self.data[(domain, split)]["uid"].shape[0])

    def train_steps(self, domain, perm=None, first_step=0, n_steps=None, lr=1e-3, optimizer="adam", loss_out=None,
                    batch_size=None, pass_rows=None):
        bs = batch_size or self.batch_size
        cols = self.data[(domain, "train")]
        n = cols["uid"].shape[0]
        p = np.arange(n, dtype=np.int32) if perm is None else np.asarray(perm)
        if n_steps is None:
            n_steps = -(-n // bs) - first_step
        if optimizer == "adam" and self.compiled[0] != "adam":
            optimizer, lr = self.compiled
        self.oracle.lr = lr
        self.oracle.use_sgd = optimizer == "sgd"
        for s in range(first_step, first_step + n_steps):
            idx = p[s * bs:(s + 1) * bs]
            self.oracle.train_on_batch(domain, cols["uid"][idx], cols["pid"][idx], cols["domain"][idx], cols["label"][idx])
        self.calls.append((domain, n_steps, optimizer, lr))
        return n_steps

    def evaluate(self, domain, split, want_preds=False):
        cols = self.data[(domain, split)]
        loss, preds = self.oracle.evaluate(domain, cols, self.batch_size)
        return float(loss), float(oauc.auc500(cols["label"], preds, self.batch_size))

    def optimizer_reset(self):
        o = self.oracle
        o.m = {n: np.zeros_like(o.params[n]) for n in o.names}
        o.v = {n: np.zeros_like(o.params[n]) for n in o.names}
        o.b1p, o.b2p, o.t = F32(1), F32(1), 0

    def set_adam_eps(self, eps):
        self.oracle.adam_eps = F32(eps)

    def close(self):
        pass


class FakeFmEngine(FakeEngine):
    """CPU stand-in for GraphEngine's single-output kinds (nfm / pnn), built on oracle/fmnets.py (tests only)."""

    def __init__(self, kind, n_user, n_item, n_domain, batch_size, expert_hidden, tower_hidden=(), dropout=0.5,
                 emb_trainable=False, emb_dim=128, **kw):
        from oracle import fmnets as ofm
        if kind not in ("nfm", "pnn"):
            raise NotImplementedError(kind)
        self.tower = self.kind = kind
        self.n_user, self.n_item, self.n_domain = n_user, n_item, n_domain
        self.batch_size = batch_size
        self.device = torch.device("cpu")
        self.dropout_seed = 1024
        params = ofm.init_params(np.random.RandomState(0), kind, n_user, n_item, n_domain, emb_dim, tuple(expert_hidden))
        self.oracle = ofm.OracleNet(params, kind, emb_trainable=emb_trainable, dropout=dropout, hidden=tuple(expert_hidden))
        self.segments, off = {}, 0
        for name in self.oracle.names:
            self.segments[name] = (off, params[name].size)
            off += params[name].size
        self.n_params = self.n_meta = off
        self.data, self.calls = {}, []
        self._ema = None
        self.aux = None


FakeEngine.graph = FakeFmEngine


class FakeStarEngine(FakeEngine):
    """CPU stand-in for TowerEngine(tower="star"), built on oracle/star.OracleStar (tests only): the flat vector = the meta
    prefix [tables if trainable | domain_emb | shared kernels | shared biases] followed by the tensors that stay outside
    theta / phi; `aux` = PartitionedNorm's moving statistics in the HIP engine's layout, aliased to the oracle's state."""

    def __init__(self, n_user, n_item, n_domain, batch_size, dropout=0.0, emb_trainable=False, tower="star", emb_dim=128,
                 hidden=(256, 128, 64), device=None, dropout_seed=1024, **kw):
        from mamdr_amd.engine import TowerEngine
        from oracle import star as ostar
        assert tower == "star" and tuple(hidden) == (256, 128, 64) and emb_dim == 128
        self.tower = "star"
        self.KERAS_NAMES = TowerEngine.KERAS_NAMES
        self.n_user, self.n_item, self.n_domain = n_user, n_item, n_domain
        self.batch_size = batch_size
        self.device = torch.device("cpu")
        self.dropout_seed = dropout_seed
        params = ostar.init_params(np.random.RandomState(0), n_user, n_item, n_domain)
        self.oracle = ostar.OracleStar(params, emb_trainable=emb_trainable, lr=1e-3)
        self.segments, off = {}, 0
        for name in self.oracle.names:
            self.segments[name] = (off, params[name].size)
            off += params[name].size
        self.n_params = off
        self.n_meta = sum(params[n].size for n in self.oracle.meta_names)
        # aux = [mov_mean | mov_var | biased_mean | biased_var] (D x 384 each) | steps (D), padded to 4 floats
        D, X = n_domain, 3 * emb_dim
        self._aux_np = np.zeros((4 * D * X + D + 3) // 4 * 4, F32)
        st = self.oracle.state
        for k, key in enumerate(("mov_mean", "mov_var", "biased_mean", "biased_var")):
            view = self._aux_np[k * D * X:(k + 1) * D * X].reshape(D, X)
            view[...] = st[key]
            st[key] = view
        view = self._aux_np[4 * D * X:4 * D * X + D]
        view[...] = st["steps"]
        st["steps"] = view
        self.aux = torch.from_numpy(self._aux_np)       # shares memory with the oracle's state
        self.data, self.calls = {}, []
        self._ema = None

    def keras_name(self, segment):
        return self.KERAS_NAMES.get(segment, segment)


def fake_graph(kind, *args, **kw):
    """the generic-layer engine's stand-ins by kind: nfm / pnn (oracle/fmnets.py), shared_bottom / mmoe / ple (oracle/mtl.py)."""
    return (FakeFmEngine if kind in ("nfm", "pnn") else FakeGraphEngine)(kind, *args, **kw)


def fake_factory(*args, **kw):
    """engine factory that also serves the Star tower and -- called with a kind name first, as DeepMTLCTR calls its factory --
    the multi-task towers (cli.main(engine_factory=fake_factory))."""
    if args and isinstance(args[0], str):
        return fake_graph(*args, **kw)
    if kw.get("tower") == "star":
        return FakeStarEngine(*args, **kw)
    return FakeEngine(*args, **kw)


fake_factory.graph = fake_graph


def _otower():
    from oracle import tower
    return tower


class MetaSubset(object):
    """an oracle model whose flat vector covers the CHOSEN tensors only: `model_meta_parms` of maml.py:167-177 under the
    oracle's loops (get_flat = _get_meta_weights, set_flat = _set_model_meta_parms)."""

    def __init__(self, model, names):
        self.model, self.meta = model, list(names)

    def __getattr__(self, k):
        return getattr(self.model, k)

    def get_flat(self):
        return _otower().flatten(self.model.params, self.meta)

    def set_flat(self, vec):
        _otower().unflatten(vec, self.model.params, self.meta)
