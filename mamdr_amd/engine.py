"""Device-resident tower: the compiled-Keras-model stand-in over libmamdr_hip.so.

`TowerEngine` owns (through torch, used purely as the device allocator / stream
provider) the flat trainable vector with its Adam slots, the frozen tables and
the per-domain split columns, and forwards every numeric operation to the C ABI
(include/mamdr_hip.h).  It exposes what the reference's wrappers use of the
Keras model:

    train_on_batch / fit        -> train_pass / train_steps   (mamdr.py:54,86,97)
    evaluate                    -> evaluate                   (base_model.py:131)
    K.batch_get_value(vars)     -> get_weights / weights      (maml.py:189-194)
    SetVarOp(vars)(values)      -> set_weights                (utils/tool.py:36-45)

plus the outer-update primitives on flat vectors.  No CPU fallback exists.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def auc_thresholds(num_thresholds=500):
    """utils/auc.py:118-126 (python doubles -> fp32 constant)."""
    inner = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
    return np.array([0.0 - 1e-7] + inner + [1.0 + 1e-7], dtype=np.float32)


def auc_from_histogram(hist):
    """AUC(num_thresholds=500) from the kernel's exact integer histogram.

    hist[c, k] = rows of class c whose prediction exceeds exactly k thresholds, so
    the confusion counts of utils/metrics_utils.py:297-354 are suffix sums:
    tp[t] = #{pos: pred > thr_t} = sum_{k > t} hist[1, k].  The Riemann sum follows
    utils/auc.py:248-281 in fp32 (counts are exact in fp32 below 2^24, as in the
    reference's fp32 accumulators).
    """
    hist = np.asarray(hist, dtype=np.int64).reshape(2, 501)
    suffix = np.cumsum(hist[:, ::-1], axis=1)[:, ::-1]      # suffix[c, k] = sum_{k' >= k}
    tp = suffix[1, 1:].astype(np.float32)                    # t = 0..499 -> k >= t+1
    fp = suffix[0, 1:].astype(np.float32)
    fn = (hist[1].sum() - suffix[1, 1:]).astype(np.float32)
    tn = (hist[0].sum() - suffix[0, 1:]).astype(np.float32)

    def div_no_nan(a, b):
        out = np.zeros_like(a)
        nz = b != 0
        out[nz] = a[nz] / b[nz]
        return out

    recall = div_no_nan(tp, tp + fn)
    fpr = div_no_nan(fp, fp + tn)
    heights = (recall[:-1] + recall[1:]) / np.float32(2.0)
    return float(np.sum((fpr[:-1] - fpr[1:]) * heights, dtype=np.float32)), (tp, fp, tn, fn)


class FlatVectorOps(object):
    """the outer (meta) updates on flat device vectors: stateless entry points of the library, shared by every engine
    (bit-exact vs the reference's numpy, see include/mamdr_hip.h).  Needs self.lib, self.stream, self.weights."""

    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    # outer updates (bit-exact vs the reference's numpy, see include/mamdr_hip.h)
    def interp(self, dst, a, b, scale):
        L.check(self.lib.mamdr_interp(_ptr(dst), _ptr(a), _ptr(b), float(scale), dst.numel(), self._s()))

    def merge(self, dst, theta, phi, method="plus"):
        mode = {"plus": L.MERGE_PLUS, "times": L.MERGE_TIMES}[method]
        L.check(self.lib.mamdr_merge(_ptr(dst), _ptr(theta), _ptr(phi), mode, dst.numel(), self._s()))

    # ---- the COMPILED optimiser (deepctr.py:54-60: `model.compile(optimizer=opt)`): the wrappers' train_on_batch / fit calls
    # -- train_steps(optimizer="adam") here -- run whatever the model was compiled with.  train.optimizer "adam" =
    # tf.train.AdamOptimizer(learning_rate); any other value is handed to Keras as a STRING, i.e. the Keras optimiser of
    # that name with ITS default hyper-parameters: "sgd" = SGD(lr 0.01, no momentum), whatever learning_rate says.
    compiled = ("adam", None)

    def compile(self, optimizer):
        if optimizer == "adam":
            self.compiled = ("adam", None)
        elif optimizer == "sgd":
            self.compiled = ("sgd", 0.01)
        else:
            raise NotImplementedError("train.optimizer '%s': Keras optimisers other than 'sgd' are not built (the reference's "
                                      "configs all use adam: deepctr.py:54-57)" % optimizer)

    def _compiled(self, optimizer, lr):
        """(optimizer, lr) a train_steps call runs: "adam" names the compiled optimiser."""
        if optimizer == "adam" and self.compiled[0] != "adam":
            return self.compiled
        return optimizer, lr

    # ---- meta parameters (MAML._get_model_meta_parms, maml.py:153-179): theta / phi vectors cover ONE contiguous range
    # [meta_off, meta_off + n_meta) of the flat vector -- the whole vector ("all"), the Star filter's prefix, or e.g.
    # "all_hidden" (everything behind the embedding tables)
    meta_off = 0

    # tensors INSIDE the meta range that are not meta parameters (a `meta_parms` name list that selects tensors which are
    # no neighbours in the flat vector, maml.py:167-177): [(offset within the range, count)].  theta / phi / merged span
    # the whole range; the slots of these tensors carry no meaning and `assign_meta` never lets them reach the model.
    meta_holes = ()

    def set_meta_range(self, off, count, holes=()):
        if off < 0 or count <= 0 or off + count > self.n_params:
            raise ValueError("meta range [%d, %d) outside the flat vector of %d floats" % (off, off + count, self.n_params))
        for o, c in holes:
            if o < 0 or c <= 0 or o + c > count:
                raise ValueError("hole [%d, %d) outside the meta range of %d floats" % (o, o + c, count))
        self.meta_off, self.n_meta = int(off), int(count)
        self.meta_holes = tuple((int(o), int(c)) for o, c in holes)

    def assign_meta(self, vec):
        """MAML._set_model_meta_parms (maml.py:181-187; SetVarOp over `model_meta_parms`): the meta parameters of the live
        model := vec.  Every other variable keeps training where it is -- the tensors between two selected ones first
        take their live values into `vec`, then the range is assigned in one copy."""
        if vec.numel() != self.n_meta:
            raise ValueError("assign_meta: %d values for %d meta parameters" % (vec.numel(), self.n_meta))
        live = self.meta_weights
        for o, c in self.meta_holes:
            L.check(self.lib.mamdr_copy(_ptr(vec[o:o + c]), _ptr(live[o:o + c]), c, self._s()))
        L.check(self.lib.mamdr_copy(_ptr(live), _ptr(vec), self.n_meta, self._s()))

    @property
    def meta_weights(self):
        """live values of the meta parameters (a view: no copy)."""
        return self.weights[self.meta_off:self.meta_off + self.n_meta]

    def dr_advance(self, phi, merged, theta, gamma, method="plus", assign_model=True):
        """one DR support step in a single pass: phi += (live - merged) * gamma; merged = theta (+|*) phi; and, for
        the next support, model := merged (mamdr.py:103-105,74) -- bit-identical to interp + merge + set_weights."""
        mode = {"plus": L.MERGE_PLUS, "times": L.MERGE_TIMES}[method]
        holes = bool(self.meta_holes) and assign_model      # (the kernel's assignment would overwrite the tensors in between)
        # self.weights (not _weights): the live table rows must be brought up to the current Adam step before they
        # are read into phi / replaced by merged (include/mamdr_hip.h: sync before reading or replacing the state)
        L.check(self.lib.mamdr_dr_advance(_ptr(phi), _ptr(self.meta_weights), _ptr(merged), _ptr(theta), float(gamma), mode,
                                          1 if (assign_model and not holes) else 0, phi.numel(), self._s()))
        if holes:
            self.assign_meta(merged)

    def sub(self, dst, a, b):
        L.check(self.lib.mamdr_sub(_ptr(dst), _ptr(a), _ptr(b), dst.numel(), self._s()))

    def accumulate(self, acc, a, b, shared=None, divisor=1.0):
        L.check(self.lib.mamdr_accumulate(_ptr(acc), _ptr(a), _ptr(b), _ptr(shared), float(divisor),
                                          acc.numel(), self._s()))

    def apply_accumulated(self, dst, acc, divisor, scale):
        L.check(self.lib.mamdr_apply_accumulated(_ptr(dst), _ptr(acc), float(divisor), float(scale),
                                                 dst.numel(), self._s()))


    def adam_apply(self, p, m, v, g, lr, beta1_power, beta2_power, grad_scale=1.0):
        """outer TF1 Adam on flat vectors (maml.py:236-243)."""
        L.check(self.lib.mamdr_adam_apply(_ptr(p), _ptr(m), _ptr(v), _ptr(g), float(grad_scale), float(lr), 0.9, 0.999,
                                          1e-8, float(beta1_power), float(beta2_power), p.numel(), self._s()))


    def pcgrad_project(self, final, aux, tensors=None):
        """PCGrad.PCGrad with final_grads is current_grads (pcgrad.py:107-124,152-160) on flat device vectors.
        tensors: [(offset, rows, cols)], default = every segment inside the vectors."""
        if tensors is None:
            shapes = self.segment_shapes()
            tensors = [(off, shapes[n][0], shapes[n][1]) for n, (off, cnt) in self.segments.items()
                       if off + cnt <= final.numel()]
        n = len(tensors)
        offs = (C.c_int64 * n)(*[t[0] for t in tensors])
        rows = (C.c_int64 * n)(*[t[1] for t in tensors])
        cols = (C.c_int32 * n)(*[t[2] for t in tensors])
        L.check(self.lib.mamdr_pcgrad_project(_ptr(final), _ptr(aux), offs, rows, cols, n, self._s()))


class TowerEngine(FlatVectorOps):
    def __init__(self, n_user, n_item, n_domain, batch_size, dropout=0.5, emb_trainable=False,
                 tower="mlp", emb_dim=128, hidden=(256, 128, 64), l2_emb=1e-5, device=None,
                 dropout_seed=1024, l2_linear=1e-5, uncertainty_weight=False, tower_tile=None):
        """tower_tile: rows per tower workgroup (mamdr_set_tower_tile: 0 automatic, 4, 16); None = automatic, except for an
        engine built on a lane of a parallel.LaneGroup of four or more lanes, which takes 16 (it shares the CUs with the
        other lanes' launches)."""
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("TowerEngine needs a HIP device (no CPU fallback for the MAMDR hot path)")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        self.n_user, self.n_item, self.n_domain = int(n_user), int(n_item), int(n_domain)
        self.batch_size = int(batch_size)
        self.dropout_seed = int(dropout_seed) & 0xFFFFFFFF
        self._acc = None
        self._ema = None            # set_moving_average
        tower_id = {"mlp": L.TOWER_MLP, "deepfm": L.TOWER_DEEPFM, "star": L.TOWER_STAR, "wdl": L.TOWER_WDL,
                    "pnn": L.TOWER_PNN, "nfm": L.TOWER_NFM}[tower]
        max_batch = (self.batch_size + 15) // 16 * 16
        cfg = L.Config(L.ABI_VERSION, tower_id, self.n_user, self.n_item, self.n_domain, emb_dim,
                       (C.c_int32 * 3)(*hidden), max_batch, 1 if emb_trainable else 0, float(dropout),
                       float(l2_emb), float(l2_linear), 0.9, 0.999, 1e-8, 1 if uncertainty_weight else 0)
        self.eval_batch = max_batch
        handle = C.c_void_p()
        L.check(self.lib.mamdr_create(C.byref(cfg), C.c_void_p(self.stream.cuda_stream), C.byref(handle)))
        self.ctx = handle
        if tower_tile is None:
            from . import parallel
            group = parallel.lanes()
            # (the PNN / NFM modes exist in the four-row tower only)
            tower_tile = 16 if (group is not None and group.n >= 4 and tower in ("mlp", "deepfm", "wdl")) else 0
        if tower_tile:
            self.set_tower_tile(tower_tile)
        self.emb_trainable = bool(emb_trainable)
        self.tower = tower
        self.n_params = int(self.lib.mamdr_param_count(self.ctx))
        # theta / phi vectors cover the META prefix only (all of it except for the Star tower)
        self.n_meta = int(self.lib.mamdr_meta_count(self.ctx))
        self.segments = {}
        for seg, name in enumerate(L.SEG_NAMES):
            off, cnt = C.c_int64(), C.c_int64()
            L.check(self.lib.mamdr_param_segment(self.ctx, seg, C.byref(off), C.byref(cnt)))
            if cnt.value:
                self.segments[name] = (off.value, cnt.value)
        # live state: weights + Adam slots (one set for the whole run)
        self._weights = self.new_vector()
        self._adam_m = self.new_vector()
        self._adam_v = self.new_vector()
        L.check(self.lib.mamdr_bind_state(self.ctx, _ptr(self._weights), _ptr(self._adam_m), _ptr(self._adam_v)))
        # non-trainable model state (Star: PartitionedNorm moving statistics, initial mean 0 / variance 1)
        self.aux = None
        n_aux = int(self.lib.mamdr_aux_count(self.ctx))
        if n_aux:
            self.aux = torch.zeros(n_aux, dtype=torch.float32, device=self.device)
            dx = self.n_domain * 3 * emb_dim
            self.aux[dx:2 * dx] = 1.0
            L.check(self.lib.mamdr_bind_aux(self.ctx, _ptr(self.aux)))
        self.tables = {}
        self.data = {}          # (domain, split) -> dict of device columns
        self._hist = torch.zeros(2 * 501, dtype=torch.int32, device=self.device)
        self._loss1 = torch.zeros(1, dtype=torch.float32, device=self.device)

    # The live state is only handed out synchronised: with trainable tables the library advances rows that
    # no batch touched lazily (mamdr_sync_tables in include/mamdr_hip.h); every read or replacement of the
    # bound vectors from the host side goes through these properties.
    def sync(self):
        L.check(self.lib.mamdr_sync_tables(self.ctx))

    @property
    def weights(self):
        self.sync()
        return self._weights

    @property
    def adam_m(self):
        self.sync()
        return self._adam_m

    @property
    def adam_v(self):
        self.sync()
        return self._adam_v

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize(self.device)
            self.lib.mamdr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ flat vectors
    def new_vector(self, like=None, meta=False):
        if like is not None:
            return like.clone()
        return torch.zeros(self.n_meta if meta else self.n_params, dtype=torch.float32, device=self.device)

    # Keras variable names of the segments, for the reference's substring filters (maml.py:153-179)
    KERAS_NAMES = {"Ws0": "kernel_shared_0", "Ws1": "kernel_shared_1", "Ws2": "kernel_shared_2",
                   "bs0": "bias_shared_0", "bs1": "bias_shared_1", "bs2": "bias_shared_2",
                   "Wd0": "kernel_specific_0", "Wd1": "kernel_specific_1", "Wd2": "kernel_specific_2",
                   "bd0": "bias_specific_0", "bd1": "bias_specific_1", "bd2": "bias_specific_2",
                   "pn_gamma_shared": "gamma_shared", "pn_beta_shared": "beta_shared",
                   "pn_gamma_spec": "gamma_specific", "pn_beta_spec": "beta_specific",
                   "wo": "dense/kernel", "gb": "dense/bias",
                   # deepctr's embedding layers: "sparse_emb_<feature>", the 1-d linear ones "linear...sparse_emb_<feature>"
                   "user_emb": "sparse_emb_uid", "item_emb": "sparse_emb_pid", "domain_emb": "sparse_emb_domain",
                   "lin_user": "linear0sparse_emb_uid", "lin_item": "linear0sparse_emb_pid",
                   "lin_domain": "linear0sparse_emb_domain"}

    def keras_name(self, segment):
        return self.KERAS_NAMES.get(segment, segment)

    def aux_state(self):
        """Star: {mov_mean, mov_var [D,384], steps [D]} as numpy (partitioned_norm.py:71-87)."""
        if self.aux is None:
            return {}
        h = self.aux.cpu().numpy()
        dx = self.n_domain * 384
        return {"mov_mean": h[0:dx].reshape(self.n_domain, 384).copy(),
                "mov_var": h[dx:2 * dx].reshape(self.n_domain, 384).copy(),
                "biased_mean": h[2 * dx:3 * dx].reshape(self.n_domain, 384).copy(),
                "biased_var": h[3 * dx:4 * dx].reshape(self.n_domain, 384).copy(),
                "steps": h[4 * dx:4 * dx + self.n_domain].copy()}

    def pack(self, named):
        """numpy dict {segment name: array} -> flat device vector (padding zero).  PNN: deepctr's first kernel
        [387, 256] is given as ONE tensor "W0"; its last three rows (the inner products') live in segment "W0x"."""
        host = np.zeros(self.n_params, np.float32)
        if "W0x" in self.segments and "W0x" not in named:
            w0 = np.asarray(named["W0"], np.float32).reshape(-1, 256)
            named = dict(named, W0=w0[:384], W0x=w0[384:])
        for name, (off, cnt) in self.segments.items():
            a = np.asarray(named[name], np.float32).ravel()
            if a.size != cnt:
                raise ValueError("segment %s has %d elements, expected %d" % (name, a.size, cnt))
            host[off:off + cnt] = a
        return torch.from_numpy(host).to(self.device)

    def unpack(self, vec):
        host = vec.detach().cpu().numpy()
        out = {name: host[off:off + cnt].copy() for name, (off, cnt) in self.segments.items()}
        if "W0x" in out:           # PNN: the caller's view is deepctr's one kernel [387, 256]
            out["W0"] = np.concatenate([out["W0"], out.pop("W0x")])
        return out

    def set_weights(self, vec):
        """SetVarOp.__call__ (utils/tool.py:36-45): device copy into the live weights.  A vector of
        meta length assigns the meta prefix only (MAML._set_model_meta_parms, maml.py:181-187)."""
        dst = self.meta_weights if (self.meta_off and vec.numel() == self.n_meta) else self.weights
        L.check(self.lib.mamdr_copy(_ptr(dst), _ptr(vec), vec.numel(), self._s()))

    def get_weights(self, out=None):
        """K.batch_get_value (maml.py:189-194): snapshot of the live weights."""
        if out is None:
            out = torch.empty_like(self.weights)
        L.check(self.lib.mamdr_copy(_ptr(out), _ptr(self.weights), self.n_params, self._s()))
        return out

    def segment_shapes(self):
        """{segment: (slices along the last axis, slice length)} of the Keras variables behind the segments --
        what numpy's axis=-1 reductions in the reference's PCGrad see (model_zoo/pcgrad.py:152-160)."""
        D = self.n_domain
        out = {}
        for name, (off, cnt) in self.segments.items():
            if name in ("user_emb", "item_emb", "domain_emb"):
                out[name] = (cnt // 128, 128)
            elif name in ("W0", "Ws0"):
                out[name] = (cnt // 256, 256)          # (NFM's first kernel has 128 rows)
            elif name == "W0x":
                out[name] = (3, 256)
            elif name in ("W1", "Ws1"):
                out[name] = (256, 128)
            elif name in ("W2", "Ws2"):
                out[name] = (128, 64)
            elif name in ("Wd0", "Wd1", "Wd2"):
                cols = {"Wd0": 256, "Wd1": 128, "Wd2": 64}[name]
                out[name] = (cnt // cols, cols)
            elif name in ("bd0", "bd1", "bd2", "pn_gamma_spec", "pn_beta_spec"):
                out[name] = (D, cnt // D)
            elif name in ("wo", "lin_user", "lin_item", "lin_domain", "log_var", "gb"):
                out[name] = (cnt, 1)
            else:                                  # 1-d biases and PartitionedNorm shared vectors
                out[name] = (1, cnt)
        return out

    def set_tower_tile(self, rows):
        L.check(self.lib.mamdr_set_tower_tile(self.ctx, int(rows)))

    def tower_tile(self, batch=None):
        """rows per tower workgroup of a training step of `batch` rows (4: k_tower4, 16: k_tower)."""
        rows = int(self.lib.mamdr_tower_tile(self.ctx, int(batch or self.batch_size)))
        if rows < 0:                    # (an error code is not a tile size: ADVICE r05)
            L.check(rows)
        return rows

    # ------------------------------------------------------------ binding
    def bind_table(self, name, rows):
        """frozen pretrained table (deepctr.py:104-116), numpy [n, 128] fp32."""
        seg = {"user_emb": L.SEG_USER_EMB, "item_emb": L.SEG_ITEM_EMB}[name]
        t = torch.from_numpy(np.ascontiguousarray(rows, np.float32)).to(self.device)
        self.tables[name] = t
        L.check(self.lib.mamdr_bind_table(self.ctx, seg, _ptr(t), t.shape[0]))

    def bind_domain_data(self, domain, split, uid, pid, dom, label):
        split_id = {"train": L.SPLIT_TRAIN, "val": L.SPLIT_VAL, "test": L.SPLIT_TEST}[split]
        uid = np.ascontiguousarray(uid, np.int32)
        pid = np.ascontiguousarray(pid, np.int32)
        dom = np.ascontiguousarray(dom, np.int32)
        if uid.size:
            if uid.min() < 0 or uid.max() >= self.n_user or pid.min() < 0 or pid.max() >= self.n_item \
                    or dom.min() < 0 or dom.max() >= self.n_domain:
                raise ValueError("domain %d %s: id out of range" % (domain, split))
        cols = {
            "uid": torch.from_numpy(uid).to(self.device),
            "pid": torch.from_numpy(pid).to(self.device),
            "domain": torch.from_numpy(dom).to(self.device),
            "label": torch.from_numpy(np.ascontiguousarray(label, np.float32)).to(self.device),
        }
        self.data[(domain, split)] = cols
        L.check(self.lib.mamdr_bind_domain_data(self.ctx, domain, split_id, _ptr(cols["uid"]), _ptr(cols["pid"]),
                                                _ptr(cols["domain"]), _ptr(cols["label"]), uid.shape[0]))

    def n_rows(self, domain, split):
        return int(self.data[(domain, split)]["uid"].shape[0])

    # ------------------------------------------------------------ the hot path
    def train_steps(self, domain, perm=None, first_step=0, n_steps=None, lr=1e-3, optimizer="adam",
                    loss_out=None, batch_size=None, pass_rows=None):
        """n_steps x train_on_batch on one domain, device-side. perm: int32 device tensor or None.
        pass_rows: the pass covers only that many positions (perm then lists that many rows of the split) --
        the take / skip sub-datasets of the meta-train / meta-val split."""
        bs = batch_size or self.batch_size
        n = self.n_rows(domain, "train") if pass_rows is None else int(pass_rows)
        if n_steps is None:
            n_steps = -(-n // bs) - first_step
        optimizer, lr = self._compiled(optimizer, lr)
        opt = {"adam": L.OPT_ADAM, "sgd": L.OPT_SGD, "accumulate": L.OPT_ACCUMULATE}[optimizer]
        if optimizer == "accumulate" and self._ema is not None:
            # average_meta_grad == "moving_mean": every meta batch updates the accumulator's moving average
            # (maml.py:219-220) -- one step at a time into a scratch gradient, then mamdr_moving_average
            ema = self._ema
            if loss_out is not None:
                raise ValueError("accumulate passes under average_meta_grad = moving_mean report no per-step loss")
            for s in range(first_step, first_step + n_steps):
                ema["scratch"].zero_()
                L.check(self.lib.mamdr_train_steps_n(self.ctx, domain, _ptr(perm), -1 if pass_rows is None else n,
                                                     s, 1, bs, self.dropout_seed, opt, float(lr), _ptr(None)))
                ema["step"] += 1
                decay = np.float32(1.0 - ema["momentum"])
                denom = np.float32(1.0) - np.power(np.float32(1.0) - decay, np.float32(ema["step"]), dtype=np.float32)
                L.check(self.lib.mamdr_moving_average(_ptr(self._acc), _ptr(ema["biased"]), _ptr(ema["scratch"]),
                                                      float(decay), float(denom), self._acc.numel(), self._s()))
            return n_steps
        L.check(self.lib.mamdr_train_steps_n(self.ctx, domain, _ptr(perm), -1 if pass_rows is None else n,
                                             first_step, n_steps, bs, self.dropout_seed, opt, float(lr),
                                             _ptr(loss_out)))
        return n_steps

    def dr_advance(self, phi, merged, theta, gamma, method="plus", assign_model=True):
        """FlatVectorOps.dr_advance on the context's live weights (mamdr_dr_advance_live): the library synchronises them
        itself, and a domain-table step the fused step path left pending is materialised inside the same launch."""
        mode = {"plus": L.MERGE_PLUS, "times": L.MERGE_TIMES}[method]
        holes = bool(self.meta_holes) and assign_model
        L.check(self.lib.mamdr_dr_advance_live(self.ctx, _ptr(phi), _ptr(merged), _ptr(theta), float(gamma), mode,
                                               1 if (assign_model and not holes) else 0, self.meta_off, phi.numel()))
        if holes:
            self.assign_meta(merged)

    def pregather(self, passes, batch_size=None):
        """hint (mamdr_pregather_passes): the next train_steps calls run these passes -- [(domain, perm device tensor or
        None[, pass_rows])] -- in this order; where a call would gather its pass's rows itself (frozen tables, fused
        step path) the library gathers them all in one launch now.  Same rows, same bits; a no-op elsewhere."""
        n = len(passes)
        if n == 0:                 # forget an earlier hint
            L.check(self.lib.mamdr_pregather_passes(self.ctx, 0, None, None, None, int(batch_size or self.batch_size)))
            return
        doms = (C.c_int32 * n)(*[int(p[0]) for p in passes])
        perms = (C.c_void_p * n)(*[(p[1].data_ptr() if p[1] is not None else None) for p in passes])
        rows = (C.c_int64 * n)(*[(-1 if len(p) < 3 or p[2] is None else int(p[2])) for p in passes])
        L.check(self.lib.mamdr_pregather_passes(self.ctx, n, doms, perms, rows, int(batch_size or self.batch_size)))

    def evaluate(self, domain, split, want_preds=False):
        """model.evaluate(data, steps=n_step) -> (loss, auc[, preds]); syncs to read back."""
        n = self.n_rows(domain, split)
        preds = torch.empty(n, dtype=torch.float32, device=self.device) if want_preds else None
        split_id = {"train": L.SPLIT_TRAIN, "val": L.SPLIT_VAL, "test": L.SPLIT_TEST}[split]
        L.check(self.lib.mamdr_eval_domain(self.ctx, domain, split_id, self.eval_batch, _ptr(self._loss1),
                                           _ptr(self._hist), _ptr(preds)))
        hist = self._hist.cpu().numpy().astype(np.int64)
        loss = float(self._loss1.cpu().numpy()[0])
        auc, _ = auc_from_histogram(hist)
        if want_preds:
            return loss, auc, hist.reshape(2, 501), preds.cpu().numpy()
        return loss, auc

    def gather(self, domain, split, perm=None, first_row=0, n_rows=None, out=None):
        n = self.n_rows(domain, split) if n_rows is None else n_rows
        if out is None:
            out = torch.empty((n, 384), dtype=torch.float32, device=self.device)
        split_id = {"train": L.SPLIT_TRAIN, "val": L.SPLIT_VAL, "test": L.SPLIT_TEST}[split]
        L.check(self.lib.mamdr_gather_rows(self.ctx, domain, split_id, _ptr(perm), first_row, n, _ptr(out)))
        return out

    def bind_accumulator(self, acc):
        """meta-gradient accumulator of the MAML meta pass (maml.py:202); optimizer="accumulate" adds to it
        (or, after set_moving_average, moves it towards every batch's gradient)."""
        self._acc = acc
        L.check(self.lib.mamdr_bind_accumulator(self.ctx, _ptr(acc if self._ema is None else self._ema["scratch"])))

    def set_moving_average(self, momentum):
        """average_meta_grad == "moving_mean" (maml.py:219-220): accumulate passes keep TF 1.12's zero-debiased
        moving average of the batch gradients in the bound accumulator.  Its hidden state (`biased`, `local_step`)
        lives as long as the engine, like the hidden variables of the reference's accumulator."""
        self._ema = {"momentum": float(momentum), "step": 0, "biased": self.new_vector(), "scratch": self.new_vector()}
        if self._acc is not None:
            self.bind_accumulator(self._acc)

    def set_counters(self, optimizer_steps, dropout_steps):
        """restore the Adam step count (with TF's running beta powers) and the dropout stream's position
        (mamdr_set_counters): a run resumed from saved weights / slots written into the bound vectors."""
        L.check(self.lib.mamdr_set_counters(self.ctx, int(optimizer_steps), int(dropout_steps)))

    def optimizer_reset(self):
        L.check(self.lib.mamdr_optimizer_reset(self.ctx))

    def step_kernel_names(self, batch=None):
        """names of the kernels behind profile_read's FWD_BWD / WGRAD / UPDATE slots for a step of `batch` rows."""
        fused = int(self.lib.mamdr_step_path(self.ctx, int(batch or self.batch_size))) == 1
        return {L.KERNEL_FWD_BWD: "k_tower<train>", L.KERNEL_WGRAD: "k_wgrad_adam" if fused else "k_wgrad",
                L.KERNEL_UPDATE: "k_dm_finish (when the live table is read)" if fused else "k_update"}

    # ------------------------------------------------------------ profiling
    def profile(self, enable):
        L.check(self.lib.mamdr_profile_enable(self.ctx, 1 if enable else 0))

    def profile_reset(self):
        L.check(self.lib.mamdr_profile_reset(self.ctx))

    def profile_read(self, kernel):
        ms, cnt = C.c_double(), C.c_int64()
        L.check(self.lib.mamdr_profile_read(self.ctx, kernel, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


def shuffle_perm(n, buffer_size, seed):
    """host: tf.data shuffle-buffer order (utils/dataset.py:27-37) via the C ABI."""
    out = np.empty(n, np.int32)
    L.check(L.load().mamdr_shuffle_perm(n, buffer_size, seed & 0xFFFFFFFFFFFFFFFF, out.ctypes.data))
    return out
