"""Device-resident multi-task tower (shared_bottom / mmoe / ple): the stand-in for the D compiled Keras models of
model_zoo/DeepMTLCTR/deep_mtl_ctr.py:51-67 over the `mamdr_graph_*` entry points of libmamdr_hip.so.

Same surface as `TowerEngine` where the reference's DeepMTLCTR uses the Keras models:

    domain_model_dict[d].fit(...)       -> train_steps(d, ...)      (deep_mtl_ctr.py:79-80,166-172)
    domain_model_dict[d].evaluate(...)  -> evaluate(d, split)       (deep_mtl_ctr.py:207)
    model.get_weights / set_weights     -> get_weights / set_weights (deep_mtl_ctr.py:146,158,192)

torch is the device allocator / stream provider only.  No CPU fallback exists.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import FlatVectorOps, _ptr, auc_from_histogram

KINDS = {"shared_bottom": L.GRAPH_SHARED_BOTTOM, "mmoe": L.GRAPH_MMOE, "ple": L.GRAPH_PLE, "nfm": L.GRAPH_NFM, "pnn": L.GRAPH_PNN,
         "ccpm": L.GRAPH_CCPM, "autoint": L.GRAPH_AUTOINT, "mlp": L.GRAPH_MLP, "wdl": L.GRAPH_WDL, "deepfm": L.GRAPH_DEEPFM}


def _arr4(values):
    v = list(values) + [0] * (4 - len(values))
    return (C.c_int32 * 4)(*v)


class GraphEngine(FlatVectorOps):
    def __init__(self, kind, n_user, n_item, n_domain, batch_size, expert_hidden, tower_hidden, gate_hidden=(),
                 num_experts=0, shared_expert_num=0, specific_expert_num=0, dropout=0.5, emb_trainable=False, emb_dim=128,
                 l2_emb=1e-5, device=None, dropout_seed=1024, l2_linear=1e-5, uncertainty_weight=False):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GraphEngine needs a HIP device (no CPU fallback)")
        if len(expert_hidden) > 4 or len(tower_hidden) > 4 or len(gate_hidden) > 4:
            raise ValueError("at most 4 hidden layers per DNN")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        self.kind = kind
        self.n_user, self.n_item, self.n_domain = int(n_user), int(n_item), int(n_domain)
        self.batch_size = int(batch_size)
        self.dropout_seed = int(dropout_seed) & 0xFFFFFFFF
        self.emb_trainable = bool(emb_trainable)
        max_batch = (self.batch_size + 63) // 64 * 64
        self.eval_batch = self.batch_size
        cfg = L.GraphConfig(L.ABI_VERSION, KINDS[kind], self.n_user, self.n_item, self.n_domain, emb_dim, max_batch,
                            1 if emb_trainable else 0, len(expert_hidden), _arr4(expert_hidden), len(tower_hidden),
                            _arr4(tower_hidden), len(gate_hidden), _arr4(gate_hidden), int(num_experts),
                            int(shared_expert_num), int(specific_expert_num), float(dropout), float(l2_emb), 0.9, 0.999, 1e-8,
                            float(l2_linear), 1 if uncertainty_weight else 0)
        handle = C.c_void_p()
        L.check(self.lib.mamdr_graph_create(C.byref(cfg), C.c_void_p(self.stream.cuda_stream), C.byref(handle)), graph=True)
        self.ctx = handle
        self.n_params = int(self.lib.mamdr_graph_param_count(self.ctx))
        self.n_meta = self.n_params
        self.segments, self.shapes = {}, {}
        buf = C.create_string_buffer(128)
        for i in range(int(self.lib.mamdr_graph_tensor_count(self.ctx))):
            off, rows, cols = C.c_int64(), C.c_int64(), C.c_int64()
            L.check(self.lib.mamdr_graph_tensor_info(self.ctx, i, buf, 128, C.byref(off), C.byref(rows), C.byref(cols)), graph=True)
            name = buf.value.decode()
            self.segments[name] = (off.value, rows.value * cols.value)
            self.shapes[name] = (rows.value, cols.value)
        self._weights = self.new_vector()
        self._adam_m = self.new_vector()
        self._adam_v = self.new_vector()
        L.check(self.lib.mamdr_graph_bind_state(self.ctx, _ptr(self._weights), _ptr(self._adam_m), _ptr(self._adam_v)), graph=True)
        self.aux = None
        self.tables, self.data = {}, {}
        self._acc, self._ema = None, None
        self._hist = torch.zeros(2 * 501, dtype=torch.int32, device=self.device)
        self._loss1 = torch.zeros(1, dtype=torch.float32, device=self.device)

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize(self.device)
            self.lib.mamdr_graph_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ flat vectors
    @property
    def weights(self):
        return self._weights

    @property
    def adam_m(self):
        return self._adam_m

    @property
    def adam_v(self):
        return self._adam_v

    def new_vector(self, like=None, meta=False):
        if like is not None:
            return like.clone()
        return torch.zeros(self.n_meta if meta else self.n_params, dtype=torch.float32, device=self.device)

    def keras_name(self, segment):
        return segment

    def pack(self, named):
        host = np.zeros(self.n_params, np.float32)
        for name, (off, cnt) in self.segments.items():
            a = np.asarray(named[name], np.float32).ravel()
            if a.size != cnt:
                raise ValueError("tensor %s has %d elements, expected %d" % (name, a.size, cnt))
            host[off:off + cnt] = a
        return torch.from_numpy(host).to(self.device)

    def unpack(self, vec):
        host = vec.detach().cpu().numpy()
        return {name: host[off:off + cnt].copy() for name, (off, cnt) in self.segments.items()}

    def set_weights(self, vec):
        dst = self.meta_weights if (self.meta_off and vec.numel() == self.n_meta) else self._weights[:vec.numel()]
        dst.copy_(vec)

    def get_weights(self, out=None):
        if out is None:
            return self._weights.clone()
        out.copy_(self._weights)
        return out

    def segment_shapes(self):
        """{tensor: (slices along the last axis, slice length)} of the Keras variables behind the tensors -- what numpy's
        axis=-1 reductions in the reference's PCGrad see (model_zoo/pcgrad.py:152-160): rows of a kernel / table, a bias
        as one slice, the Dense(1) head kernels and deepctr's 1-d linear tables ([n, 1]) as n slices of one element."""
        out = {}
        for name, (rows, cols) in self.shapes.items():
            cnt = rows * cols
            if name.startswith("lin_") or name.endswith("/w") or name in ("wo", "gb") or name.endswith("/gb"):
                out[name] = (cnt, 1)
            else:
                out[name] = (rows, cols)
        return out

    def task_ranges(self, domain):
        """[(offset, count)] of the flat vector a step on `domain` trains (Model(inputs, outputs[domain]).trainable_weights)."""
        v = [C.c_int64() for _ in range(4)]
        L.check(self.lib.mamdr_graph_task_ranges(self.ctx, int(domain), *[C.byref(x) for x in v]), graph=True)
        return [(v[0].value, v[1].value), (v[2].value, v[3].value)]

    # ------------------------------------------------------------ binding
    def bind_table(self, name, rows):
        seg = {"user_emb": L.SEG_USER_EMB, "item_emb": L.SEG_ITEM_EMB}[name]
        t = torch.from_numpy(np.ascontiguousarray(rows, np.float32)).to(self.device)
        self.tables[name] = t
        L.check(self.lib.mamdr_graph_bind_table(self.ctx, seg, _ptr(t), t.shape[0]), graph=True)

    def bind_domain_data(self, domain, split, uid, pid, dom, label):
        split_id = {"train": L.SPLIT_TRAIN, "val": L.SPLIT_VAL, "test": L.SPLIT_TEST}[split]
        uid = np.ascontiguousarray(uid, np.int32)
        pid = np.ascontiguousarray(pid, np.int32)
        dom = np.ascontiguousarray(dom, np.int32)
        if uid.size and (uid.min() < 0 or uid.max() >= self.n_user or pid.min() < 0 or pid.max() >= self.n_item
                         or dom.min() < 0 or dom.max() >= self.n_domain):
            raise ValueError("domain %d %s: id out of range" % (domain, split))
        cols = {"uid": torch.from_numpy(uid).to(self.device), "pid": torch.from_numpy(pid).to(self.device),
                "domain": torch.from_numpy(dom).to(self.device),
                "label": torch.from_numpy(np.ascontiguousarray(label, np.float32)).to(self.device)}
        self.data[(domain, split)] = cols
        L.check(self.lib.mamdr_graph_bind_domain_data(self.ctx, domain, split_id, _ptr(cols["uid"]), _ptr(cols["pid"]),
                                                      _ptr(cols["domain"]), _ptr(cols["label"]), uid.shape[0]), graph=True)

    def n_rows(self, domain, split):
        return int(self.data[(domain, split)]["uid"].shape[0])

    # ------------------------------------------------------------ steps / evaluation
    def train_steps(self, domain, perm=None, first_step=0, n_steps=None, lr=1e-3, optimizer="adam", loss_out=None,
                    batch_size=None, pass_rows=None):
        """as TowerEngine.train_steps (same optimiser names, windows and moving-average accumulate passes)."""
        bs = batch_size or self.batch_size
        n = self.n_rows(domain, "train") if pass_rows is None else int(pass_rows)
        if n_steps is None:
            n_steps = -(-n // bs) - first_step
        optimizer, lr = self._compiled(optimizer, lr)
        opt = {"adam": L.OPT_ADAM, "sgd": L.OPT_SGD, "accumulate": L.OPT_ACCUMULATE}[optimizer]
        rows = -1 if pass_rows is None else n
        if optimizer == "accumulate" and self._ema is not None:       # average_meta_grad == "moving_mean" (maml.py:219-220)
            ema = self._ema
            if loss_out is not None:
                raise ValueError("accumulate passes under average_meta_grad = moving_mean report no per-step loss")
            for s in range(first_step, first_step + n_steps):
                ema["scratch"].zero_()
                L.check(self.lib.mamdr_graph_train_steps_n(self.ctx, domain, _ptr(perm), rows, s, 1, bs, self.dropout_seed, opt,
                                                           float(lr), _ptr(None)), graph=True)
                ema["step"] += 1
                decay = np.float32(1.0 - ema["momentum"])
                denom = np.float32(1.0) - np.power(np.float32(1.0) - decay, np.float32(ema["step"]), dtype=np.float32)
                L.check(self.lib.mamdr_moving_average(_ptr(self._acc), _ptr(ema["biased"]), _ptr(ema["scratch"]),
                                                      float(decay), float(denom), self._acc.numel(), self._s()))
            return n_steps
        L.check(self.lib.mamdr_graph_train_steps_n(self.ctx, domain, _ptr(perm), rows, first_step, n_steps, bs,
                                                   self.dropout_seed, opt, float(lr), _ptr(loss_out)), graph=True)
        return n_steps

    def bind_accumulator(self, acc):
        """meta-gradient accumulator of the MAML / MLDG / PCGrad meta passes (maml.py:202)."""
        self._acc = acc
        L.check(self.lib.mamdr_graph_bind_accumulator(self.ctx, _ptr(acc if self._ema is None else self._ema["scratch"])),
                graph=True)

    def set_moving_average(self, momentum):
        self._ema = {"momentum": float(momentum), "step": 0, "biased": self.new_vector(), "scratch": self.new_vector()}
        if self._acc is not None:
            self.bind_accumulator(self._acc)

    def evaluate(self, domain, split, want_preds=False):
        n = self.n_rows(domain, split)
        preds = torch.empty(n, dtype=torch.float32, device=self.device) if want_preds else None
        split_id = {"train": L.SPLIT_TRAIN, "val": L.SPLIT_VAL, "test": L.SPLIT_TEST}[split]
        L.check(self.lib.mamdr_graph_eval_domain(self.ctx, domain, split_id, self.eval_batch, _ptr(self._loss1), _ptr(self._hist),
                                                 _ptr(preds)), graph=True)
        hist = self._hist.cpu().numpy().astype(np.int64)
        loss = float(self._loss1.cpu().numpy()[0])
        auc, _ = auc_from_histogram(hist)
        if want_preds:
            return loss, auc, hist.reshape(2, 501), preds.cpu().numpy()
        return loss, auc

    def set_counters(self, optimizer_steps, dropout_steps):
        """mamdr_graph_set_counters: a run resumed from saved weights / slots written into the bound vectors."""
        L.check(self.lib.mamdr_graph_set_counters(self.ctx, int(optimizer_steps), int(dropout_steps)), graph=True)

    def optimizer_reset(self):
        L.check(self.lib.mamdr_graph_optimizer_reset(self.ctx), graph=True)

    def set_adam_eps(self, eps):
        L.check(self.lib.mamdr_graph_set_adam_eps(self.ctx, float(eps)), graph=True)
