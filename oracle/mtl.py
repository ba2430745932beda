"""fp32 numpy restatement of the reference's multi-task towers (test infrastructure).

Reference: model_zoo/DeepMTLCTR/deep_mtl_ctr.py:21-49 builds deepctr's `SharedBottom`, `MMOE` or `PLE` with one
binary task per domain and, per domain, a Keras model `Model(inputs, outputs[domain])` compiled on ONE shared
`tf.train.AdamOptimizer` (:53-67); `train()` (:69-96) fits domain d's model on domain d's data, so a step moves
only the variables on the path to output d (the tables, the experts that output mixes, gate d, tower d, head d).

PARITY UNPINNED: the layers live in deepctr==0.9.0 (requirements.txt:6; not in the tree, not installable) and the
reference holds no test or golden vector for them.  Their published algorithms (deepctr/models/multitask/
{sharedbottom,mmoe,ple}.py, deepctr/layers/core.py `DNN`, `PredictionLayer`) are restated here:

* input     x = [U[uid] | I[pid] | Dm[dom]]  (deep_mtl_ctr.py:98-106 == deepctr.py:95-116), l2 1e-5 on the three tables
* DNN       per hidden width: Dense(glorot_normal, zero bias) -> relu -> Dropout(rate), every layer incl. the last
* experts   SharedBottom: one bottom DNN(hidden_dim);  MMOE: `num_experts` DNNs on x;  PLE, num_levels = 1 (all
            reference configs): per task `specific_expert_num` DNNs + `shared_expert_num` shared ones, all on x
* gate d    MMOE / PLE: softmax(DNN(gate_dnn_hidden_units)(x) . Wg_d), Wg_d without bias, over the experts task d
            mixes (all of MMOE's; PLE: its own specific ones, then the shared ones); mix = sum_e gate_e * expert_e
* tower d   DNN(tower_hidden_dim) -> Dense(1, no bias) -> PredictionLayer('binary') = + global_bias_d, sigmoid
* loss      Keras binary_crossentropy + the embedding regularisers;  optimiser: TF1 Adam with ONE pair of beta
            powers for all domain models (the optimizer object is shared), per-variable slots, untouched variables
            neither move nor decay.

Gradients are derived by hand; tests/test_oracle_crosscheck.py holds them to float64 autograd of
oracle/torch_ref.mtl_forward.  Dropout masks / shuffles: the build's counter-based streams (oracle/rng.py).
"""
import numpy as np

from . import rng
from . import tower as T

F32 = np.float32


class Spec(object):
    """structure of one multi-task tower: expert DNNs, which experts each task mixes, gate / tower widths."""

    def __init__(self, kind, n_domain, expert_hidden, tower_hidden, gate_hidden=(), num_experts=1,
                 shared_expert_num=0, specific_expert_num=0, emb_dim=128):
        self.kind, self.D = kind, int(n_domain)
        self.expert_hidden, self.tower_hidden, self.gate_hidden = tuple(expert_hidden), tuple(tower_hidden), tuple(gate_hidden)
        self.xdim = 3 * emb_dim
        self.emb_dim = emb_dim
        if kind == "shared_bottom":
            self.shared_experts = ["bottom"]
            self.task_experts = [[] for _ in range(self.D)]
            self.gated = False
        elif kind == "mmoe":
            self.shared_experts = ["expert_%d" % e for e in range(num_experts)]
            self.task_experts = [[] for _ in range(self.D)]
            self.gated = True
        elif kind == "ple":
            self.shared_experts = ["shared_expert_%d" % e for e in range(shared_expert_num)]
            self.task_experts = [["task_%d_expert_%d" % (d, e) for e in range(specific_expert_num)] for d in range(self.D)]
            self.gated = True
        else:
            raise ValueError(kind)
        if not self.expert_hidden or not self.tower_hidden:
            raise ValueError("expert and tower DNNs need at least one hidden layer")

    def mix(self, d):
        """experts task d mixes, in the order deepctr stacks them (PLE: specific ones first, then the shared ones)."""
        return self.task_experts[d] + self.shared_experts

    def dnn_tensors(self, name, in_dim, hidden):
        out, i = [], in_dim
        for l, h in enumerate(hidden):
            out += [("%s/W%d" % (name, l), (i, h)), ("%s/b%d" % (name, l), (h,))]
            i = h
        return out

    def tensors(self, emb_trainable, n_user=0, n_item=0):
        """[(name, shape)] in flat-vector order: the block every task's model shares, then one block per task."""
        t = []
        if emb_trainable:
            t += [("user_emb", (n_user, self.emb_dim)), ("item_emb", (n_item, self.emb_dim))]
        t.append(("domain_emb", (self.D, self.emb_dim)))
        for e in self.shared_experts:
            t += self.dnn_tensors(e, self.xdim, self.expert_hidden)
        H = self.expert_hidden[-1]
        for d in range(self.D):
            for e in self.task_experts[d]:
                t += self.dnn_tensors(e, self.xdim, self.expert_hidden)
            if self.gated:
                t += self.dnn_tensors("gate_%d" % d, self.xdim, self.gate_hidden)
                t.append(("gate_%d/Wg" % d, (self.gate_hidden[-1] if self.gate_hidden else self.xdim, len(self.mix(d)))))
            t += self.dnn_tensors("tower_%d" % d, H, self.tower_hidden)
            t += [("head_%d/w" % d, (self.tower_hidden[-1], 1)), ("head_%d/gb" % d, (1,))]
        return t

    def task_tensors(self, d, emb_trainable):
        """names of the variables domain d's model trains (deep_mtl_ctr.py:59-66: Model(inputs, outputs[d]))."""
        pre = ("user_emb", "item_emb") if emb_trainable else ()
        keep = list(pre) + ["domain_emb"]
        owners = set(self.mix(d)) | {"gate_%d" % d, "tower_%d" % d, "head_%d" % d}
        return keep, owners


def init_params(rs, spec, n_user, n_item, pretrained=True):
    """injected initial tensors (TF initialiser streams are unreproducible): glorot-normal-like kernels, zero biases,
    N(0, 1e-4^2) domain table (deepctr SparseFeat default), N(0, 0.1^2) 'pretrained' / N(0, 1e-4^2) fresh tables."""
    p = {}
    sd = 0.1 if pretrained else 1e-4
    p["user_emb"] = (rs.standard_normal((n_user, spec.emb_dim)) * sd).astype(F32)
    p["item_emb"] = (rs.standard_normal((n_item, spec.emb_dim)) * sd).astype(F32)
    for name, shape in spec.tensors(False):
        if name == "domain_emb":
            p[name] = (rs.standard_normal(shape) * 1e-4).astype(F32)
        elif len(shape) == 2:
            s = np.sqrt(2.0 / (shape[0] + shape[1]))
            p[name] = (np.clip(rs.standard_normal(shape), -2, 2) * s).astype(F32)
        else:
            p[name] = np.zeros(shape, F32)
    return p


def layer_ids(spec):
    """dropout stream: every DNN layer draws its mask under its own id = its position among all kernels `/W<l>`."""
    ids, k = {}, 0
    for name, shape in spec.tensors(False):
        if "/W" in name and not name.endswith("/Wg"):
            ids[name] = k
            k += 1
    return ids


def dnn_forward(P, name, hidden, x, masks, keep_scale):
    """-> (output, [layer outputs after relu / dropout])."""
    h, outs = x, []
    for l in range(len(hidden)):
        z = (h @ P["%s/W%d" % (name, l)] + P["%s/b%d" % (name, l)]).astype(F32)
        a = np.maximum(z, F32(0))
        if masks is not None:
            a = (a * keep_scale * masks["%s/W%d" % (name, l)]).astype(F32)
        outs.append(a)
        h = a
    return h, outs


def dnn_backward(P, name, hidden, x, outs, dout, keep_scale, g):
    """dout = d loss / d (DNN output); fills g with the kernels' / biases' gradients; -> d loss / d x."""
    dh = dout
    for l in range(len(hidden) - 1, -1, -1):
        dz = (dh * ((outs[l] > 0).astype(F32) * keep_scale)).astype(F32)
        inp = x if l == 0 else outs[l - 1]
        g["%s/W%d" % (name, l)] = (inp.T @ dz).astype(F32)
        g["%s/b%d" % (name, l)] = np.sum(dz, axis=0, dtype=np.float64).astype(F32)
        dh = (dz @ P["%s/W%d" % (name, l)].T).astype(F32)
    return dh


def softmax_rows(z):
    z = (z - z.max(axis=1, keepdims=True)).astype(F32)
    e = np.exp(z, dtype=F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def forward(P, spec, d, uid, pid, dom, masks=None, keep_scale=F32(1)):
    """task d's output for the batch: -> (p, cache)."""
    x = T.gather(P, uid, pid, dom)
    c = {"x": x, "experts": {}}
    mix = spec.mix(d)
    for e in mix:
        c["experts"][e] = dnn_forward(P, e, spec.expert_hidden, x, masks, keep_scale)
    if spec.gated:
        q, qouts = dnn_forward(P, "gate_%d" % d, spec.gate_hidden, x, masks, keep_scale) if spec.gate_hidden else (x, [])
        gate = softmax_rows((q @ P["gate_%d/Wg" % d]).astype(F32))
        m = np.zeros_like(c["experts"][mix[0]][0])
        for k, e in enumerate(mix):
            m = (m + gate[:, k:k + 1] * c["experts"][e][0]).astype(F32)
        c.update(q=q, qouts=qouts, gate=gate)
    else:
        m = c["experts"][mix[0]][0]
    t, touts = dnn_forward(P, "tower_%d" % d, spec.tower_hidden, m, masks, keep_scale)
    logit = (t @ P["head_%d/w" % d]).astype(F32)[:, 0] + P["head_%d/gb" % d][0]
    c.update(m=m, t=t, touts=touts)
    return T.sigmoid(logit), c


def train_masks(spec, seed, step, n_rows, rate):
    ids = layer_ids(spec)
    shapes = dict(spec.tensors(False))
    return {n: rng.dropout_mask(seed, step, k, n_rows, shapes[n][1], rate) for n, k in ids.items()}


def loss_and_grads(P, spec, d, uid, pid, dom, label, masks, rate, emb_trainable, frozen_sumsq=None):
    """one batch of task d: total loss (mean BCE + regularisers) and the gradients of every tensor on the path."""
    B = uid.shape[0]
    keep_scale = F32(1.0 / (1.0 - rate)) if masks is not None else F32(1)
    p, c = forward(P, spec, d, uid, pid, dom, masks, keep_scale)
    y = label.astype(F32)
    loss = F32(np.mean(T.bce_per_row(p, y), dtype=np.float64)) + T.reg_loss(P, frozen_sumsq)
    inside = ((p >= T.EPS_CLIP) & (p <= F32(1) - T.EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    g = {}
    g["head_%d/w" % d] = (c["t"].T @ dlogit[:, None]).astype(F32)
    g["head_%d/gb" % d] = np.array([np.sum(dlogit, dtype=np.float64)], F32)
    dt = (dlogit[:, None] * P["head_%d/w" % d][:, 0][None, :]).astype(F32)
    dm = dnn_backward(P, "tower_%d" % d, spec.tower_hidden, c["m"], c["touts"], dt, keep_scale, g)
    mix = spec.mix(d)
    dx = np.zeros_like(c["x"])
    if spec.gated:
        gate = c["gate"]
        dgate = np.stack([np.sum((dm * c["experts"][e][0]).astype(np.float64), axis=1).astype(F32) for e in mix], axis=1)
        dgl = (gate * (dgate - np.sum((gate * dgate).astype(np.float64), axis=1, keepdims=True).astype(F32))).astype(F32)
        g["gate_%d/Wg" % d] = (c["q"].T @ dgl).astype(F32)
        dq = (dgl @ P["gate_%d/Wg" % d].T).astype(F32)
        if spec.gate_hidden:
            dx = (dx + dnn_backward(P, "gate_%d" % d, spec.gate_hidden, c["x"], c["qouts"], dq, keep_scale, g)).astype(F32)
        else:
            dx = (dx + dq).astype(F32)
        for k, e in enumerate(mix):
            de = (gate[:, k:k + 1] * dm).astype(F32)
            dx = (dx + dnn_backward(P, e, spec.expert_hidden, c["x"], c["experts"][e][1], de, keep_scale, g)).astype(F32)
    else:
        e = mix[0]
        dx = (dx + dnn_backward(P, e, spec.expert_hidden, c["x"], c["experts"][e][1], dm, keep_scale, g)).astype(F32)
    E = spec.emb_dim
    two_l2 = F32(2) * T.L2_EMB
    onehot = (dom[:, None] == np.arange(spec.D)[None, :]).astype(np.float64)
    g["domain_emb"] = ((onehot.T @ dx[:, 2 * E:3 * E].astype(np.float64)).astype(F32) + two_l2 * P["domain_emb"]).astype(F32)
    if emb_trainable:
        gu = np.zeros_like(P["user_emb"], dtype=np.float64)
        np.add.at(gu, uid, dx[:, 0:E].astype(np.float64))
        g["user_emb"] = (gu.astype(F32) + two_l2 * P["user_emb"]).astype(F32)
        gi = np.zeros_like(P["item_emb"], dtype=np.float64)
        np.add.at(gi, pid, dx[:, E:2 * E].astype(np.float64))
        g["item_emb"] = (gi.astype(F32) + two_l2 * P["item_emb"]).astype(F32)
    return loss, g, p


class OracleMTL(object):
    """stand-in for the D compiled Keras models over one set of variables and ONE Adam optimizer object."""

    def __init__(self, params, spec, emb_trainable=False, dropout=0.5, lr=1e-4, dropout_seed=1024):
        self.params, self.spec = params, spec
        self.emb_trainable = emb_trainable
        self.names = [n for n, _ in spec.tensors(emb_trainable, params["user_emb"].shape[0], params["item_emb"].shape[0])]
        self.m = {n: np.zeros_like(params[n]) for n in self.names}
        self.v = {n: np.zeros_like(params[n]) for n in self.names}
        self.b1p, self.b2p, self.t = F32(1), F32(1), 0
        self.rate, self.lr, self.seed = float(dropout), lr, dropout_seed
        self.step = 0
        self.use_sgd = False
        self._frozen = None

    def get_flat(self):
        return T.flatten(self.params, self.names)

    def set_flat(self, vec):
        T.unflatten(vec, self.params, self.names)

    def frozen_sumsq(self):
        if self.emb_trainable:
            return None
        key = (id(self.params["user_emb"]), id(self.params["item_emb"]))
        if self._frozen is None or self._frozen[0] != key:
            self._frozen = (key, {n: T.table_sumsq(self.params[n]) for n in ("user_emb", "item_emb")})
        return self._frozen[1]

    def train_on_batch(self, d, uid, pid, dom, label):
        B = uid.shape[0]
        masks = train_masks(self.spec, self.seed, self.step, B, self.rate) if self.rate > 0 else None
        loss, g, _ = loss_and_grads(self.params, self.spec, d, uid, pid, dom, label, masks, self.rate,
                                    self.emb_trainable, self.frozen_sumsq())
        if self.use_sgd:
            for n, gr in g.items():
                self.params[n] -= (gr * F32(self.lr)).astype(F32)
        else:
            # tf.train.AdamOptimizer (A.5): the beta powers belong to the optimizer object -> every step of ANY
            # domain model advances them; only the variables of this model have gradients and move
            self.t += 1
            self.b1p, self.b2p = F32(self.b1p * T.BETA1), F32(self.b2p * T.BETA2)
            alpha = F32(F32(self.lr) * np.sqrt(F32(1) - self.b2p, dtype=F32) / (F32(1) - self.b1p))
            omb1, omb2 = F32(F32(1) - T.BETA1), F32(F32(1) - T.BETA2)
            for n, gr in g.items():
                m, v = self.m[n], self.v[n]
                m += ((gr - m) * omb1).astype(F32)
                v += ((gr * gr - v) * omb2).astype(F32)
                self.params[n] -= ((m * alpha) / (np.sqrt(v, dtype=F32) + getattr(self, "adam_eps", T.ADAM_EPS))).astype(F32)
        self.step += 1
        return loss

    def train_pass(self, d, data, perm, batch_size, max_steps=0):
        n = perm.shape[0]
        n_step = -(-n // batch_size)
        if max_steps > 0:
            n_step = min(n_step, max_steps)
        losses = []
        for s in range(n_step):
            idx = perm[s * batch_size:(s + 1) * batch_size]
            losses.append(self.train_on_batch(d, data["uid"][idx], data["pid"][idx], data["domain"][idx], data["label"][idx]))
        return losses

    def evaluate(self, d, data, batch_size):
        """domain_model_dict[d].evaluate (deep_mtl_ctr.py:207): mean over batches of the batch-mean loss (+ reg)."""
        n = data["uid"].shape[0]
        reg = T.reg_loss(self.params, self.frozen_sumsq())
        batch_losses, preds = [], np.empty(n, F32)
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            p, _ = forward(self.params, self.spec, d, data["uid"][sl], data["pid"][sl], data["domain"][sl])
            preds[sl] = p
            batch_losses.append(F32(np.mean(T.bce_per_row(p, data["label"][sl].astype(F32)), dtype=np.float64)) + reg)
        return F32(np.mean(np.array(batch_losses, np.float64))), preds
