"""torch-CPU restatement of the reference's inner step with AUTOGRAD gradients (test infrastructure).

Two uses, both outside the product path (only tests/ and bench.py's cpu_baseline leg import this file):

* an independent cross-check of oracle/tower.py and oracle/star.py: those derive every gradient by hand in
  numpy fp32; here only the FORWARD is written down (the same published
  tensorflow-gpu==1.12.0 / deepctr==0.9.0 algorithms, requirements.txt:1,6) and torch.autograd differentiates
  it in float64 (tests/test_oracle_crosscheck.py; the same file holds the mlp tower and the Adam update to scikit-learn's
  MLPClassifier._backprop / AdamOptimizer, a third-party implementation of both).  PARITY UNPINNED all the same: no side is TF.
* the CPU baseline of bench.py ("torch-CPU fp32, same step list, torch.set_num_threads(all cores)"):
  `TorchCpuModel.train_on_batch` = gather -> tower forward -> Keras BCE + regularisers -> autograd backward ->
  TF1 dense Adam over every trainable tensor, what `model.train_on_batch` of the compiled Keras model executes
  per step (model_zoo/mamdr.py:54,86,97; model_zoo/DeepCTR/deepctr.py:54-60,118-136).

Formulas cited next to each function; names of the tensors = oracle/tower.py / oracle/star.py.
"""
import numpy as np
import torch

EPS_CLIP = 1e-7      # K.epsilon(): Keras binary_crossentropy clips the probability to [eps, 1 - eps]   (A.4)
L2_EMB = 1e-5        # deepctr l2_reg_embedding, deepctr.py:119,125-126                                  (A.3)
L2_LIN = 1e-5        # deepctr DeepFM / WDL l2_reg_linear default                                        (A.8)
PN_EPS = 1e-3        # partitioned_norm.py:60 epsilon
BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8     # tf.train.AdamOptimizer defaults, deepctr.py:55            (A.5)


def keras_bce(p, y):
    """Keras binary_crossentropy on probabilities (A.4): clip, back to logits, sigmoid_cross_entropy_with_logits
    (max(z, 0) - z y + log1p(exp(-|z|))).  The clip's zero gradient outside [eps, 1 - eps] comes from autograd."""
    pc = torch.clamp(p, EPS_CLIP, 1.0 - EPS_CLIP)
    z = torch.log(pc / (1.0 - pc))
    return torch.clamp(z, min=0.0) - z * y + torch.log1p(torch.exp(-torch.abs(z)))


def _rows(table, ids):
    """embedding lookup; F.embedding's dense backward is the fast CPU path for the tables' gradients."""
    return torch.nn.functional.embedding(ids, table)


def deepctr_forward(P, uid, pid, dom, masks, keep_scale, tower):
    """deepctr.py:118-136 (A.1): concat(user, item, domain rows) -> DNN(256, 128, 64; relu; dropout) ->
    Dense(1, no bias) + global_bias -> sigmoid; DeepFM (A.8) adds the three 1-d linear tables and the FM
    second-order term over the three 128-d fields, WDL the linear tables only (deepctr.py:29-32)."""
    u, i, d = _rows(P["user_emb"], uid), _rows(P["item_emb"], pid), _rows(P["domain_emb"], dom)
    h = torch.cat([u, i, d], dim=1)
    l = 0
    while "W%d" % l in P:          # hidden_dim of any length (deepctr.py:26-49 passes it through as dnn_hidden_units)
        h = torch.relu(torch.addmm(P["b%d" % l], h, P["W%d" % l]))
        if masks is not None:
            h = h * keep_scale * masks[l]
        l += 1

    logit = (h @ P["wo"])[:, 0] + P["gb"][0]
    if tower in ("deepfm", "wdl"):
        logit = logit + P["lin_user"][uid] + P["lin_item"][pid] + P["lin_domain"][dom]
    if tower == "deepfm":
        s = u + i + d
        logit = logit + 0.5 * torch.sum(s * s - (u * u + i * i + d * d), dim=1)
    return torch.sigmoid(logit)


def deepctr_loss(P, uid, pid, dom, y, masks, keep_scale, tower, uncertainty=False, frozen_reg=None):
    """mean BCE + l2 * sum(W^2) over every embedding table, frozen or not (A.3) (+ the linear tables for DeepFM /
    WDL); uncertainty weighting (weighted_loss.py:30-43): mean(BCE) / var^2 + log var, var = log_var[dom[0]]."""
    p = deepctr_forward(P, uid, pid, dom, masks, keep_scale, tower)
    bce = keras_bce(p, y).mean()
    if frozen_reg is not None:      # frozen tables: their (constant) term computed once by the caller
        reg = frozen_reg + L2_EMB * P["domain_emb"].pow(2).sum()
    else:
        reg = L2_EMB * (P["user_emb"].pow(2).sum() + P["item_emb"].pow(2).sum() + P["domain_emb"].pow(2).sum())
    if tower in ("deepfm", "wdl"):
        reg = reg + L2_LIN * (P["lin_user"].pow(2).sum() + P["lin_item"].pow(2).sum() + P["lin_domain"].pow(2).sum())
    if uncertainty:
        var = P["log_var"][int(dom[0])]
        return bce / (var * var) + torch.log(var) + reg, p
    return bce + reg, p


def star_forward(P, state, uid, pid, dom, training):
    """Star tower (A.7): d = domain of the first row (partitioned_norm.py:136, star_fcn.py:112);
    PartitionedNorm with batch statistics (population variance) in training, domain d's moving statistics in
    inference, gamma = gamma_shared * gamma_specific[d], beta = beta_shared + beta_specific[d], eps 1e-3
    (partitioned_norm.py:102-110,143-174); StarFCN kernel = shared * specific[d], bias = shared + specific[d]
    (star_fcn.py:105-139); Dense(1, sigmoid) with bias (star.py:95); no dropout, no regularisers."""
    d = int(dom[0])
    x = torch.cat([_rows(P["user_emb"], uid), _rows(P["item_emb"], pid), _rows(P["domain_emb"], dom)], dim=1)
    if training:
        mean = x.mean(dim=0)
        var = ((x - mean) ** 2).mean(dim=0)
    else:
        mean, var = state["mov_mean"][d], state["mov_var"][d]
    gamma = P["pn_gamma_shared"] * P["pn_gamma_spec"][d]
    beta = P["pn_beta_shared"] + P["pn_beta_spec"][d]
    h = (x - mean) * torch.rsqrt(var + PN_EPS) * gamma + beta
    for l in range(3):
        h = torch.relu(h @ (P["Ws%d" % l] * P["Wd%d" % l][d]) + P["bs%d" % l] + P["bd%d" % l][d])
    logit = (h @ P["wo"])[:, 0] + P["gb"][0]
    return torch.sigmoid(logit), mean.detach(), var.detach()


def fmnet_loss_and_grads(params, names, kind, uid, pid, dom, label, masks=None, rate=0.0, dtype=torch.float64,
                         uncertainty=False):
    """deepctr NFM / PNN (deepctr.py:33-35,44-46), float64 autograd.  NFM: linear tables + DNN(BiInteractionPooling of the
    three fields); PNN: DNN([fields | inner products of the field pairs (0,1), (0,2), (1,2)]).  l2 1e-5 on the tables
    (+ the linear tables for NFM)."""
    P = _as_tensors(params, names, dtype, set(names))
    ui, pi, di = (torch.from_numpy(np.asarray(a, np.int64)) for a in (uid, pid, dom))
    y = torch.from_numpy(np.asarray(label, np.float32)).to(dtype)
    u, i, d = _rows(P["user_emb"], ui), _rows(P["item_emb"], pi), _rows(P["domain_emb"], di)
    if kind == "nfm":
        s = u + i + d
        h = 0.5 * (s * s - (u * u + i * i + d * d))
    else:
        ip = torch.stack([(u * i).sum(1), (u * d).sum(1), (i * d).sum(1)], dim=1)
        h = torch.cat([u, i, d, ip], dim=1)
    keep = 1.0 / (1.0 - rate) if masks is not None else 1.0
    for l in range(3):
        h = torch.relu(torch.addmm(P["b%d" % l], h, P["W%d" % l]))
        if masks is not None:
            h = h * keep * torch.from_numpy(np.asarray(masks[l], np.float32)).to(dtype)
    logit = (h @ P["wo"])[:, 0] + P["gb"][0]
    reg = L2_EMB * (P["user_emb"].pow(2).sum() + P["item_emb"].pow(2).sum() + P["domain_emb"].pow(2).sum())
    if kind == "nfm":
        logit = logit + P["lin_user"][ui] + P["lin_item"][pi] + P["lin_domain"][di]
        reg = reg + L2_LIN * (P["lin_user"].pow(2).sum() + P["lin_item"].pow(2).sum() + P["lin_domain"].pow(2).sum())
    p = torch.sigmoid(logit)
    bce = keras_bce(p, y).mean()
    if uncertainty:            # weighted_loss.py:30-43: mean(BCE) / var^2 + log var, var = log_var[dom[0]]
        var = P["log_var"][int(dom[0])]
        loss = bce / (var * var) + torch.log(var) + reg
    else:
        loss = bce + reg
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    g = {n: (gr.numpy() if gr is not None else np.zeros(params[n].shape)) for n, gr in zip(names, grads)}
    return float(loss.detach()), g, p.detach().numpy()


def convnet_loss_and_grads(params, names, kind, uid, pid, dom, label, masks=None, rate=0.0, dtype=torch.float64,
                           uncertainty=False):
    """deepctr CCPM / AutoInt (deepctr.py:37-43) in float64 autograd -- written with torch's own conv2d / softmax, not with
    the oracle's loops.  CCPM: Conv2D((6, 1), 'same', tanh) over the field axis (TF pads an even kernel 2 before / 3 after),
    max over the fields, Conv2D((5, 1)) on the remaining row = its centre tap, tanh, flatten [128 x 4], DNN, linear tables.
    AutoInt: three InteractingLayers (4 heads x 8, residual, relu, no scaling) beside the DNN, one Dense(1) on both."""
    import torch.nn.functional as Fn
    P = _as_tensors(params, names, dtype, set(names))
    ui, pi, di = (torch.from_numpy(np.asarray(a, np.int64)) for a in (uid, pid, dom))
    y = torch.from_numpy(np.asarray(label, np.float32)).to(dtype)
    u, i, d = _rows(P["user_emb"], ui), _rows(P["item_emb"], pi), _rows(P["domain_emb"], di)
    x = torch.cat([u, i, d], dim=1)
    B, E = u.shape
    if kind == "ccpm":
        img = torch.stack([u, i, d], dim=1)[:, None]                         # [B, 1, 3 fields, E]
        k1 = P["conv1_w"].t()[:, None, :, None]                              # [4 out, 1 in, 6, 1]
        a1 = torch.tanh(Fn.conv2d(Fn.pad(img, (0, 0, 2, 3)), k1) + P["conv1_b"][None, :, None, None])      # [B, 4, 3, E]
        m1 = a1.max(dim=2).values                                            # [B, 4, E]
        a2 = torch.tanh(torch.einsum("bie,io->beo", m1, P["conv2_w"]) + P["conv2_b"])                         # [B, E, 4]
        h = a2.reshape(B, E * 4)
        att = None
    else:
        X = torch.stack([u, i, d], dim=1)                                    # [B, 3, E]
        for l in range(3):
            Pj = X @ P["att%d_w" % l]
            Q, K, V, R = (Pj[..., k * 32:(k + 1) * 32] for k in range(4))
            heads = lambda a: a.reshape(B, 3, 4, 8).permute(0, 2, 1, 3)
            A = torch.softmax(heads(Q) @ heads(K).transpose(-1, -2), dim=-1)
            O = (A @ heads(V)).permute(0, 2, 1, 3).reshape(B, 3, 32)
            X = torch.relu(O + R)
        att = X.reshape(B, 96)
        h = x
    keep = 1.0 / (1.0 - rate) if masks is not None else 1.0
    for l in range(3):
        h = torch.relu(torch.addmm(P["b%d" % l], h, P["W%d" % l]))
        if masks is not None:
            h = h * keep * torch.from_numpy(np.asarray(masks[l], np.float32)).to(dtype)
    top = h if att is None else torch.cat([att, h], dim=1)
    logit = (top @ P["wo"])[:, 0] + P["gb"][0] + P["lin_user"][ui] + P["lin_item"][pi] + P["lin_domain"][di]
    reg = L2_EMB * (P["user_emb"].pow(2).sum() + P["item_emb"].pow(2).sum() + P["domain_emb"].pow(2).sum()) + \
        L2_LIN * (P["lin_user"].pow(2).sum() + P["lin_item"].pow(2).sum() + P["lin_domain"].pow(2).sum())
    p = torch.sigmoid(logit)
    loss = keras_bce(p, y).mean()
    if uncertainty:            # weighted_loss.py:30-43: mean(BCE) / var^2 + log var, var = log_var[dom[0]]
        var = P["log_var"][int(dom[0])]
        loss = loss / (var * var) + torch.log(var)
    loss = loss + reg
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    g = {n: (gr.numpy() if gr is not None else np.zeros(params[n].shape)) for n, gr in zip(names, grads)}
    return float(loss.detach()), g, p.detach().numpy()


def mtl_forward(P, spec, d, uid, pid, dom, masks, keep_scale):
    """multi-task towers (deep_mtl_ctr.py:21-49; deepctr SharedBottom / MMOE / PLE with num_levels = 1), output of task d:
    experts = DNN(hidden_dim) on x; MMOE / PLE: gate_d = softmax(DNN(gate_dnn_hidden_units)(x) . Wg_d) over the experts
    the task mixes, mix = sum_e gate_e expert_e; tower_d = DNN(tower_hidden_dim); Dense(1, no bias) + global_bias_d ->
    sigmoid.  Every DNN layer: Dense -> relu -> dropout.  `spec` = oracle/mtl.Spec (structure only)."""
    x = torch.cat([_rows(P["user_emb"], uid), _rows(P["item_emb"], pid), _rows(P["domain_emb"], dom)], dim=1)

    def dnn(name, hidden, h):
        for l in range(len(hidden)):
            h = torch.relu(torch.addmm(P["%s/b%d" % (name, l)], h, P["%s/W%d" % (name, l)]))
            if masks is not None:
                h = h * keep_scale * masks["%s/W%d" % (name, l)]
        return h
    outs = [dnn(e, spec.expert_hidden, x) for e in spec.mix(d)]
    if spec.gated:
        q = dnn("gate_%d" % d, spec.gate_hidden, x)
        gate = torch.softmax(q @ P["gate_%d/Wg" % d], dim=1)
        m = sum(gate[:, k:k + 1] * o for k, o in enumerate(outs))
    else:
        m = outs[0]
    t = dnn("tower_%d" % d, spec.tower_hidden, m)
    return torch.sigmoid((t @ P["head_%d/w" % d])[:, 0] + P["head_%d/gb" % d][0])


def mtl_loss_and_grads(params, names, spec, d, uid, pid, dom, label, masks=None, rate=0.0, dtype=torch.float64):
    """loss (mean Keras BCE + l2 on the three tables) of task d and its autograd gradients for `names`."""
    P = _as_tensors(params, names, dtype, set(names))
    ui, pi, di = (torch.from_numpy(np.asarray(a, np.int64)) for a in (uid, pid, dom))
    y = torch.from_numpy(np.asarray(label, np.float32)).to(dtype)
    m = {k: torch.from_numpy(np.asarray(v, np.float32)).to(dtype) for k, v in masks.items()} if masks is not None else None
    keep = 1.0 / (1.0 - rate) if masks is not None else 1.0
    p = mtl_forward(P, spec, d, ui, pi, di, m, keep)
    loss = keras_bce(p, y).mean() + L2_EMB * (P["user_emb"].pow(2).sum() + P["item_emb"].pow(2).sum() + P["domain_emb"].pow(2).sum())
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    g = {n: (gr.numpy() if gr is not None else None) for n, gr in zip(names, grads)}
    return float(loss.detach()), g, p.detach().numpy()


def _as_tensors(params, names, dtype, trainable):
    out = {}
    for n, a in params.items():
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
        if n in trainable:
            t.requires_grad_(True)
        out[n] = t
    return out


def loss_and_grads(params, names, uid, pid, dom, label, masks=None, rate=0.0, tower="mlp", uncertainty=False,
                   state=None, dtype=torch.float64):
    """loss and d loss / d (every tensor in `names`) by autograd in `dtype`; numpy in, numpy out.
    tower "star": `state` carries the moving statistics (unused in training mode); masks / rate ignored."""
    P = _as_tensors(params, names, dtype, set(names))
    ui, pi, di = (torch.from_numpy(np.asarray(a, np.int64)) for a in (uid, pid, dom))
    y = torch.from_numpy(np.asarray(label, np.float32)).to(dtype)
    if tower == "star":
        p, mean, var = star_forward(P, None, ui, pi, di, True)
        loss = keras_bce(p, y).mean()
        extra = {"mean": mean.numpy(), "var": var.numpy()}
    else:
        m = [torch.from_numpy(np.asarray(k, np.float32)).to(dtype) for k in masks] if masks is not None else None
        keep = 1.0 / (1.0 - rate) if masks is not None else 1.0
        loss, p = deepctr_loss(P, ui, pi, di, y, m, keep, tower, uncertainty)
        extra = {}
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    g = {n: (gr.numpy() if gr is not None else np.zeros(params[n].shape)) for n, gr in zip(names, grads)}
    return float(loss.detach()), g, p.detach().numpy(), extra


def adam_step(params, grads, names, lr, t=1, m=None, v=None):
    """ONE tf.train.AdamOptimizer step in float64 (A.5; ApplyAdam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);
    m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); p -= lr_t m / (sqrt(v) + eps), eps OUTSIDE the bias
    correction).  Returns the new parameters (float64 numpy)."""
    lr_t = lr * np.sqrt(1.0 - BETA2 ** t) / (1.0 - BETA1 ** t)
    out = {}
    for n in names:
        g = np.asarray(grads[n], np.float64)
        m0 = np.zeros_like(g) if m is None else np.asarray(m[n], np.float64)
        v0 = np.zeros_like(g) if v is None else np.asarray(v[n], np.float64)
        m1 = m0 + (g - m0) * (1.0 - BETA1)
        v1 = v0 + (g * g - v0) * (1.0 - BETA2)
        out[n] = np.asarray(params[n], np.float64) - lr_t * m1 / (np.sqrt(v1) + ADAM_EPS)
    return out


class TorchCpuModel(object):
    """fp32 stand-in for the compiled Keras model on the host cores: what one `train_on_batch` costs a CPU.
    Dense TF1 Adam over every trainable tensor each step (with trainable tables that is the whole table: the
    l2 regulariser makes their gradient dense; Star's tables go through Adam's sparse apply, which still decays
    and moves every row -- A.5).  Dropout masks cycle through a pre-drawn pool (the mask stream is not timed)."""

    def __init__(self, params, names, tower="mlp", dropout=0.5, lr=1e-3, n_domain=None):
        self.names = list(names)
        self.tower = tower
        self.rate = float(dropout)
        self.lr = float(lr)
        self.P = _as_tensors(params, self.names, torch.float32, set(self.names))
        self.m = {n: torch.zeros_like(self.P[n]) for n in self.names}
        self.v = {n: torch.zeros_like(self.P[n]) for n in self.names}
        self.t = 0
        self._masks = None
        self.frozen_reg = None
        if tower != "star" and "user_emb" not in self.names:
            with torch.no_grad():
                self.frozen_reg = L2_EMB * (self.P["user_emb"].pow(2).sum() + self.P["item_emb"].pow(2).sum())
        if tower == "star":
            D, X = self.P["pn_gamma_spec"].shape
            self.state = {"mov_mean": torch.zeros(D, X), "mov_var": torch.ones(D, X),
                          "biased_mean": torch.zeros(D, X), "biased_var": torch.zeros(D, X), "steps": torch.zeros(D)}

    def train_on_batch(self, uid, pid, dom, label):
        ui, pi, di = (torch.from_numpy(np.asarray(a, np.int64)) for a in (uid, pid, dom))
        y = torch.from_numpy(np.asarray(label, np.float32))
        if self.tower == "star":
            p, mean, var = star_forward(self.P, self.state, ui, pi, di, True)
            loss = keras_bce(p, y).mean()
            d = int(di[0])
            st = self.state                       # assign_moving_average(zero_debias=True), momentum 0.99
            st["steps"][d] += 1.0
            factor = 1.0 - 0.99 ** float(st["steps"][d])
            for key, value in (("mean", mean), ("var", var)):
                st["biased_" + key][d] += (value - st["biased_" + key][d]) * 0.01
                st["mov_" + key][d] = st["biased_" + key][d] / factor
        else:
            masks, keep = None, 1.0
            if self.rate > 0:
                # torch's CPU generators are serial (3 ms per 1024 x 448 mask): the masks come from a small pool
                # drawn once, outside the timed steps -- the mask STREAM is not what this model is timed for
                B = ui.shape[0]
                if self._masks is None or self._masks[0][0].shape[0] < B:
                    self._masks = [[(torch.rand(B, h) >= self.rate).float() for h in (256, 128, 64)]
                                   for _ in range(8)]
                keep = 1.0 / (1.0 - self.rate)
                masks = [k[:B] for k in self._masks[self.t % 8]]
            loss, _ = deepctr_loss(self.P, ui, pi, di, y, masks, keep, self.tower, frozen_reg=self.frozen_reg)
        grads = torch.autograd.grad(loss, [self.P[n] for n in self.names], allow_unused=True)
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - BETA2 ** self.t) / (1.0 - BETA1 ** self.t)
        with torch.no_grad():
            for n, g in zip(self.names, grads):
                p, m, v = self.P[n], self.m[n], self.v[n]
                if g is None:
                    g = torch.zeros_like(p)
                m.add_(g - m, alpha=1.0 - BETA1)
                v.add_(g * g - v, alpha=1.0 - BETA2)
                p.sub_(lr_t * m / (v.sqrt() + ADAM_EPS))
        return float(loss.detach())
