"""fp32 numpy restatement of the reference's inner step (test infrastructure).

One "domain-step"  = `model.train_on_batch(iter)` at
model_zoo/mamdr.py:54,86,97, model_zoo/domain_negotiation.py:72,
model_zoo/reptile.py:70, model_zoo/DeepCTR/deepctr.py:76 on the model built by
model_zoo/DeepCTR/deepctr.py:95-136 and compiled at deepctr.py:54-60.

PARITY UNPINNED: the arithmetic below lives in tensorflow-gpu==1.12.0 /
deepctr==0.9.0 (requirements.txt:1,6; not in the tree, not installable).  It
restates their published algorithms:

* tower      deepctr.py:118-136 -> concat(user, item, domain 128-d rows) ->
             deepctr `DNN(256,128,64, relu, dropout)` -> Dense(1, no bias) ->
             PredictionLayer('binary') (= + global_bias, sigmoid)
* regulariser deepctr.py:119,125-126 l2_reg_embedding=1e-5 on all three tables
* loss       deepctr.py:59 Keras binary_crossentropy (clip 1e-7, logit form)
* optimiser  deepctr.py:55 tf.train.AdamOptimizer (ApplyAdam kernel form);
             finetune: GradientDescentOptimizer (specific_base_model.py:120,
             base_model.py:69)
"""
import numpy as np

from . import bigtable, rng

F32 = np.float32
EPS_CLIP = F32(1e-7)        # K.epsilon()
BETA1 = F32(0.9)
BETA2 = F32(0.999)
ADAM_EPS = F32(1e-8)
L2_EMB = F32(1e-5)          # deepctr.py:118 l2_reg_embedding
L2_LIN = F32(1e-5)          # deepctr DeepFM l2_reg_linear default

def dense_names(n_layers=3):
    """DNN kernels, DNN biases, final dense kernel, global bias -- deepctr's `DNN(hidden_units)` creates ALL kernels
    before ALL biases (layers/core.py build()), for any number of hidden layers (deepctr.py:26-49 passes
    `hidden_dim` through as `dnn_hidden_units`)."""
    return tuple("W%d" % l for l in range(n_layers)) + tuple("b%d" % l for l in range(n_layers)) + ("wo", "gb")


DENSE_NAMES = dense_names(3)      # the reference's configs: hidden_dim [256, 128, 64]


def n_layers(params):
    n = 0
    while "W%d" % n in params:
        n += 1
    return n


def param_names(emb_trainable, deepfm=False, uncertainty=False, n_layers=3):
    """Flat meta-vector order = Keras `trainable_weights` order:
    trainable embeddings in feature order (deepctr.py:102), DNN kernels, DNN
    biases, final dense kernel, global bias.  DeepFM (A.8) adds the 1-d linear
    tables; the user / item ones inherit `trainable` from their feature column
    (deepctr create_embedding_dict), so they stay frozen at their zero
    initialisation unless emb_trainable.  (Their position in Keras' order is [dep];
    here: user/item linear tables behind the embedding tables, the domain one last.)"""
    emb = ("user_emb", "item_emb") if emb_trainable else ()
    lin = ("lin_user", "lin_item") if (emb_trainable and deepfm) else ()
    tail = ("lin_domain",) if deepfm else ()
    # uncertainty weighting (model_zoo/uncertainty_weight/weighted_loss.py:23-28): one trainable scalar per domain
    tail = tail + (("log_var",) if uncertainty else ())
    return emb + lin + ("domain_emb",) + dense_names(n_layers) + tail


def init_params(rs, n_user, n_item, n_domain, emb_dim=128, hidden=(256, 128, 64),
                pretrained=True):
    """Injected initial tensors (TF initialiser streams are unreproducible, A.2).
    glorot-normal-like kernels, zero biases, N(0,1e-4^2) domain table, N(0,0.1^2)
    'pretrained' user/item tables."""
    p = {}
    sd = 0.1 if pretrained else 1e-4
    p["user_emb"] = (rs.standard_normal((n_user, emb_dim)) * sd).astype(F32)
    p["item_emb"] = (rs.standard_normal((n_item, emb_dim)) * sd).astype(F32)
    p["domain_emb"] = (rs.standard_normal((n_domain, emb_dim)) * 1e-4).astype(F32)
    dims = (3 * emb_dim,) + tuple(hidden)
    for l in range(len(hidden)):
        s = np.sqrt(2.0 / (dims[l] + dims[l + 1]))
        p["W%d" % l] = (np.clip(rs.standard_normal((dims[l], dims[l + 1])), -2, 2) * s).astype(F32)
        p["b%d" % l] = np.zeros(dims[l + 1], F32)
    s = np.sqrt(2.0 / (dims[-1] + 1))
    p["wo"] = (np.clip(rs.standard_normal((dims[-1], 1)), -2, 2) * s).astype(F32)
    p["gb"] = np.zeros(1, F32)
    # DeepFM linear tables: Zeros initialiser (deepctr get_linear_logit)
    p["lin_user"] = np.zeros(n_user, F32)
    p["lin_item"] = np.zeros(n_item, F32)
    p["lin_domain"] = np.zeros(n_domain, F32)
    # uncertainty weighting: `log_var` [D, 1], Constant(1.) (weighted_loss.py:23-28; despite its name it is
    # used as the standard deviation: weight 1 / var^2, penalty log(var))
    p["log_var"] = np.ones(n_domain, F32)
    return p


def flatten(params, names):
    return np.concatenate([bigtable.densify(params[n]).ravel() for n in names]).astype(F32)


def unflatten(vec, params, names):
    """write `vec` back into the arrays of `params` (in place)."""
    o = 0
    for n in names:
        a = params[n]
        a[...] = vec[o:o + a.size].reshape(a.shape)
        o += a.size
    assert o == vec.size


def sigmoid(z):
    z = z.astype(F32)
    out = np.empty_like(z)
    pos = z >= 0
    ez = np.exp(-z[pos], dtype=F32)
    out[pos] = F32(1) / (F32(1) + ez)
    ez = np.exp(z[~pos], dtype=F32)
    out[~pos] = ez / (F32(1) + ez)
    return out


def bce_per_row(p, y):
    """Keras binary_crossentropy (A.4): clip, back to logits, stable CE."""
    pc = np.clip(p, EPS_CLIP, F32(1) - EPS_CLIP).astype(F32)
    z = np.log(pc / (F32(1) - pc), dtype=F32)
    return (np.maximum(z, F32(0)) - z * y + np.log1p(np.exp(-np.abs(z), dtype=F32), dtype=F32)).astype(F32)


def table_sumsq(table):
    return bigtable.table_sumsq(table)


def reg_loss(params, frozen_sumsq=None, deepfm=False, l2=None):
    """deepctr l2_reg_embedding * sum(W^2) on every table, frozen or not (A.3); DeepFM adds
    l2_reg_linear * sum(w^2) on its three linear tables (A.8).
    frozen_sumsq: optional {name: sum of squares} of tables that never change (computed once)."""
    l2_emb, l2_lin = (L2_EMB, L2_LIN) if l2 is None else (F32(l2[0]), F32(l2[1]))    # (l2: Star's plain-DNN form has none)
    r = F32(0)
    for n in ("user_emb", "item_emb", "domain_emb"):
        ss = frozen_sumsq[n] if frozen_sumsq and n in frozen_sumsq else table_sumsq(params[n])
        r = F32(r + l2_emb * ss)
    if deepfm:
        for n in ("lin_user", "lin_item", "lin_domain"):
            r = F32(r + l2_lin * table_sumsq(params[n]))
    return F32(r)


def gather(params, uid, pid, dom):
    """K1: x[b] = [U[uid_b] | I[pid_b] | Dm[dom_b]] (deepctr.py:102 feature order)."""
    return np.concatenate([params["user_emb"][uid], params["item_emb"][pid],
                           params["domain_emb"][dom]], axis=1)


def fm_and_linear(params, x, uid, pid, dom, with_fm=True):
    """DeepFM extras (A.8): sum_f w_f[id_f] + 1/2 sum_k [(sum_f e_fk)^2 - sum_f e_fk^2] over the three
    128-d fields = sum_k (u i + u d + i d)_k.  with_fm=False: the linear ("wide") part only = deepctr's WDL
    (model_zoo/DeepCTR/deepctr.py:29-32: final_logit = linear_logit + dnn_logit)."""
    E = params["domain_emb"].shape[1]
    u, i, d = x[:, :E], x[:, E:2 * E], x[:, 2 * E:]
    lin = (params["lin_user"][uid] + params["lin_item"][pid] + params["lin_domain"][dom]).astype(F32)
    if not with_fm:
        return lin
    fm = np.sum((u * i + u * d + i * d).astype(np.float64), axis=1).astype(F32)
    return (fm + lin).astype(F32)


def forward(params, uid, pid, dom, masks=None, keep_scale=F32(1), deepfm=False):
    """returns (p, cache). masks: 3 float32 keep masks or None (inference)."""
    x = gather(params, uid, pid, dom)
    hs = [x]
    h = x
    for l in range(n_layers(params)):
        z = (h @ params["W%d" % l] + params["b%d" % l]).astype(F32)
        a = np.maximum(z, F32(0))
        if masks is not None:
            a = (a * keep_scale * masks[l]).astype(F32)
        hs.append(a)
        h = a
    logit = (h @ params["wo"]).astype(F32)[:, 0] + params["gb"][0]
    if deepfm:                                   # 1 / True: DeepFM, 2: WDL (linear part only)
        logit = (logit + fm_and_linear(params, x, uid, pid, dom, deepfm != 2)).astype(F32)
    p = sigmoid(logit)
    return p, hs


def train_masks(seed, step, n_rows, hidden, rate):
    return [rng.dropout_mask(seed, step, l, n_rows, hidden[l], rate) for l in range(len(hidden))]


def loss_and_grads(params, uid, pid, dom, label, masks, rate, emb_trainable, frozen_sumsq=None, deepfm=False,
                   uncertainty=False, l2=None):
    """one batch: total loss (BCE mean + regularisers) and dense gradients.
    uncertainty (weighted_loss.py:30-43): loss = mean(BCE / var^2 + log var) + regularisers with
    var = log_var[domain of the batch's first row]."""
    B = uid.shape[0]
    keep_scale = F32(1.0 / (1.0 - rate)) if masks is not None else F32(1)
    p, hs = forward(params, uid, pid, dom, masks, keep_scale, deepfm)
    y = label.astype(F32)
    mean_bce = F32(np.mean(bce_per_row(p, y), dtype=np.float64))
    inside = ((p >= EPS_CLIP) & (p <= F32(1) - EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    g = {}
    if uncertainty:
        d0 = int(dom[0])
        var = params["log_var"][d0]
        w = F32(F32(1) / F32(var * var))
        loss = F32(w * mean_bce) + F32(np.log(var, dtype=F32)) + reg_loss(params, frozen_sumsq, deepfm, l2)
        dlogit = (dlogit * w).astype(F32)
        g["log_var"] = np.zeros_like(params["log_var"])
        g["log_var"][d0] = F32(F32(-2) * mean_bce / F32(var * var * var)) + F32(F32(1) / var)
    else:
        loss = mean_bce + reg_loss(params, frozen_sumsq, deepfm, l2)
    L = len(hs) - 1
    g["wo"] = (hs[L].T @ dlogit[:, None]).astype(F32)
    g["gb"] = np.array([np.sum(dlogit, dtype=np.float64)], F32)
    dh = (dlogit[:, None] * params["wo"][:, 0][None, :]).astype(F32)
    for l in range(L - 1, -1, -1):
        gate = (hs[l + 1] > 0).astype(F32) * keep_scale     # relu' * dropout mask * 1/keep
        dz = (dh * gate).astype(F32)
        g["W%d" % l] = (hs[l].T @ dz).astype(F32)
        g["b%d" % l] = np.sum(dz, axis=0, dtype=np.float64).astype(F32)
        dh = (dz @ params["W%d" % l].T).astype(F32)
    E = params["domain_emb"].shape[1]
    two_l2 = F32(2) * (L2_EMB if l2 is None else F32(l2[0]))
    if deepfm:
        # d fm / d e_f = sum of the other two fields' embeddings; d linear / d w_f[id] = 1
        x = hs[0]
        u, it, d = x[:, :E], x[:, E:2 * E], x[:, 2 * E:]
        if deepfm != 2:                          # WDL has no FM term
            dh = dh.copy()
            dh[:, :E] += dlogit[:, None] * (it + d)
            dh[:, E:2 * E] += dlogit[:, None] * (u + d)
            dh[:, 2 * E:] += dlogit[:, None] * (u + it)
        two_l2_lin = F32(2) * (L2_LIN if l2 is None else F32(l2[1]))
        nd = params["lin_domain"].shape[0]
        gl = np.bincount(dom, weights=dlogit.astype(np.float64), minlength=nd)
        g["lin_domain"] = (gl.astype(F32) + two_l2_lin * params["lin_domain"]).astype(F32)
        if emb_trainable:
            gl = np.bincount(uid, weights=dlogit.astype(np.float64), minlength=params["lin_user"].shape[0])
            g["lin_user"] = (gl.astype(F32) + two_l2_lin * params["lin_user"]).astype(F32)
            gl = np.bincount(pid, weights=dlogit.astype(np.float64), minlength=params["lin_item"].shape[0])
            g["lin_item"] = (gl.astype(F32) + two_l2_lin * params["lin_item"]).astype(F32)
    # segmented sum over the few domain rows as a one-hot contraction (float64 accumulation)
    onehot = (dom[:, None] == np.arange(params["domain_emb"].shape[0])[None, :]).astype(np.float64)
    gd = onehot.T @ dh[:, 2 * E:3 * E].astype(np.float64)
    g["domain_emb"] = (gd.astype(F32) + two_l2 * params["domain_emb"]).astype(F32)
    if emb_trainable:
        for name, ids, cols in (("user_emb", uid, slice(0, E)), ("item_emb", pid, slice(E, 2 * E))):
            if bigtable.use_rows(params[name]):
                # the same dense gradient, held as (touched rows, sums, 2 l2): oracle/bigtable.py
                g[name] = bigtable.RowGrad(params[name], ids, dh[:, cols], two_l2)
                continue
            gt = np.zeros_like(params[name], dtype=np.float64)
            np.add.at(gt, ids, dh[:, cols].astype(np.float64))
            g[name] = (gt.astype(F32) + two_l2 * params[name]).astype(F32)
    return loss, g, p


def beta_powers(t):
    """fp32 running products beta^t as TF keeps them (beta1_power variable)."""
    b1, b2 = F32(1), F32(1)
    for _ in range(t):
        b1 = F32(b1 * BETA1)
        b2 = F32(b2 * BETA2)
    return b1, b2


class Optimizer(object):
    """TF1 Adam (A.5): one (m, v) slot per variable and one step count for the
    whole run -- never reset by weight assignment, between domains, between DN
    and DR, or across epochs."""

    def __init__(self, params, names):
        self.names = names
        self.m = {n: np.zeros_like(params[n]) for n in names}
        self.v = {n: np.zeros_like(params[n]) for n in names}
        self.t = 0
        self.b1p = F32(1)
        self.b2p = F32(1)
        self.eps = ADAM_EPS     # tf.train.AdamOptimizer's 1e-8 (a Keras Adam compiled by name has K.epsilon() = 1e-7)

    def adam(self, params, grads, lr):
        self.t += 1
        self.b1p = F32(self.b1p * BETA1)
        self.b2p = F32(self.b2p * BETA2)
        alpha = F32(F32(lr) * np.sqrt(F32(1) - self.b2p, dtype=F32) / (F32(1) - self.b1p))
        omb1 = F32(F32(1) - BETA1)
        omb2 = F32(F32(1) - BETA2)
        for n in self.names:
            gr = grads[n]
            m, v = self.m[n], self.v[n]
            if isinstance(gr, bigtable.RowGrad):
                bigtable.adam_rows(params[n], m, v, gr, alpha, omb1, omb2, self.eps)
                continue
            m += ((gr - m) * omb1).astype(F32)
            v += ((gr * gr - v) * omb2).astype(F32)
            params[n] -= ((m * alpha) / (np.sqrt(v, dtype=F32) + self.eps)).astype(F32)

    def sgd(self, params, grads, lr):
        for n in self.names:
            if isinstance(grads[n], bigtable.RowGrad):
                bigtable.sgd_rows(params[n], grads[n], F32(lr))
                continue
            params[n] -= (grads[n] * F32(lr)).astype(F32)


class OracleModel(object):
    """Stand-in for the compiled Keras model: train_on_batch / evaluate."""

    def __init__(self, params, emb_trainable=False, dropout=0.5, lr=1e-3, hidden=(256, 128, 64),
                 dropout_seed=1024, tower="mlp", uncertainty=False, l2_emb=None, l2_linear=None):
        self.params = params
        # regularisers (deepctr's defaults; Star's plain-DNN form passes 0 / 0)
        self.l2_emb = float(L2_EMB if l2_emb is None else l2_emb)
        self.l2_linear = float(L2_LIN if l2_linear is None else l2_linear)
        self.l2 = None if (l2_emb is None and l2_linear is None) else (F32(self.l2_emb), F32(self.l2_linear))
        self.dropout = float(dropout)
        self.emb_trainable = emb_trainable
        self.deepfm = {"deepfm": 1, "wdl": 2}.get(tower, 0)      # tower with linear tables (+ FM term for 1)
        self.uncertainty = bool(uncertainty)
        self.names = param_names(emb_trainable, self.deepfm, self.uncertainty, len(hidden))
        self.opt = Optimizer(params, self.names)
        self.rate = float(dropout)
        self.lr = lr
        self.hidden = hidden
        self.seed = dropout_seed
        self.step = 0          # global inner-step counter (dropout stream + Adam t)
        self.use_sgd = False
        self._frozen = None    # (ids of the frozen tables, their sums of squares)

    # weights in / out (maml.py:181-194, utils/tool.py:36-45)
    def get_flat(self):
        return flatten(self.params, self.names)

    def set_flat(self, vec):
        unflatten(vec, self.params, self.names)

    def train_on_batch(self, uid, pid, dom, label):
        B = uid.shape[0]
        masks = train_masks(self.seed, self.step, B, self.hidden, self.rate) if self.rate > 0 else \
            [np.ones((B, h), F32) for h in self.hidden]
        loss, g, _ = loss_and_grads(self.params, uid, pid, dom, label, masks, self.rate, self.emb_trainable,
                                    self.frozen_sumsq(), self.deepfm, self.uncertainty, self.l2)
        if self.use_sgd:
            self.opt.sgd(self.params, g, self.lr)
        else:
            self.opt.adam(self.params, g, self.lr)
        self.step += 1
        return loss

    def frozen_sumsq(self):
        """sums of squares of the frozen tables, recomputed only when a table object is replaced."""
        if self.emb_trainable:
            return None
        key = (id(self.params["user_emb"]), id(self.params["item_emb"]))
        if self._frozen is None or self._frozen[0] != key:
            self._frozen = (key, {n: table_sumsq(self.params[n]) for n in ("user_emb", "item_emb")})
        return self._frozen[1]

    def accumulate_on_batch(self, acc, uid, pid, dom, label):
        """meta pass of first-order MAML (model_zoo/maml.py:107-109,196-229): add
        d total_loss / d theta at the current weights to `acc` (a flat vector); no update,
        learning phase 0 = dropout off.  The dropout counter still advances (one per step)."""
        _, g, _ = loss_and_grads(self.params, uid, pid, dom, label, None, 0.0, self.emb_trainable,
                                 self.frozen_sumsq(), self.deepfm, self.uncertainty, self.l2)
        if getattr(self, "moving_average", None) is not None:       # average_meta_grad == "moving_mean" (maml.py:219-220)
            from . import outer
            ma = self.moving_average
            ma["step"] = outer.moving_average_update(acc, ma["biased"], flatten(g, self.names), ma["momentum"], ma["step"])
        else:
            acc += flatten(g, self.names)
        self.step += 1

    def train_pass(self, data, perm, batch_size, max_steps=0, accumulate_into=None):
        """one pass over one domain's train split in `perm` order; final partial
        batch kept (utils/dataset.py:25)."""
        n = perm.shape[0]
        n_step = -(-n // batch_size)
        if max_steps > 0:
            n_step = min(n_step, max_steps)
        losses = []
        for s in range(n_step):
            idx = perm[s * batch_size:(s + 1) * batch_size]
            if accumulate_into is not None:
                self.accumulate_on_batch(accumulate_into, data["uid"][idx], data["pid"][idx], data["domain"][idx],
                                         data["label"][idx])
                losses.append(None)
            else:
                losses.append(self.train_on_batch(data["uid"][idx], data["pid"][idx], data["domain"][idx],
                                                  data["label"][idx]))
        return losses


    def predict(self, uid, pid, dom):
        p, _ = forward(self.params, uid, pid, dom, None, deepfm=self.deepfm)
        return p

    def evaluate(self, data, batch_size):
        """Keras evaluate (A.6): loss = mean over batches of batch-mean loss (+reg),
        predictions for the AUC over all rows in file order."""
        n = data["uid"].shape[0]
        reg = reg_loss(self.params, self.frozen_sumsq(), self.deepfm, self.l2)
        batch_losses = []
        preds = np.empty(n, F32)
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            p = self.predict(data["uid"][sl], data["pid"][sl], data["domain"][sl])
            preds[sl] = p
            batch_losses.append(F32(np.mean(bce_per_row(p, data["label"][sl].astype(F32)), dtype=np.float64)) + reg)
        return F32(np.mean(np.array(batch_losses, np.float64))), preds


class OuterAdam(object):
    """the separate tf.train.AdamOptimizer(meta_learning_rate) of MAML (model_zoo/maml.py:201)."""

    def __init__(self, n):
        self.m = np.zeros(n, F32)
        self.v = np.zeros(n, F32)
        self.b1p = F32(1)
        self.b2p = F32(1)

    def apply(self, theta, grad, lr, grad_scale=1.0):
        self.b1p = F32(self.b1p * BETA1)
        self.b2p = F32(self.b2p * BETA2)
        alpha = F32(F32(lr) * np.sqrt(F32(1) - self.b2p, dtype=F32) / (F32(1) - self.b1p))
        g = (grad * F32(grad_scale)).astype(F32)
        self.m += ((g - self.m) * F32(F32(1) - BETA1)).astype(F32)
        self.v += ((g * g - self.v) * F32(F32(1) - BETA2)).astype(F32)
        theta -= ((self.m * alpha) / (np.sqrt(self.v, dtype=F32) + ADAM_EPS)).astype(F32)
