"""CPU restatement of the Star tower (test infrastructure).

Follows model_zoo/Star/star.py:70-127 (structure: 3 x 128-d embeddings -> PartitionedNorm ->
3 x StarFCN(256,128,64, relu) -> Dense(1, sigmoid); auxiliary_net false, norm "pn", dense "star"),
model_zoo/Star/partitioned_norm.py:102-203 and model_zoo/Star/star_fcn.py:105-139.

PARITY UNPINNED: the arithmetic of those layers lives in tensorflow-gpu==1.12.0 (requirements.txt:1),
which cannot be installed here.  Restated from the published TF 1.12 algorithms:
  * d = domain_indicator[0, 0]: the whole batch uses the first row's domain (partitioned_norm.py:136,
    star_fcn.py:112).
  * PN training: K.normalize_batch_in_training -> nn.moments (population variance, two-pass) and
    nn.batch_normalization: inv = rsqrt(var + eps) * gamma; y = x * inv + (beta - mean * inv), with
    gamma = gamma_shared * gamma_specific[d], beta = beta_shared + beta_specific[d], eps = 1e-3
    (partitioned_norm.py:102-110,172-174).
  * moving statistics of domain d only: K.moving_average_update = assign_moving_average(x, value,
    momentum=0.99, zero_debias=True) in TF 1.12: biased += (value - biased) * (1 - momentum);
    step += 1; x = biased / (1 - momentum^step)  (partitioned_norm.py:177-193).  (SURVEY A.7 quotes the
    plain EMA; the zero-debiased form is what that TF version executes.)
  * inference: domain d's moving statistics (partitioned_norm.py:143-165).
  * StarFCN: kernel = kernel_shared * kernel_specific[d], bias = bias_shared + bias_specific[d]
    (star_fcn.py:105-110).  No dropout, no regularisers anywhere in this tower (star.py:70-127), so the
    embedding tables carry no L2 here.
  * the specific tensors are read through embedding_lookup, so their gradients are IndexedSlices and
    tf.train.AdamOptimizer applies its sparse rule, which decays m, v and moves EVERY slice (zero
    gradient for the other domains) -- arithmetically the dense rule of oracle/tower.Optimizer up to
    rounding.
"""
import numpy as np

from . import tower as T

F32 = np.float32
PN_EPS = F32(1e-3)
PN_MOMENTUM = F32(0.99)

META_FILTER = ("emb", "kernel_shared", "bias_shared")     # config/Taobao-10/star_taobao.json:37-41


def param_names(emb_trainable):
    """flat order: the meta parameters (name filter above: tables, shared kernels, shared biases) first,
    then the tensors that stay outside theta / phi."""
    emb = ("user_emb", "item_emb") if emb_trainable else ()
    meta = emb + ("domain_emb", "Ws0", "Ws1", "Ws2", "bs0", "bs1", "bs2")
    rest = ("pn_gamma_shared", "pn_beta_shared", "pn_gamma_spec", "pn_beta_spec",
            "Wd0", "Wd1", "Wd2", "bd0", "bd1", "bd2", "wo", "gb")
    return meta, rest


def glorot_uniform(rs, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(F32)


def init_params(rs, n_user, n_item, n_domain, emb_dim=128, hidden=(256, 128, 64)):
    """Keras defaults of the reference layers: Embedding uniform(-0.05, 0.05), kernels glorot uniform
    (the specific kernel's fans computed by Keras from its 3-d shape are approximated by the 2-d ones),
    biases zero, gamma one, beta zero."""
    p = {}
    p["user_emb"] = rs.uniform(-0.05, 0.05, (n_user, emb_dim)).astype(F32)
    p["item_emb"] = rs.uniform(-0.05, 0.05, (n_item, emb_dim)).astype(F32)
    p["domain_emb"] = rs.uniform(-0.05, 0.05, (n_domain, emb_dim)).astype(F32)
    dims = (3 * emb_dim,) + tuple(hidden)
    p["pn_gamma_shared"] = np.ones(dims[0], F32)
    p["pn_beta_shared"] = np.zeros(dims[0], F32)
    p["pn_gamma_spec"] = np.ones((n_domain, dims[0]), F32)
    p["pn_beta_spec"] = np.zeros((n_domain, dims[0]), F32)
    for l in range(3):
        p["Ws%d" % l] = glorot_uniform(rs, (dims[l], dims[l + 1]), dims[l], dims[l + 1])
        p["Wd%d" % l] = glorot_uniform(rs, (n_domain, dims[l], dims[l + 1]), dims[l], dims[l + 1])
        p["bs%d" % l] = np.zeros(dims[l + 1], F32)
        p["bd%d" % l] = np.zeros((n_domain, dims[l + 1]), F32)
    p["wo"] = glorot_uniform(rs, (dims[3], 1), dims[3], 1)
    p["gb"] = np.zeros(1, F32)
    return p


def init_state(n_domain, dim=384):
    """non-trainable PartitionedNorm state: moving mean / variance per domain (zeros / ones initialisers)
    and the zero-debias slots TF creates for each of them (biased accumulators, local step)."""
    return {"mov_mean": np.zeros((n_domain, dim), F32), "mov_var": np.ones((n_domain, dim), F32),
            "biased_mean": np.zeros((n_domain, dim), F32), "biased_var": np.zeros((n_domain, dim), F32),
            "steps": np.zeros(n_domain, F32)}


def effective(params, d):
    """the per-domain tower of this batch: PN scale / offset and merged kernels / biases."""
    gamma = (params["pn_gamma_shared"] * params["pn_gamma_spec"][d]).astype(F32)
    beta = (params["pn_beta_shared"] + params["pn_beta_spec"][d]).astype(F32)
    K = [(params["Ws%d" % l] * params["Wd%d" % l][d]).astype(F32) for l in range(3)]
    b = [(params["bs%d" % l] + params["bd%d" % l][d]).astype(F32) for l in range(3)]
    return gamma, beta, K, b


def batch_moments(x):
    mean = np.mean(x, axis=0, dtype=np.float64).astype(F32)
    var = np.mean(np.square((x - mean).astype(F32), dtype=F32), axis=0, dtype=np.float64).astype(F32)
    return mean, var


def forward(params, state, uid, pid, dom, training):
    """returns (p, cache); training -> batch statistics, else domain d's moving statistics."""
    d = int(dom[0])
    x = T.gather(params, uid, pid, dom)
    gamma, beta, K, b = effective(params, d)
    if training:
        mean, var = batch_moments(x)
    else:
        mean, var = state["mov_mean"][d], state["mov_var"][d]
    inv = (F32(1) / np.sqrt(var + PN_EPS, dtype=F32)).astype(F32)
    scale = (inv * gamma).astype(F32)
    xn = (x * scale + (beta - mean * scale).astype(F32)).astype(F32)
    hs = [xn]
    h = xn
    for l in range(3):
        h = np.maximum((h @ K[l] + b[l]).astype(F32), F32(0))
        hs.append(h)
    logit = (h @ params["wo"]).astype(F32)[:, 0] + params["gb"][0]
    p = T.sigmoid(logit)
    return p, dict(d=d, x=x, mean=mean, var=var, inv=inv, gamma=gamma, K=K, hs=hs)


def update_moving(state, d, mean, var):
    """assign_moving_average(zero_debias=True) for domain d's two moving variables."""
    state["steps"][d] += F32(1)
    factor = F32(F32(1) - np.power(PN_MOMENTUM, state["steps"][d], dtype=F32))
    for key, value in (("mean", mean), ("var", var)):
        biased = state["biased_" + key]
        biased[d] += ((value - biased[d]) * F32(F32(1) - PN_MOMENTUM)).astype(F32)
        state["mov_" + key][d] = (biased[d] / factor).astype(F32)


def loss_and_grads(params, state, uid, pid, dom, label, emb_trainable):
    """one training batch: BCE mean (no regularisers in this tower) and the gradient of every trainable."""
    B = uid.shape[0]
    p, c = forward(params, state, uid, pid, dom, True)
    d = c["d"]
    y = label.astype(F32)
    loss = F32(np.mean(T.bce_per_row(p, y), dtype=np.float64))
    inside = ((p >= T.EPS_CLIP) & (p <= F32(1) - T.EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    hs, K = c["hs"], c["K"]
    big = [n for n in ("user_emb", "item_emb") if emb_trainable and T.bigtable.use_rows(params[n])]
    g = {n: np.zeros_like(params[n]) for n in sum(param_names(emb_trainable), ()) if n not in big}
    g["wo"] = (hs[3].T @ dlogit[:, None]).astype(F32)
    g["gb"] = np.array([np.sum(dlogit, dtype=np.float64)], F32)
    dh = (dlogit[:, None] * params["wo"][:, 0][None, :]).astype(F32)
    for l in (2, 1, 0):
        dz = (dh * (hs[l + 1] > 0)).astype(F32)
        dK = (hs[l].T @ dz).astype(F32)
        db = np.sum(dz, axis=0, dtype=np.float64).astype(F32)
        g["Ws%d" % l] = (dK * params["Wd%d" % l][d]).astype(F32)
        g["Wd%d" % l][d] = (dK * params["Ws%d" % l]).astype(F32)
        g["bs%d" % l] = db
        g["bd%d" % l][d] = db
        dh = (dz @ K[l].T).astype(F32)
    # PartitionedNorm backward through the batch statistics
    dxn = dh
    xhat = ((c["x"] - c["mean"]) * c["inv"]).astype(F32)
    s1 = np.sum(dxn, axis=0, dtype=np.float64).astype(F32)                  # d beta_eff
    s2 = np.sum((dxn * xhat).astype(F32), axis=0, dtype=np.float64).astype(F32)   # d gamma_eff
    g["pn_beta_shared"] = s1
    g["pn_beta_spec"][d] = s1
    g["pn_gamma_shared"] = (s2 * params["pn_gamma_spec"][d]).astype(F32)
    g["pn_gamma_spec"][d] = (s2 * params["pn_gamma_shared"]).astype(F32)
    coef = (c["gamma"] * c["inv"]).astype(F32)
    dx = (coef * (dxn - (s1 / F32(B)).astype(F32) - (xhat * (s2 / F32(B)).astype(F32)).astype(F32))).astype(F32)
    E = params["domain_emb"].shape[1]
    g["domain_emb"][d] = np.sum(dx[:, 2 * E:], axis=0, dtype=np.float64).astype(F32)
    if emb_trainable:
        for name, ids, cols in (("user_emb", uid, slice(0, E)), ("item_emb", pid, slice(E, 2 * E))):
            if name in big:       # the same dense gradient (zero rows included), held sparsely: oracle/bigtable.py
                g[name] = T.bigtable.RowGrad(params[name], ids, dx[:, cols], 0.0)
                continue
            gt = np.zeros_like(params[name], dtype=np.float64)
            np.add.at(gt, ids, dx[:, cols].astype(np.float64))
            g[name] = gt.astype(F32)
    return loss, g, p, c


class OracleStar(object):
    """stand-in for the compiled Keras Star model: train_on_batch / evaluate, one Adam for all trainables."""

    def __init__(self, params, emb_trainable=True, lr=1e-3):
        self.params = params
        self.emb_trainable = emb_trainable
        self.meta_names, self.rest_names = param_names(emb_trainable)
        self.names = self.meta_names + self.rest_names
        self.state = init_state(params["domain_emb"].shape[0], 3 * params["domain_emb"].shape[1])
        self.opt = T.Optimizer(params, self.names)
        self.lr = lr
        self.use_sgd = False
        self.step = 0

    def get_flat(self, meta_only=False):
        return T.flatten(self.params, self.meta_names if meta_only else self.names)

    def set_flat(self, vec, meta_only=False):
        T.unflatten(vec, self.params, self.meta_names if meta_only else self.names)

    def train_on_batch(self, uid, pid, dom, label):
        loss, g, _, c = loss_and_grads(self.params, self.state, uid, pid, dom, label, self.emb_trainable)
        update_moving(self.state, c["d"], c["mean"], c["var"])
        if self.use_sgd:
            self.opt.sgd(self.params, g, self.lr)
        else:
            self.opt.adam(self.params, g, self.lr)
        self.step += 1
        return loss

    def train_pass(self, data, perm, batch_size, max_steps=0):
        n = perm.shape[0]
        n_step = -(-n // batch_size)
        if max_steps > 0:
            n_step = min(n_step, max_steps)
        return [self.train_on_batch(data["uid"][perm[s * batch_size:(s + 1) * batch_size]],
                                    data["pid"][perm[s * batch_size:(s + 1) * batch_size]],
                                    data["domain"][perm[s * batch_size:(s + 1) * batch_size]],
                                    data["label"][perm[s * batch_size:(s + 1) * batch_size]]) for s in range(n_step)]

    def evaluate(self, data, batch_size):
        n = data["uid"].shape[0]
        batch_losses = []
        preds = np.empty(n, F32)
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            p, _ = forward(self.params, self.state, data["uid"][sl], data["pid"][sl], data["domain"][sl], False)
            preds[sl] = p
            batch_losses.append(F32(np.mean(T.bce_per_row(p, data["label"][sl].astype(F32)), dtype=np.float64)))
        return F32(np.mean(np.array(batch_losses, np.float64))), preds
