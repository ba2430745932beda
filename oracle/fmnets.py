"""fp32 numpy restatement of deepctr's NFM and PNN towers (test infrastructure).

Reference: model_zoo/DeepCTR/deepctr.py:33-35 (`models.NFM(linear_feature_columns, dnn_feature_columns,
dnn_hidden_units=hidden_dim, dnn_dropout=dropout)`) and :44-46 (`models.PNN(dnn_feature_columns, dnn_hidden_units,
dnn_dropout)`), same inputs / compile / training loops as the mlp tower (deepctr.py:54-93,95-116).

PARITY UNPINNED: the layers live in deepctr==0.9.0 (requirements.txt:6; not in the tree, not installable), the
reference holds no test for them.  Published algorithms restated (deepctr/models/{nfm,pnn}.py, layers/interaction.py):

* NFM  logit = sum_f w_f[id_f]  (1-d linear tables, Zeros init, l2_reg_linear 1e-5; the user / item ones train only
       when their feature column does)  +  DNN(bi) . w_o ;  bi = 1/2 ((sum_f e_f)^2 - sum_f e_f^2) over the three
       128-d fields (BiInteractionPooling; bi_dropout 0) ;  DNN = hidden_dim, relu, dropout ;  + global bias, sigmoid
* PNN  defaults use_inner = True, use_outter = False:  deep input = [e_u | e_i | e_d | <e_u,e_i> <e_u,e_d> <e_i,e_d>]
       (InnerProductLayer over the field pairs (0,1), (0,2), (1,2)) -> DNN(hidden_dim) -> Dense(1, no bias) -> + global
       bias, sigmoid ; no linear part
Both carry l2_reg_embedding 1e-5 on the three tables.  Tensor names follow oracle/tower.py; PNN's first kernel `W0` has
384 + 3 rows (the inner products feed its last three).
"""
import numpy as np

from . import tower as T

F32 = np.float32


def param_names(kind, emb_trainable):
    emb = ("user_emb", "item_emb") if emb_trainable else ()
    if kind == "nfm":
        lin = ("lin_user", "lin_item") if emb_trainable else ()
        return emb + lin + ("domain_emb",) + T.DENSE_NAMES + ("lin_domain",)
    return emb + ("domain_emb",) + T.DENSE_NAMES


def init_params(rs, kind, n_user, n_item, n_domain, emb_dim=128, hidden=(256, 128, 64), pretrained=True):
    p = T.init_params(rs, n_user, n_item, n_domain, emb_dim, hidden, pretrained)
    in_dim = emb_dim if kind == "nfm" else 3 * emb_dim + 3
    s = np.sqrt(2.0 / (in_dim + hidden[0]))
    p["W0"] = (np.clip(rs.standard_normal((in_dim, hidden[0])), -2, 2) * s).astype(F32)
    return p


def features(kind, x, E):
    u, i, d = x[:, :E], x[:, E:2 * E], x[:, 2 * E:]
    if kind == "nfm":
        return (u * i + u * d + i * d).astype(F32)             # = 1/2 ((u+i+d)^2 - u^2 - i^2 - d^2), elementwise
    ip = np.stack([np.sum((u * i).astype(np.float64), axis=1), np.sum((u * d).astype(np.float64), axis=1),
                   np.sum((i * d).astype(np.float64), axis=1)], axis=1).astype(F32)
    return np.concatenate([x, ip], axis=1).astype(F32)


def forward(P, kind, uid, pid, dom, masks=None, keep_scale=F32(1)):
    E = P["domain_emb"].shape[1]
    x = T.gather(P, uid, pid, dom)
    f = features(kind, x, E)
    hs, h = [f], f
    for l in range(3):
        a = np.maximum((h @ P["W%d" % l] + P["b%d" % l]).astype(F32), F32(0))
        if masks is not None:
            a = (a * keep_scale * masks[l]).astype(F32)
        hs.append(a)
        h = a
    logit = (h @ P["wo"]).astype(F32)[:, 0] + P["gb"][0]
    if kind == "nfm":
        logit = (logit + (P["lin_user"][uid] + P["lin_item"][pid] + P["lin_domain"][dom]).astype(F32)).astype(F32)
    return T.sigmoid(logit), hs, x


def reg_loss(P, kind, frozen_sumsq=None):
    r = T.reg_loss(P, frozen_sumsq, False)
    if kind == "nfm":
        for n in ("lin_user", "lin_item", "lin_domain"):
            r = F32(r + T.L2_LIN * T.table_sumsq(P[n]))
    return F32(r)


def loss_and_grads(P, kind, uid, pid, dom, label, masks, rate, emb_trainable, frozen_sumsq=None):
    B = uid.shape[0]
    E = P["domain_emb"].shape[1]
    keep_scale = F32(1.0 / (1.0 - rate)) if masks is not None else F32(1)
    p, hs, x = forward(P, kind, uid, pid, dom, masks, keep_scale)
    y = label.astype(F32)
    loss = F32(np.mean(T.bce_per_row(p, y), dtype=np.float64)) + reg_loss(P, kind, frozen_sumsq)
    inside = ((p >= T.EPS_CLIP) & (p <= F32(1) - T.EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    g = {}
    g["wo"] = (hs[3].T @ dlogit[:, None]).astype(F32)
    g["gb"] = np.array([np.sum(dlogit, dtype=np.float64)], F32)
    dh = (dlogit[:, None] * P["wo"][:, 0][None, :]).astype(F32)
    for l in (2, 1, 0):
        dz = (dh * ((hs[l + 1] > 0).astype(F32) * keep_scale)).astype(F32)
        g["W%d" % l] = (hs[l].T @ dz).astype(F32)
        g["b%d" % l] = np.sum(dz, axis=0, dtype=np.float64).astype(F32)
        dh = (dz @ P["W%d" % l].T).astype(F32)
    u, i, d = x[:, :E], x[:, E:2 * E], x[:, 2 * E:]
    if kind == "nfm":
        dx = np.concatenate([dh * (i + d), dh * (u + d), dh * (u + i)], axis=1).astype(F32)
        two_l2_lin = F32(2) * T.L2_LIN
        gl = np.bincount(dom, weights=dlogit.astype(np.float64), minlength=P["lin_domain"].shape[0])
        g["lin_domain"] = (gl.astype(F32) + two_l2_lin * P["lin_domain"]).astype(F32)
        if emb_trainable:
            gl = np.bincount(uid, weights=dlogit.astype(np.float64), minlength=P["lin_user"].shape[0])
            g["lin_user"] = (gl.astype(F32) + two_l2_lin * P["lin_user"]).astype(F32)
            gl = np.bincount(pid, weights=dlogit.astype(np.float64), minlength=P["lin_item"].shape[0])
            g["lin_item"] = (gl.astype(F32) + two_l2_lin * P["lin_item"]).astype(F32)
    else:
        dx = dh[:, :3 * E].copy()
        dip = dh[:, 3 * E:]
        dx[:, :E] += dip[:, 0:1] * i + dip[:, 1:2] * d
        dx[:, E:2 * E] += dip[:, 0:1] * u + dip[:, 2:3] * d
        dx[:, 2 * E:] += dip[:, 1:2] * u + dip[:, 2:3] * i
        dx = dx.astype(F32)
    two_l2 = F32(2) * T.L2_EMB
    onehot = (dom[:, None] == np.arange(P["domain_emb"].shape[0])[None, :]).astype(np.float64)
    g["domain_emb"] = ((onehot.T @ dx[:, 2 * E:].astype(np.float64)).astype(F32) + two_l2 * P["domain_emb"]).astype(F32)
    if emb_trainable:
        gu = np.zeros_like(P["user_emb"], dtype=np.float64)
        np.add.at(gu, uid, dx[:, :E].astype(np.float64))
        g["user_emb"] = (gu.astype(F32) + two_l2 * P["user_emb"]).astype(F32)
        gi = np.zeros_like(P["item_emb"], dtype=np.float64)
        np.add.at(gi, pid, dx[:, E:2 * E].astype(np.float64))
        g["item_emb"] = (gi.astype(F32) + two_l2 * P["item_emb"]).astype(F32)
    return loss, g, p


class OracleNet(T.OracleModel):
    """OracleModel with the NFM / PNN forward and gradients (same optimiser, weights in / out, passes)."""

    def __init__(self, params, kind, emb_trainable=False, dropout=0.5, lr=1e-3, hidden=(256, 128, 64), dropout_seed=1024):
        T.OracleModel.__init__(self, params, emb_trainable, dropout, lr, hidden, dropout_seed, "mlp", False)
        self.kind = kind
        self.names = param_names(kind, emb_trainable)
        self.opt = T.Optimizer(params, self.names)

    def train_on_batch(self, uid, pid, dom, label):
        B = uid.shape[0]
        masks = T.train_masks(self.seed, self.step, B, self.hidden, self.rate) if self.rate > 0 else None
        loss, g, _ = loss_and_grads(self.params, self.kind, uid, pid, dom, label, masks, self.rate, self.emb_trainable,
                                    self.frozen_sumsq())
        if self.use_sgd:
            self.opt.sgd(self.params, g, self.lr)
        else:
            self.opt.adam(self.params, g, self.lr)
        self.step += 1
        return loss

    def predict(self, uid, pid, dom):
        return forward(self.params, self.kind, uid, pid, dom)[0]

    def evaluate(self, data, batch_size):
        n = data["uid"].shape[0]
        reg = reg_loss(self.params, self.kind, self.frozen_sumsq())
        batch_losses, preds = [], np.empty(n, F32)
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            p = self.predict(data["uid"][sl], data["pid"][sl], data["domain"][sl])
            preds[sl] = p
            batch_losses.append(F32(np.mean(T.bce_per_row(p, data["label"][sl].astype(F32)), dtype=np.float64)) + reg)
        return F32(np.mean(np.array(batch_losses, np.float64))), preds
