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


def param_names(kind, emb_trainable, uncertainty=False):
    """(uncertainty: run.py:49-50 wraps ANY tower in the weighted loss -- one trainable `log_var` per domain, last)"""
    emb = ("user_emb", "item_emb") if emb_trainable else ()
    uw = ("log_var",) if uncertainty else ()
    if kind == "nfm":
        lin = ("lin_user", "lin_item") if emb_trainable else ()
        return emb + lin + ("domain_emb",) + T.DENSE_NAMES + ("lin_domain",) + uw
    return emb + ("domain_emb",) + T.DENSE_NAMES + uw


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


def loss_and_grads(P, kind, uid, pid, dom, label, masks, rate, emb_trainable, frozen_sumsq=None, uncertainty=False):
    """uncertainty (model_zoo/uncertainty_weight/weighted_loss.py:30-43, as oracle/tower.loss_and_grads): loss =
    mean(BCE) / var^2 + log var + regularisers, var = log_var[domain of the batch's first row]."""
    B = uid.shape[0]
    E = P["domain_emb"].shape[1]
    keep_scale = F32(1.0 / (1.0 - rate)) if masks is not None else F32(1)
    p, hs, x = forward(P, kind, uid, pid, dom, masks, keep_scale)
    y = label.astype(F32)
    mean_bce = F32(np.mean(T.bce_per_row(p, y), dtype=np.float64))
    inside = ((p >= T.EPS_CLIP) & (p <= F32(1) - T.EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    g = {}
    if uncertainty:
        d0 = int(dom[0])
        var = P["log_var"][d0]
        w = F32(F32(1) / F32(var * var))
        loss = F32(w * mean_bce) + F32(np.log(var, dtype=F32)) + reg_loss(P, kind, frozen_sumsq)
        dlogit = (dlogit * w).astype(F32)
        g["log_var"] = np.zeros_like(P["log_var"])
        g["log_var"][d0] = F32(F32(-2) * mean_bce / F32(var * var * var)) + F32(F32(1) / var)
    else:
        loss = mean_bce + reg_loss(P, kind, frozen_sumsq)
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

    def __init__(self, params, kind, emb_trainable=False, dropout=0.5, lr=1e-3, hidden=(256, 128, 64), dropout_seed=1024,
                 uncertainty=False):
        T.OracleModel.__init__(self, params, emb_trainable, dropout, lr, hidden, dropout_seed, "mlp", False)
        self.kind = kind
        self.conv = kind in ("ccpm", "autoint")
        self.uncertainty = bool(uncertainty)
        self.names = (ccpm_param_names(emb_trainable, uncertainty) if kind == "ccpm"
                      else autoint_param_names(emb_trainable, uncertainty)) if self.conv \
            else param_names(kind, emb_trainable, uncertainty)
        self.opt = T.Optimizer(params, self.names)

    def train_on_batch(self, uid, pid, dom, label):
        B = uid.shape[0]
        masks = T.train_masks(self.seed, self.step, B, self.hidden, self.rate) if self.rate > 0 else None
        if self.conv:
            loss, g, _ = loss_and_grads_conv(self.params, self.kind, uid, pid, dom, label, masks, self.rate, self.emb_trainable,
                                             self.frozen_sumsq(), self.uncertainty)
        else:
            loss, g, _ = loss_and_grads(self.params, self.kind, uid, pid, dom, label, masks, self.rate, self.emb_trainable,
                                        self.frozen_sumsq(), self.uncertainty)
        if self.use_sgd:
            self.opt.sgd(self.params, g, self.lr)
        else:
            self.opt.adam(self.params, g, self.lr)
        self.step += 1
        return loss

    def accumulate_on_batch(self, acc, uid, pid, dom, label):
        """the meta pass of first-order MAML / MLDG / PCGrad on these towers (maml.py:107-109,196-229): gradient of the
        total loss at the current weights added to `acc`, learning phase 0 (dropout off), no update."""
        if self.conv:
            _, g, _ = loss_and_grads_conv(self.params, self.kind, uid, pid, dom, label, None, 0.0, self.emb_trainable,
                                          self.frozen_sumsq(), self.uncertainty)
        else:
            _, g, _ = loss_and_grads(self.params, self.kind, uid, pid, dom, label, None, 0.0, self.emb_trainable,
                                     self.frozen_sumsq(), self.uncertainty)
        if getattr(self, "moving_average", None) is not None:
            from . import outer
            ma = self.moving_average
            ma["step"] = outer.moving_average_update(acc, ma["biased"], T.flatten(g, self.names), ma["momentum"], ma["step"])
        else:
            acc += T.flatten(g, self.names)
        self.step += 1

    def predict(self, uid, pid, dom):
        return (forward_conv if self.conv else forward)(self.params, self.kind, uid, pid, dom)[0]

    def evaluate(self, data, batch_size):
        n = data["uid"].shape[0]
        reg = reg_loss(self.params, "nfm" if self.conv else self.kind, self.frozen_sumsq())
        batch_losses, preds = [], np.empty(n, F32)
        for s in range(0, n, batch_size):
            sl = slice(s, min(n, s + batch_size))
            p = self.predict(data["uid"][sl], data["pid"][sl], data["domain"][sl])
            preds[sl] = p
            batch_losses.append(F32(np.mean(T.bce_per_row(p, data["label"][sl].astype(F32)), dtype=np.float64)) + reg)
        return F32(np.mean(np.array(batch_losses, np.float64))), preds


# ====================================================================================================
# CCPM and AutoInt (deepctr.py:37-43).  Same file conventions; tensor names are this build's own.
#
# CCPM  `models.CCPM(linear, dnn, dnn_hidden_units, dnn_dropout)`, defaults conv_kernel_width = (6, 5), conv_filters = (4, 4):
#       the three field embeddings as an image [3 fields x 128 x 1 channel]; layer 1: Conv2D(4 filters, kernel (6, 1), 'same',
#       tanh, bias) along the FIELD axis (TF 'same' for an even width: 2 taps before, 3 after), then KMaxPooling over the
#       fields with k = min(max(1, int((1 - (1/2)^1) 3)), 3) = 1, i.e. the maximum over the three fields; layer 2:
#       Conv2D(4, (5, 1), 'same', tanh, bias) on the single remaining row (only the centre tap meets data), k = min(3, 1) = 1;
#       flatten [128 x 4] -> DNN(hidden_dim) -> Dense(1, no bias) + linear tables + global bias -> sigmoid.
# AutoInt `models.AutoInt(linear, dnn, dnn_hidden_units, att_head_num=4, dnn_dropout)`, defaults att_layer_num = 3,
#       att_embedding_size = 8, att_res = True: three InteractingLayers on the [3 x d] field matrix (d = 128, then 32):
#       Q, K, V, R = X W_q, X W_k, X W_v, X W_res (each [d, 32]); per head h (8 columns): softmax(Q_h K_h^T) V_h over the
#       three fields (no scaling); concat heads, + R, relu.  Output [3 x 32] flattened (96) next to DNN(hidden_dim)(x);
#       Dense(1, no bias) on the 96 + 64 columns + linear tables + global bias -> sigmoid.
CCPM_W1, CCPM_F = 6, 4
ATT_LAYERS, ATT_HEADS, ATT_DIM = 3, 4, 8
ATT_OUT = ATT_HEADS * ATT_DIM          # 32


def ccpm_param_names(emb_trainable, uncertainty=False):
    emb = ("user_emb", "item_emb", "lin_user", "lin_item") if emb_trainable else ()
    return emb + ("domain_emb", "conv1_w", "conv1_b", "conv2_w", "conv2_b") + T.DENSE_NAMES + ("lin_domain",) + \
        (("log_var",) if uncertainty else ())


def autoint_param_names(emb_trainable, uncertainty=False):
    emb = ("user_emb", "item_emb", "lin_user", "lin_item") if emb_trainable else ()
    att = tuple("att%d_w" % l for l in range(ATT_LAYERS))
    return emb + ("domain_emb",) + att + T.DENSE_NAMES + ("lin_domain",) + (("log_var",) if uncertainty else ())


def init_params_conv(rs, kind, n_user, n_item, n_domain, emb_dim=128, hidden=(256, 128, 64), pretrained=True):
    p = T.init_params(rs, n_user, n_item, n_domain, emb_dim, hidden, pretrained)
    if kind == "ccpm":
        p["conv1_w"] = (rs.standard_normal((CCPM_W1, CCPM_F)) * 0.3).astype(F32)          # Conv2D kernel [6, 1, 1, 4]
        p["conv1_b"] = np.zeros(CCPM_F, F32)
        p["conv2_w"] = (rs.standard_normal((CCPM_F, CCPM_F)) * 0.3).astype(F32)           # centre tap of [5, 1, 4, 4]: [in, out]
        p["conv2_b"] = np.zeros(CCPM_F, F32)
        in_dim = emb_dim * CCPM_F
        s = np.sqrt(2.0 / (in_dim + hidden[0]))
        p["W0"] = (np.clip(rs.standard_normal((in_dim, hidden[0])), -2, 2) * s).astype(F32)
    else:
        d = emb_dim
        for l in range(ATT_LAYERS):             # [d, 4 * 32]: W_query | W_key | W_value | W_res, TruncatedNormal(0.05)
            p["att%d_w" % l] = (np.clip(rs.standard_normal((d, 4 * ATT_OUT)), -2, 2) * 0.05).astype(F32)
            d = ATT_OUT
        s = np.sqrt(2.0 / (3 * ATT_OUT + hidden[2] + 1))
        p["wo"] = (np.clip(rs.standard_normal((3 * ATT_OUT + hidden[2], 1)), -2, 2) * s).astype(F32)
    return p


def ccpm_features(P, x, E):
    """-> (f [B, E * 4] in (embedding column, filter) order, cache)."""
    B = x.shape[0]
    X = x.reshape(B, 3, E)
    w1, b1 = P["conv1_w"], P["conv1_b"]
    # 'same' padding of an even kernel in TF: pad_before = (6 - 1) // 2 = 2, pad_after = 3: out[p] = sum_t in[p + t - 2] w[t]
    pre1 = np.zeros((B, 3, E, CCPM_F), F32)
    for pos in range(3):
        for t in range(CCPM_W1):
            src = pos + t - 2
            if 0 <= src < 3:
                pre1[:, pos] += X[:, src, :, None] * w1[t][None, None, :]
    a1 = np.tanh((pre1 + b1).astype(F32), dtype=F32)
    arg = np.argmax(a1, axis=1)                                 # KMaxPooling k = 1: the maximum over the three fields
    m1 = np.take_along_axis(a1, arg[:, None], axis=1)[:, 0]     # [B, E, 4]
    pre2 = (m1 @ P["conv2_w"] + P["conv2_b"]).astype(F32)       # centre tap only: [B, E, 4]
    a2 = np.tanh(pre2, dtype=F32)
    return a2.reshape(B, E * CCPM_F), dict(X=X, a1=a1, arg=arg, m1=m1, a2=a2)


def ccpm_features_backward(P, c, df, g, E):
    """df = d loss / d f [B, E * 4]; fills the conv gradients; -> d x [B, 3 E]."""
    B = df.shape[0]
    dpre2 = (df.reshape(B, E, CCPM_F) * (F32(1) - c["a2"] * c["a2"])).astype(F32)
    g["conv2_w"] = np.einsum("bei,beo->io", c["m1"].astype(np.float64), dpre2.astype(np.float64)).astype(F32)
    g["conv2_b"] = np.sum(dpre2.astype(np.float64), axis=(0, 1)).astype(F32)
    dm1 = (dpre2 @ P["conv2_w"].T).astype(F32)
    da1 = np.zeros_like(c["a1"])
    np.put_along_axis(da1, c["arg"][:, None], dm1[:, None], axis=1)
    dpre1 = (da1 * (F32(1) - c["a1"] * c["a1"])).astype(F32)
    g["conv1_b"] = np.sum(dpre1.astype(np.float64), axis=(0, 1, 2)).astype(F32)
    gw = np.zeros((CCPM_W1, CCPM_F), np.float64)
    dX = np.zeros_like(c["X"])
    for pos in range(3):
        for t in range(CCPM_W1):
            src = pos + t - 2
            if 0 <= src < 3:
                gw[t] += np.einsum("be,bef->f", c["X"][:, src].astype(np.float64), dpre1[:, pos].astype(np.float64))
                dX[:, src] += (dpre1[:, pos] @ P["conv1_w"][t]).astype(F32)
    g["conv1_w"] = gw.astype(F32)
    return dX.reshape(B, 3 * E)


def att_layer_forward(W, X):
    """InteractingLayer: X [B, 3, d] -> relu(concat_h softmax(Q_h K_h^T) V_h + X W_res) [B, 3, 32]."""
    Pj = (X @ W).astype(F32)                                    # [B, 3, 128] = Q | K | V | R
    Q, K, V, R = (Pj[..., i * ATT_OUT:(i + 1) * ATT_OUT] for i in range(4))
    B = X.shape[0]
    Qh, Kh, Vh = (a.reshape(B, 3, ATT_HEADS, ATT_DIM).transpose(0, 2, 1, 3) for a in (Q, K, V))      # [B, H, 3, 8]
    S = (Qh @ Kh.transpose(0, 1, 3, 2)).astype(F32)             # [B, H, 3, 3]
    S = S - S.max(axis=-1, keepdims=True)
    A = np.exp(S, dtype=F32)
    A = (A / A.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    O = (A @ Vh).astype(F32).transpose(0, 2, 1, 3).reshape(B, 3, ATT_OUT)
    Y = np.maximum((O + R).astype(F32), F32(0))
    return Y, dict(X=X, Qh=Qh, Kh=Kh, Vh=Vh, A=A, Y=Y)


def att_layer_backward(W, c, dY):
    """-> (dX [B, 3, d], dW [d, 128])."""
    B = dY.shape[0]
    dZ = (dY * (c["Y"] > 0)).astype(F32)                        # relu
    dR = dZ
    dOh = dZ.reshape(B, 3, ATT_HEADS, ATT_DIM).transpose(0, 2, 1, 3)
    dA = (dOh @ c["Vh"].transpose(0, 1, 3, 2)).astype(F32)
    dVh = (c["A"].transpose(0, 1, 3, 2) @ dOh).astype(F32)
    dS = (c["A"] * (dA - np.sum((dA * c["A"]).astype(np.float64), axis=-1, keepdims=True).astype(F32))).astype(F32)
    dQh = (dS @ c["Kh"]).astype(F32)
    dKh = (dS.transpose(0, 1, 3, 2) @ c["Qh"]).astype(F32)
    back = lambda a: a.transpose(0, 2, 1, 3).reshape(B, 3, ATT_OUT)
    dP = np.concatenate([back(dQh), back(dKh), back(dVh), dR], axis=-1).astype(F32)      # [B, 3, 128]
    dW = np.einsum("btd,bte->de", c["X"].astype(np.float64), dP.astype(np.float64)).astype(F32)
    dX = (dP @ W.T).astype(F32)
    return dX, dW


def forward_conv(P, kind, uid, pid, dom, masks=None, keep_scale=F32(1)):
    E = P["domain_emb"].shape[1]
    x = T.gather(P, uid, pid, dom)
    c = {"x": x}
    if kind == "ccpm":
        f, c["conv"] = ccpm_features(P, x, E)
        h = f
    else:
        X = x.reshape(x.shape[0], 3, E)
        c["att"] = []
        for l in range(ATT_LAYERS):
            X, cl = att_layer_forward(P["att%d_w" % l], X)
            c["att"].append(cl)
        c["att_out"] = X.reshape(x.shape[0], 3 * ATT_OUT)
        h = x
    hs = [h]
    for l in range(3):
        a = np.maximum((h @ P["W%d" % l] + P["b%d" % l]).astype(F32), F32(0))
        if masks is not None:
            a = (a * keep_scale * masks[l]).astype(F32)
        hs.append(a)
        h = a
    top = h if kind == "ccpm" else np.concatenate([c["att_out"], h], axis=1).astype(F32)
    logit = (top @ P["wo"]).astype(F32)[:, 0] + P["gb"][0]
    logit = (logit + (P["lin_user"][uid] + P["lin_item"][pid] + P["lin_domain"][dom]).astype(F32)).astype(F32)
    c.update(hs=hs, top=top)
    return T.sigmoid(logit), c


def loss_and_grads_conv(P, kind, uid, pid, dom, label, masks, rate, emb_trainable, frozen_sumsq=None, uncertainty=False):
    B = uid.shape[0]
    E = P["domain_emb"].shape[1]
    keep_scale = F32(1.0 / (1.0 - rate)) if masks is not None else F32(1)
    p, c = forward_conv(P, kind, uid, pid, dom, masks, keep_scale)
    y = label.astype(F32)
    mean_bce = F32(np.mean(T.bce_per_row(p, y), dtype=np.float64))
    inside = ((p >= T.EPS_CLIP) & (p <= F32(1) - T.EPS_CLIP)).astype(F32)
    dlogit = ((p - y) * inside / F32(B)).astype(F32)
    g = {}
    if uncertainty:             # weighted_loss.py:30-43, as loss_and_grads above
        d0 = int(dom[0])
        var = P["log_var"][d0]
        w = F32(F32(1) / F32(var * var))
        loss = F32(w * mean_bce) + F32(np.log(var, dtype=F32)) + reg_loss(P, "nfm", frozen_sumsq)
        dlogit = (dlogit * w).astype(F32)
        g["log_var"] = np.zeros_like(P["log_var"])
        g["log_var"][d0] = F32(F32(-2) * mean_bce / F32(var * var * var)) + F32(F32(1) / var)
    else:
        loss = mean_bce + reg_loss(P, "nfm", frozen_sumsq)
    g["wo"] = (c["top"].T @ dlogit[:, None]).astype(F32)
    g["gb"] = np.array([np.sum(dlogit, dtype=np.float64)], F32)
    dtop = (dlogit[:, None] * P["wo"][:, 0][None, :]).astype(F32)
    hs = c["hs"]
    dh = dtop if kind == "ccpm" else dtop[:, 3 * ATT_OUT:]
    for l in (2, 1, 0):
        dz = (dh * ((hs[l + 1] > 0).astype(F32) * keep_scale)).astype(F32)
        g["W%d" % l] = (hs[l].T @ dz).astype(F32)
        g["b%d" % l] = np.sum(dz, axis=0, dtype=np.float64).astype(F32)
        dh = (dz @ P["W%d" % l].T).astype(F32)
    if kind == "ccpm":
        dx = ccpm_features_backward(P, c["conv"], dh, g, E)
    else:
        dx = dh
        dX = dtop[:, :3 * ATT_OUT].reshape(B, 3, ATT_OUT)
        for l in range(ATT_LAYERS - 1, -1, -1):
            dX, g["att%d_w" % l] = att_layer_backward(P["att%d_w" % l], c["att"][l], dX)
        dx = (dx + dX.reshape(B, 3 * E)).astype(F32)
    two_l2_lin = F32(2) * T.L2_LIN
    gl = np.bincount(dom, weights=dlogit.astype(np.float64), minlength=P["lin_domain"].shape[0])
    g["lin_domain"] = (gl.astype(F32) + two_l2_lin * P["lin_domain"]).astype(F32)
    two_l2 = F32(2) * T.L2_EMB
    onehot = (dom[:, None] == np.arange(P["domain_emb"].shape[0])[None, :]).astype(np.float64)
    g["domain_emb"] = ((onehot.T @ dx[:, 2 * E:].astype(np.float64)).astype(F32) + two_l2 * P["domain_emb"]).astype(F32)
    if emb_trainable:
        for name, ids, sl in (("user", uid, slice(0, E)), ("item", pid, slice(E, 2 * E))):
            gu = np.zeros_like(P[name + "_emb"], dtype=np.float64)
            np.add.at(gu, ids, dx[:, sl].astype(np.float64))
            g[name + "_emb"] = (gu.astype(F32) + two_l2 * P[name + "_emb"]).astype(F32)
            gl = np.bincount(ids, weights=dlogit.astype(np.float64), minlength=P["lin_" + name].shape[0])
            g["lin_" + name] = (gl.astype(F32) + two_l2_lin * P["lin_" + name]).astype(F32)
    return loss, g, p
