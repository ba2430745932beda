"""Star tower -- host-side mirror of model_zoo/Star/star.py.

Structure (star.py:70-97): 3 x 128-d embeddings -> PartitionedNorm (`norm: "pn"`) -> StarFCN per hidden
width (`dense: "star"`) -> Dense(1, sigmoid); the AuxiliaryNet branch is built by the reference but only
joins the graph when `auxiliary_net` is true, which no BASELINE config sets -- it raises here.  `norm: "none"` with
`dense: "dense"` (star.py:74-87: no normalisation, plain Keras Dense layers) is the mlp tower without dropout and
without regularisers, Keras initial values, Keras variable names (`dense/kernel` ...): it runs on the mlp engines
(step kernels for [256, 128, 64], the generic-layer engine otherwise).  `norm: "bn"` and the mixed forms raise.  The
plain `star` name trains with the same alternate loop as DeepCTR (star.py:34-69 == deepctr.py:63-93).
Initial tensors follow the Keras defaults of the reference's layers: Embedding uniform(-0.05, 0.05)
unless pretrained (star.py:113-127), glorot-uniform kernels (fans of the 3-d specific kernel include
the domain axis, as Keras computes them), zero biases, gamma one / beta zero; no regularisers, no
dropout.  Numerics: `TowerEngine(tower="star")` (csrc/star_kernels.hip).
"""
import numpy as np

from .deepctr import DeepCTR


def glorot_uniform(rs, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def initial_tensors(rs, n_user, n_item, n_domain, emb_dim, hidden, user_emb=None, item_emb=None):
    t = {}
    t["user_emb"] = user_emb if user_emb is not None else rs.uniform(-0.05, 0.05, (n_user, emb_dim)).astype(np.float32)
    t["item_emb"] = item_emb if item_emb is not None else rs.uniform(-0.05, 0.05, (n_item, emb_dim)).astype(np.float32)
    t["domain_emb"] = rs.uniform(-0.05, 0.05, (n_domain, emb_dim)).astype(np.float32)
    dims = (3 * emb_dim,) + tuple(hidden)
    t["pn_gamma_shared"] = np.ones(dims[0], np.float32)
    t["pn_beta_shared"] = np.zeros(dims[0], np.float32)
    t["pn_gamma_spec"] = np.ones((n_domain, dims[0]), np.float32)
    t["pn_beta_spec"] = np.zeros((n_domain, dims[0]), np.float32)
    for l in range(3):
        t["Wd%d" % l] = glorot_uniform(rs, (n_domain, dims[l], dims[l + 1]), dims[l] * n_domain, dims[l + 1] * n_domain)
        t["bd%d" % l] = np.zeros((n_domain, dims[l + 1]), np.float32)
        t["Ws%d" % l] = glorot_uniform(rs, (dims[l], dims[l + 1]), dims[l], dims[l + 1])
        t["bs%d" % l] = np.zeros(dims[l + 1], np.float32)
    t["wo"] = glorot_uniform(rs, (dims[3], 1), dims[3], 1)
    t["gb"] = np.zeros(1, np.float32)
    return t


def dense_initial_tensors(rs, n_user, n_item, n_domain, emb_dim, hidden, user_emb=None, item_emb=None):
    """star.py:84-86,95 with `dense: "dense"`: Keras Dense layers (glorot-uniform kernels, zero biases) on the concatenated
    embeddings, Dense(1, sigmoid) on top -- in the mlp tower's segment names (the output unit's kernel / bias = wo / gb)."""
    t = {}
    t["user_emb"] = user_emb if user_emb is not None else rs.uniform(-0.05, 0.05, (n_user, emb_dim)).astype(np.float32)
    t["item_emb"] = item_emb if item_emb is not None else rs.uniform(-0.05, 0.05, (n_item, emb_dim)).astype(np.float32)
    t["domain_emb"] = rs.uniform(-0.05, 0.05, (n_domain, emb_dim)).astype(np.float32)
    dims = (3 * emb_dim,) + tuple(hidden)
    for l in range(len(hidden)):
        t["W%d" % l] = glorot_uniform(rs, (dims[l], dims[l + 1]), dims[l], dims[l + 1])
        t["b%d" % l] = np.zeros(dims[l + 1], np.float32)
    t["wo"] = glorot_uniform(rs, (dims[-1], 1), dims[-1], 1)
    t["gb"] = np.zeros(1, np.float32)
    return t


class Star(DeepCTR):
    def plain_dnn(self):
        mc = self.model_config
        return mc.get("norm") == "none" and mc.get("dense") == "dense"

    def tower_kind(self):
        mc = self.model_config
        if mc.get("auxiliary_net"):
            raise NotImplementedError("auxiliary_net (model_zoo/Star/auxiliary_net.py) is not built")
        if self.plain_dnn():
            return "mlp"
        if mc.get("norm") != "pn" or mc.get("dense") != "star":
            raise NotImplementedError("Star with norm=%r dense=%r: built are the PartitionedNorm + StarFCN form of the BASELINE "
                                      "configs and the plain form (norm none, dense dense)" % (mc.get("norm"), mc.get("dense")))
        return "star"

    def engine_kwargs(self):
        # star.py:74-95 attaches no regulariser to any layer and has no dropout
        return dict(l2_emb=0.0, l2_linear=0.0) if self.plain_dnn() else {}

    def dropout_rate(self):
        return 0.0                  # (star.py builds no Dropout layer, whatever the config says)

    def build_model(self):
        eng = super(Star, self).build_model()
        if self.plain_dnn():
            # Keras names of star.py's layers (for the substring filters of maml.py:153-179): Embedding layers named after
            # their attribute, Dense layers numbered in creation order, the output unit last
            n = len(self.model_config["hidden_dim"])
            names = {"user_emb": "user_emb/embeddings", "item_emb": "item_emb/embeddings", "domain_emb": "domain_emb/embeddings",
                     "wo": "dense_%d/kernel" % n, "gb": "dense_%d/bias" % n}
            for l in range(n):
                stem = "dense" if l == 0 else "dense_%d" % l
                names["W%d" % l], names["b%d" % l] = stem + "/kernel", stem + "/bias"
            eng.keras_name = lambda segment: names.get(segment, segment)
        return eng

    def draw_initial_tensors(self):
        mc = self.model_config
        make = dense_initial_tensors if self.plain_dnn() else initial_tensors
        return make(self.init_rs, self.n_uid, self.n_pid, self.n_domain, mc["user_dim"],
                    tuple(mc["hidden_dim"]), self.pretrained[0], self.pretrained[1])
