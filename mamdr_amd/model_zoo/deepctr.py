"""DeepCTR tower -- host-side mirror of model_zoo/DeepCTR/deepctr.py.

`build_model` keeps the reference's substring registry (deepctr.py:24-50): names
containing `mlp` build the 3 x 128-d embedding -> DNN(hidden_dim) -> Dense(1) -> sigmoid
tower (deepctr.py:95-136) on the HIP engine; names containing `deepfm` add the linear
tables and the FM second-order term to the logit (deepctr.py:36-38);
`wdl` is the same without the FM term (deepctr.py:29-32); `nfm` (deepctr.py:33-35: linear tables + DNN over the
bi-interaction of the three fields) and `pnn` (deepctr.py:44-46: DNN over the fields and their pairwise inner
products) and `ccpm` (deepctr.py:41-43: convolutions over the field axis) run on the generic-layer engine
(`GraphEngine`, csrc/graph_engine.hip), and so does `autoint` (deepctr.py:37-40: three multi-head self-attention
layers over the fields beside the DNN).  Initial tensors follow the reference's initialisers (glorot normal for the
kernels, zeros for biases, N(0, 1e-4^2) for the domain table and for user/item tables
without pretraining, constants from the pretrained tables otherwise) drawn from a numpy
stream seeded with dataset.seed -- TF's own streams are not reproducible.
"""
import random

import numpy as np

from .base_model import BaseModel

GRAPH_TOWERS = ("nfm", "pnn", "ccpm", "autoint")


def glorot_normal(rs, fan_in, fan_out, shape):
    """Keras glorot_normal: truncated normal (2 sigma), stddev = sqrt(2 / (fan_in + fan_out))."""
    std = np.sqrt(2.0 / (fan_in + fan_out))
    x = rs.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rs.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def initial_tensors(rs, n_user, n_item, n_domain, emb_dim, hidden, user_emb=None, item_emb=None):
    """every initializer of the model, in layer order (what `init_layer` re-runs,
    specific_base_model.py:174-178)."""
    t = {}
    t["user_emb"] = user_emb if user_emb is not None else (rs.standard_normal((n_user, emb_dim)) * 1e-4).astype(np.float32)
    t["item_emb"] = item_emb if item_emb is not None else (rs.standard_normal((n_item, emb_dim)) * 1e-4).astype(np.float32)
    t["domain_emb"] = (rs.standard_normal((n_domain, emb_dim)) * 1e-4).astype(np.float32)
    dims = (3 * emb_dim,) + tuple(hidden)
    for l in range(len(hidden)):          # (the step kernels take exactly three layers; the generic-layer towers 1 - 4)
        t["W%d" % l] = glorot_normal(rs, dims[l], dims[l + 1], (dims[l], dims[l + 1]))
        t["b%d" % l] = np.zeros(dims[l + 1], np.float32)
    t["wo"] = glorot_normal(rs, dims[-1], 1, (dims[-1], 1))
    t["gb"] = np.zeros(1, np.float32)
    # DeepFM 1-d linear tables: Zeros initialiser (deepctr get_linear_logit); unused by the mlp tower
    t["lin_user"] = np.zeros(n_user, np.float32)
    t["lin_item"] = np.zeros(n_item, np.float32)
    t["lin_domain"] = np.zeros(n_domain, np.float32)
    # uncertainty weighting: `log_var` [D], Constant(1.) (uncertainty_weight/weighted_loss.py:23-28)
    t["log_var"] = np.ones(n_domain, np.float32)
    return t


class DeepCTR(BaseModel):
    def __init__(self, dataset, config, engine_factory=None):
        super(DeepCTR, self).__init__(dataset, config, engine_factory)

    def tower_kind(self):
        name = self.model_config["name"]
        if "mlp" in name:
            tower = "mlp"
        elif "wdl" in name:               # deepctr.py:29-32: linear tables + DNN (DeepFM without the FM term)
            tower = "wdl"
        elif "nfm" in name:               # deepctr.py:33-35
            tower = "nfm"
        elif "autoint" in name:           # deepctr.py:37-40
            tower = "autoint"
        elif "ccpm" in name:              # deepctr.py:41-43
            tower = "ccpm"
        elif "pnn" in name:               # deepctr.py:44-46
            tower = "pnn"
        elif "deepfm" in name:
            tower = "deepfm"
        else:
            raise ValueError("model: {} not found".format(name))
        return tower

    def build_model(self):
        tower = self.tower_kind()
        mc, tc = self.model_config, self.train_config
        if not (mc["user_dim"] == mc["item_dim"] == mc["domain_dim"]):
            raise ValueError("user_dim, item_dim and domain_dim must be equal")
        n_hidden = len(mc["hidden_dim"])
        if not 1 <= n_hidden <= 4:
            raise ValueError("hidden_dim %r: the '%s' tower takes 1 to 4 hidden layers" % (mc["hidden_dim"], tower))
        # mlp / wdl / deepfm with a hidden_dim other than the reference configs' [256, 128, 64] (deepctr.py:26-49 passes any
        # list through as dnn_hidden_units): the step kernels are built for that one shape, the generic-layer engine takes
        # 1 - 4 layers of widths that are multiples of 64 (MAMDR_GRAPH_MLP / WDL / DEEPFM, round 5)
        if tower == "star" and n_hidden != 3:
            raise ValueError("hidden_dim %r: the Star tower's kernels are built for three hidden layers (the reference's "
                             "configs all have [256, 128, 64])" % (mc["hidden_dim"],))
        self.graph_dnn = (tower not in GRAPH_TOWERS and tower != "star" and tuple(mc["hidden_dim"]) != (256, 128, 64))
        factory = self.engine_factory
        # PNN and NFM on the step kernels (round 4: MAMDR_TOWER_PNN = the mlp tower + the inner products' three rows of the
        # first kernel; MAMDR_TOWER_NFM = WDL's linear tables + the DNN on the bi-interaction in the domain field's place;
        # both on k_tower4's FM instances): batches of up to 2,048 rows and hidden_dim [256, 128, 64] -- every reference
        # config; anything else, and MAMDR_PNN_ENGINE / MAMDR_NFM_ENGINE=graph (the parity twins), runs them on the
        # generic-layer engine
        import os
        self.step_pnn = (tower in ("pnn", "nfm") and factory is None and self.batch_size <= 2048 and
                         tuple(mc["hidden_dim"]) == (256, 128, 64) and mc["user_dim"] == 128 and
                         os.environ.get("MAMDR_PNN_ENGINE", "step") != "graph" and
                         os.environ.get("MAMDR_%s_ENGINE" % tower.upper(), "step") != "graph")
        if self.step_pnn:
            from ..engine import TowerEngine
            factory = TowerEngine
        elif self.graph_dnn:
            if factory is None:       # (an injected factory -- the tests' CPU stand-in -- takes `hidden` itself)
                if any(h <= 0 or h % 64 for h in mc["hidden_dim"]):
                    raise ValueError("hidden_dim %r: layer widths must be multiples of 64" % (mc["hidden_dim"],))
                from ..graph_engine import GraphEngine
                factory = GraphEngine
            else:
                self.graph_dnn = False
        elif tower in GRAPH_TOWERS:       # generic-layer engine; an injected factory offers it as `.graph` (tests)
            if factory is not None:
                factory = getattr(factory, "graph", None)
                if factory is None:
                    raise NotImplementedError("the injected engine factory has no '%s' tower" % tower)
            else:
                from ..graph_engine import GraphEngine
                factory = GraphEngine
        elif factory is None:
            from ..engine import TowerEngine
            factory = TowerEngine
        kw = dict(self.engine_kwargs())
        if "uncertainty_weight" in mc["name"]:     # run.py:49-50: the weighted loss joins the compiled model
            kw["uncertainty_weight"] = True
        # deepctr.py:104-116: `trainable=emb_trainable` reaches SparseFeat only on the pretrained branch; without
        # pretrained tables the column is built with deepctr's default (trainable) WHATEVER emb_trainable says
        self.tables_trainable = bool(tc["emb_trainable"]) or not bool(tc["load_pretrain_emb"])
        if (tower in GRAPH_TOWERS and not self.step_pnn) or self.graph_dnn:
            eng = factory(tower, self.n_uid, self.n_pid, self.n_domain, self.batch_size, expert_hidden=tuple(mc["hidden_dim"]),
                          tower_hidden=(), dropout=self.dropout_rate(), emb_trainable=self.tables_trainable,
                          emb_dim=mc["user_dim"], **kw)
        else:
            eng = factory(self.n_uid, self.n_pid, self.n_domain, self.batch_size, dropout=self.dropout_rate(),
                          emb_trainable=self.tables_trainable, tower=tower, emb_dim=mc["user_dim"],
                          hidden=tuple(mc["hidden_dim"]), **kw)
        self.tower = tower
        self.init_rs = np.random.RandomState(self.dataset.seed)
        pre = bool(tc["load_pretrain_emb"])
        self.pretrained = (self.dataset.user_emb, self.dataset.item_emb) if pre else (None, None)
        if pre and self.pretrained[0] is None:
            raise ValueError("load_pretrain_emb is set but the dataset has no pretrained tables")
        tensors = self.draw_initial_tensors()
        if not self.tables_trainable:
            eng.bind_table("user_emb", tensors["user_emb"])
            eng.bind_table("item_emb", tensors["item_emb"])
        for split, store in (("train", self.dataset.train_dataset), ("val", self.dataset.val_dataset),
                             ("test", self.dataset.test_dataset)):
            for d, v in store.items():
                c = v["data"]
                eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
        eng.set_weights(eng.pack(tensors))
        self.optimizer = tc["optimizer"]
        eng.compile(self.optimizer)      # "adam" -> tf.train.AdamOptimizer(learning_rate); a Keras name otherwise (deepctr.py:54-57)
        if tc["loss"] != "binary_crossentropy":
            raise NotImplementedError("loss '%s': only binary_crossentropy is on the hot path" % tc["loss"])
        return eng

    def engine_kwargs(self):
        """extra keyword arguments of the engine (subclasses: Star's plain-DNN form has no regularisers)."""
        return {}

    def dropout_rate(self):
        return self.model_config.get("dropout", 0.0)

    def draw_initial_tensors(self):
        mc = self.model_config
        t = initial_tensors(self.init_rs, self.n_uid, self.n_pid, self.n_domain, mc["user_dim"],
                            tuple(mc["hidden_dim"]), self.pretrained[0], self.pretrained[1])
        tower = getattr(self, "tower", None)
        if tower in GRAPH_TOWERS:         # first kernel: NFM on the 128 interaction columns, PNN on the fields + 3 inner products,
            E, h0 = mc["user_dim"], mc["hidden_dim"][0]       # CCPM on the [128 x 4] convolution features
            in_dim = {"nfm": E, "pnn": 3 * E + 3, "ccpm": 4 * E, "autoint": 3 * E}[tower]
            t["W0"] = glorot_normal(self.init_rs, in_dim, h0, (in_dim, h0))
            if tower == "autoint":        # InteractingLayer: W_Query | W_key | W_Value | W_Res, TruncatedNormal(stddev 0.05) each
                d_in = E
                for l in range(3):
                    w = self.init_rs.standard_normal((d_in, 128))
                    bad = np.abs(w) > 2.0
                    while bad.any():
                        w[bad] = self.init_rs.standard_normal(int(bad.sum()))
                        bad = np.abs(w) > 2.0
                    t["att%d_w" % l] = (w * 0.05).astype(np.float32)
                    d_in = 32
                h_last = mc["hidden_dim"][-1]
                t["wo"] = glorot_normal(self.init_rs, 96 + h_last, 1, (96 + h_last, 1))
            if tower == "ccpm":           # Keras Conv2D defaults: glorot_uniform kernels [6,1,1,4] / [5,1,4,4] (centre tap kept), zero biases
                lim1, lim2 = np.sqrt(6.0 / (6 * 1 + 6 * 4)), np.sqrt(6.0 / (5 * 4 + 5 * 4))
                t["conv1_w"] = self.init_rs.uniform(-lim1, lim1, (6, 4)).astype(np.float32)
                t["conv1_b"] = np.zeros(4, np.float32)
                t["conv2_w"] = self.init_rs.uniform(-lim2, lim2, (4, 4)).astype(np.float32)
                t["conv2_b"] = np.zeros(4, np.float32)
        return t

    def train(self):
        """alternate ('joint') training, deepctr.py:63-93."""
        self.model.optimizer_reset()
        train_sequence = list(range(self.n_domain))
        rng = random.Random(self.dataset.seed)
        for epoch in range(self.train_config["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            rng.shuffle(train_sequence)
            for idx in train_sequence:
                self.fit_domain(idx, phase="alt")
            print("Val Result: ")
            avg_loss, avg_auc, domain_loss, domain_auc = self.val_and_test("val")
            if self.early_stop_step(avg_auc):
                break
            print("Test Result: ")
            # as in the reference, this reloads the best checkpoint into the live model
            self.val_and_test("test")
