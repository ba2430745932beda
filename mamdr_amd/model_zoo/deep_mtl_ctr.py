"""DeepMTLCTR -- host-side mirror of model_zoo/DeepMTLCTR/deep_mtl_ctr.py (shared_bottom / mmoe / ple).

`build_model` keeps the reference's substring registry (deep_mtl_ctr.py:25-49): one binary task per domain on deepctr's
SharedBottom / MMOE / PLE, and, as the reference compiles `Model(inputs, outputs[d])` per domain on ONE shared Adam
optimizer (:53-67), every step on domain d trains the variables on the path to output d only -- the multi-task engine
(`GraphEngine`, csrc/graph_engine.hip) takes the domain with every call.  `train()` is the alternate loop of :69-96,
`val_and_test` scores domain d with domain d's model (:189-222), `separate_train_val_test` restarts every domain from the
same weights (:128-187): with init_parms every domain's model is compiled with the STRING 'adam' (:147-148) -- a fresh
Keras Adam per domain, lr 1e-3 and epsilon 1e-7 whatever `learning_rate` says, zero slots -- and for the finetune stage
with plain SGD at `learning_rate` (:143-146); Keras EarlyStopping(val_AUC, min_delta=1e-4) + best-only checkpoint.  Initial tensors: deepctr's initialisers (glorot normal DNN kernels, glorot uniform gate / head kernels, zero biases,
N(0, 1e-4^2) domain table, pretrained constants for the user / item tables) from a numpy stream seeded with dataset.seed.
"""
import random

import numpy as np

from .base_model import BaseModel
from .deepctr import glorot_normal

KINDS = ("shared_bottom", "mmoe", "ple")


def tensor_plan(kind, n_domain, emb_dim, expert_hidden, tower_hidden, gate_hidden, num_experts, shared_expert_num,
                specific_expert_num):
    """[(name, shape)] of every trainable tensor behind the tables, in the engine's flat-vector order."""
    xdim = 3 * emb_dim

    def dnn(name, in_dim, hidden):
        out, i = [], in_dim
        for l, h in enumerate(hidden):
            out += [("%s/W%d" % (name, l), (i, h)), ("%s/b%d" % (name, l), (h,))]
            i = h
        return out
    if kind == "shared_bottom":
        shared, specific = ["bottom"], 0
    elif kind == "mmoe":
        shared, specific = ["expert_%d" % e for e in range(num_experts)], 0
    else:
        shared, specific = ["shared_expert_%d" % e for e in range(shared_expert_num)], specific_expert_num
    t = [("domain_emb", (n_domain, emb_dim))]
    for e in shared:
        t += dnn(e, xdim, expert_hidden)
    for d in range(n_domain):
        for e in range(specific):
            t += dnn("task_%d_expert_%d" % (d, e), xdim, expert_hidden)
        if kind != "shared_bottom":
            t += dnn("gate_%d" % d, xdim, gate_hidden)
            t.append(("gate_%d/Wg" % d, (gate_hidden[-1], specific + len(shared))))
        t += dnn("tower_%d" % d, expert_hidden[-1], tower_hidden)
        t += [("head_%d/w" % d, (tower_hidden[-1], 1)), ("head_%d/gb" % d, (1,))]
    return t


def initial_tensors(rs, plan, emb_dim):
    t = {}
    for name, shape in plan:
        if name == "domain_emb":
            t[name] = (rs.standard_normal(shape) * 1e-4).astype(np.float32)        # deepctr SparseFeat default
        elif name.endswith("/Wg") or (name.startswith("head_") and name.endswith("/w")):
            # the gate's softmax kernel and the per-task output unit are plain `tf.keras.layers.Dense(..)` in deepctr
            # 0.9.0's sharedbottom.py / mmoe.py / ple.py [dep]: Keras' default glorot_uniform, not the DNN's glorot_normal
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            t[name] = rs.uniform(-lim, lim, shape).astype(np.float32)
        elif len(shape) == 2:
            t[name] = glorot_normal(rs, shape[0], shape[1], shape)                 # DNN kernels (experts, gate DNN, towers)
        else:
            t[name] = np.zeros(shape, np.float32)                                  # biases, PredictionLayer global_bias
    return t


class DeepMTLCTR(BaseModel):
    def __init__(self, dataset, config, engine_factory=None):
        super(DeepMTLCTR, self).__init__(dataset, config, engine_factory)

    def tower_kind(self):
        name = self.model_config["name"]
        for k in KINDS:                       # deep_mtl_ctr.py:25,31,39: substring tests in this order
            if k in name:
                return k
        raise ValueError("model: {} not found".format(name))

    def build_model(self):
        kind = self.tower_kind()
        mc, tc = self.model_config, self.train_config
        if not (mc["user_dim"] == mc["item_dim"] == mc["domain_dim"]):
            raise ValueError("user_dim, item_dim and domain_dim must be equal")
        if kind == "ple" and mc.get("num_levels", 1) != 1:
            raise NotImplementedError("ple with num_levels = %r: the reference's configs all use one level" % mc.get("num_levels"))
        factory = self.engine_factory
        if factory is None:
            from ..graph_engine import GraphEngine
            factory = GraphEngine
        self.tables_trainable = bool(tc["emb_trainable"]) or not bool(tc["load_pretrain_emb"])
        gate = tuple(mc.get("gate_dnn_hidden_units", ())) if kind != "shared_bottom" else ()
        shape = dict(expert_hidden=tuple(mc["hidden_dim"]), tower_hidden=tuple(mc["tower_hidden_dim"]), gate_hidden=gate,
                     num_experts=int(mc.get("num_experts", 0)), shared_expert_num=int(mc.get("shared_expert_num", 0)),
                     specific_expert_num=int(mc.get("specific_expert_num", 0)))
        eng = factory(kind, self.n_uid, self.n_pid, self.n_domain, self.batch_size, dropout=mc.get("dropout", 0.0),
                      emb_trainable=self.tables_trainable, emb_dim=mc["user_dim"], **shape)
        self.init_rs = np.random.RandomState(self.dataset.seed)
        pre = bool(tc["load_pretrain_emb"])
        if pre and self.dataset.user_emb is None:
            raise ValueError("load_pretrain_emb is set but the dataset has no pretrained tables")
        self.plan = tensor_plan(kind, self.n_domain, mc["user_dim"], **shape)
        tensors = initial_tensors(self.init_rs, self.plan, mc["user_dim"])
        # deep_mtl_ctr.py:108-121: pretrained constants when load_pretrain_emb, deepctr's N(0, 1e-4^2) otherwise
        E = mc["user_dim"]
        user = self.dataset.user_emb if pre else (self.init_rs.standard_normal((self.n_uid, E)) * 1e-4).astype(np.float32)
        item = self.dataset.item_emb if pre else (self.init_rs.standard_normal((self.n_pid, E)) * 1e-4).astype(np.float32)
        if self.tables_trainable:
            tensors["user_emb"], tensors["item_emb"] = user, item
            self.plan = [("user_emb", (self.n_uid, E)), ("item_emb", (self.n_pid, E))] + self.plan
        else:
            eng.bind_table("user_emb", user)
            eng.bind_table("item_emb", item)
        for split, store in (("train", self.dataset.train_dataset), ("val", self.dataset.val_dataset),
                             ("test", self.dataset.test_dataset)):
            for d, v in store.items():
                c = v["data"]
                eng.bind_domain_data(d, split, c["uid"], c["pid"], c["domain"], c["label"])
        eng.set_weights(eng.pack(tensors))
        eng.compile(tc["optimizer"])     # "adam" -> tf.train.AdamOptimizer(learning_rate); a Keras name otherwise (deep_mtl_ctr.py:53-56)
        if tc["loss"] != "binary_crossentropy":
            raise NotImplementedError("loss '%s': only binary_crossentropy is built" % tc["loss"])
        return eng

    def separate_train_val_test(self, init_parms=True):
        """deep_mtl_ctr.py:128-187.  init_parms: `compile(optimizer=self.train_config['optimizer'])` with the string
        'adam' builds a NEW Keras Adam for every domain's model -- learning rate 1e-3 (not `learning_rate`), epsilon
        K.epsilon() = 1e-7, empty slots; the shared tf.train.AdamOptimizer of the alternate loop is not involved.
        Finetune (init_parms false): SGD at `learning_rate`, as BaseModel."""
        if not init_parms:
            return BaseModel.separate_train_val_test(self, init_parms=False)
        from .. import parallel
        if parallel.world()[1] > 1:
            raise NotImplementedError("<%s>_separate under several processes" % self.tower_kind())
        weights = self.model.get_weights()
        self.model.set_adam_eps(1e-7)
        try:
            return self._finetune_domains(lambda d: weights, "adam", 1e-3, per_domain_reset=True)
        finally:
            self.model.set_adam_eps(1e-8)
            self.model.optimizer_reset()

    def train(self):
        """deep_mtl_ctr.py:69-96: per epoch one full pass per domain through that domain's model, shuffled order."""
        self.model.optimizer_reset()
        train_sequence = list(range(self.n_domain))
        rng = random.Random(self.dataset.seed)
        for epoch in range(self.train_config["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            rng.shuffle(train_sequence)
            for idx in train_sequence:
                print("Train on: Domain {}".format(idx))
                self.fit_domain(idx, phase="alt")
            print("Val Result: ")
            avg_loss, avg_auc, domain_loss, domain_auc = self.val_and_test("val")
            if self.early_stop_step(avg_auc):
                break
            print("Test Result: ")
            self.val_and_test("test")
