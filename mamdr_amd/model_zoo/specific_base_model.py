"""SpecificBase -- shared theta + per-domain phi_d (mirror of model_zoo/specific_base_model.py).

Evaluation uses the merged weights theta (+|*) phi_d per domain
(specific_base_model.py:64-97); the best theta / phi_d are kept as device copies instead of
host deep copies (specific_base_model.py:44-62); the finetune stage restarts every domain from
its best merged weights with plain SGD, lr 0.001 hard-coded (specific_base_model.py:99-162,
:120 -- the reference's comment says Adam, the code says GradientDescent).
"""
from .. import parallel
from .maml import MAML

FINETUNE_SGD_LR = 0.001     # specific_base_model.py:120


class SpecificBase(MAML):
    def __init__(self, base_model):
        super(SpecificBase, self).__init__(base_model)
        self.meta_weights = None
        self.domain_weights = {}
        self.best_shared_weights = None
        self.best_domain_weights = None
        self.best_where = None

    def build_meta_sequence(self):
        t = self.train_config["target_domain"]
        seq = [idx for idx in self.dataset.train_dataset if not (t >= 0 and idx == t)]
        ms = self.train_config.get("meta_sequence")
        if isinstance(ms, list):
            if len(ms) != len(seq):
                raise ValueError("All the domains must be given in the sequence")
            seq = list(ms)
        return seq

    def _merge_weights(self, shared_weights, specific_weights, out=None):
        method = self.train_config["merged_method"]
        if method not in ("plus", "times"):
            raise ValueError("merged_method must be 'plus' or 'times', not: {}".format(method))
        if out is None:
            out = self.model.new_vector(meta=True)
        self.model.merge(out, shared_weights, specific_weights, method)
        return out

    def _owner(self, d, i, best=False):
        """rank that evaluates / finetunes domain d (the i-th of its split): the holder of the current phi_d -- of the
        best phi_d for the test / finetune stages -- under the sharded MAMDR loop, round-robin where every rank holds
        the same values."""
        rank, world = parallel.world()
        if world == 1:
            return 0
        where = self.best_where if best else getattr(getattr(self, "balanced", None), "where", None)
        w = where.get(d) if where else None
        return i % world if w is None else w

    def _snapshot_best(self):
        self.best_shared_weights = self.meta_weights.clone()
        # (a rank's copy of a slot it does not hold is stale and never read: best_where says who holds which)
        self.best_domain_weights = {d: w.clone() for d, w in self.domain_weights.items()}
        bal = getattr(self, "balanced", None)
        self.best_where = dict(bal.where) if bal is not None else None

    def early_stop_step(self, metric):
        base = self.base_model
        if base.best_metric is None:
            base.best_metric = metric
            self._snapshot_best()
            self.save_model(self.checkpoint_path)
        elif metric <= base.best_metric:
            base.counter += 1
            print("EarlyStopping counter: {} out of {}, Best AUC: {}".format(base.counter, base.patience,
                                                                           base.best_metric))
            if base.counter >= base.patience:
                base.early_stop = True
        else:
            self.save_model(self.checkpoint_path)
            base.best_metric = metric
            self._snapshot_best()
            base.counter = 0
        return base.early_stop

    def val_and_test(self, mode):
        if mode == "val":
            shared, specific = self.meta_weights, self.domain_weights
            store = self.dataset.val_dataset
        elif mode == "test":
            # specific_base_model.py:71: the best checkpoint comes back first -- for the Star tower that restores
            # the tensors outside theta / phi (PartitionedNorm gamma / beta and moving statistics, the specific
            # kernels, the output unit), and training goes on from them afterwards, as in the reference
            self.load_model(self.checkpoint_path)
            bal = getattr(self, "balanced", None)
            if bal is not None:                        # (the restored tail is the same on every rank: the new common value)
                bal.tail.rebase()
            shared, specific = self.best_shared_weights, self.best_domain_weights
            store = self.dataset.test_dataset
        else:
            raise ValueError("Mode can be either val or test, not: {}".format(mode))
        domain_loss, domain_auc = {}, {}
        merged = self.model.new_vector(meta=True)
        rank, world = parallel.world()
        for i, idx in enumerate(store):
            if world > 1 and self._owner(idx, i, best=(mode == "test")) != rank:
                continue                               # the holder of phi_idx scores the domain
            self._set_model_meta_parms(self._merge_weights(shared, specific[idx], out=merged))
            p_loss, p_auc = self.evaluate_domain(idx, mode)
            domain_loss[idx], domain_auc[idx] = float(p_loss), float(p_auc)
        if world > 1:                                  # owners evaluated; every rank gets every scalar
            local = {d: (domain_loss[d], domain_auc[d]) for d in domain_loss}
            domain_loss, domain_auc = parallel.gather_domain_scalars(local, self.n_domain, self.model.device)
        return self.base_model._summarise(mode, domain_loss, domain_auc)

    def separate_train_val_test(self, init_parms=True):
        if init_parms:
            return self.base_model.separate_train_val_test(init_parms=True)
        merged = self.model.new_vector(meta=True)

        def start(d):
            return self._merge_weights(self.best_shared_weights, self.best_domain_weights[d], out=merged)
        rank, world = parallel.world()
        if world > 1:                                  # the holder of the best phi_d finetunes domain d
            mine = [d for i, d in enumerate(self.dataset.train_dataset) if self._owner(d, i, best=True) == rank]
            _, _, dl, da = self.base_model._finetune_domains(start, "sgd", FINETUNE_SGD_LR, domains=mine, summarise=False)
            dl, da = parallel.gather_domain_scalars({d: (dl[d], da[d]) for d in dl}, self.n_domain, self.model.device)
            return self.base_model._summarise("test", dl, da)
        return self.base_model._finetune_domains(start, "sgd", FINETUNE_SGD_LR)
