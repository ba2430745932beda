"""Domain Negotiation wrapper (mirror of model_zoo/domain_negotiation.py).

Per epoch (domain_negotiation.py:37-116): shuffle the meta sequence, set the model to
theta once, run one pass per domain WITHOUT resetting the weights in between, then
theta += beta * (theta~ - theta) as one elementwise kernel, validate, early-stop, test.
"""
from .. import meta, parallel
from .maml import MAML


class DomainNegotiation(MAML):
    def __init__(self, base_model):
        super(DomainNegotiation, self).__init__(base_model)

    def build_meta_sequence(self):
        """domain_negotiation.py:125-146: all domains except the target; an explicit list
        in train.meta_sequence is honoured (the string "random" is not a list -> ignored)."""
        t = self.train_config["target_domain"]
        seq = [idx for idx in self.dataset.train_dataset if not (t >= 0 and idx == t)]
        ms = self.train_config.get("meta_sequence")
        if isinstance(ms, list):
            if len(ms) != len(seq):
                raise ValueError("All the domains must be given in the sequence")
            seq = list(ms)
        return seq

    def train(self):
        print("Start Domain Negotiation on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        target = tc["target_domain"]
        self._get_model_meta_parms()
        meta_weights = self._get_meta_weights()
        self.model.optimizer_reset()
        meta_sequence = self.build_meta_sequence()
        # one process per GPU: every rank runs its own sub-sequence from the same theta, one
        # all-reduce of the displacements per epoch (parallel.dn_phase_sharded)
        rank, world = parallel.world()
        sizes = [self.dataset.train_dataset[d]["n_data"] for d in range(self.n_domain)]
        owner = parallel.lpt_partition(sizes, world)
        delta, zero = (self.model.new_vector(meta=True), self.model.new_vector(meta=True)) if world > 1 else (None, None)
        self.trace = []
        for epoch in range(tc["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            if tc["shuffle_sequence"]:
                self.rng.shuffle(meta_sequence)
            if world > 1:
                parallel.dn_phase_sharded(self.model, meta, meta_weights, [d for d in meta_sequence if owner[d] == rank],
                                          self.shuffler, self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                          self.trace, delta, zero, tc["meta_train_step"], target)
                self.model.assign_meta(meta_weights)
                if target >= 0:        # domain_negotiation.py:89-93: the model (not theta) takes one more pass over the target
                    meta.run_pass(self.model, target, self.shuffler, self.batch_size, self.learning_rate, self.trace,
                                  "target")
                    # (per-rank shuffle streams and Adam slots: rank 0's model is everybody's from here on)
                    parallel.broadcast_live(self.model, src=0)
            else:
                self.trace += meta.dn_epoch(self.model, meta_weights, list(meta_sequence), self.shuffler,
                                            self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                            tc["meta_train_step"], target)
            if epoch % tc["val_every_step"] == 0:
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                self.val_and_test("test")
                self._set_model_meta_parms(meta_weights)    # val_and_test("test") loaded the best weights
