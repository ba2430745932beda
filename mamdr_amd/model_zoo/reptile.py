"""Reptile wrapper (mirror of model_zoo/reptile.py).

Per epoch (reptile.py:40-125): shuffle all domains; for each, reset the model to theta,
run one pass, then either theta += beta * (theta~ - theta) immediately, or (names
containing "batch") accumulate theta~ - theta and apply the sum once per epoch.
"""
from .. import meta, parallel
from .maml import MAML


class Reptile(MAML):
    def __init__(self, base_model):
        super(Reptile, self).__init__(base_model)

    def train(self):
        print("Start reptile on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        self._get_model_meta_parms()
        meta_weights = self._get_meta_weights()
        self.model.optimizer_reset()
        train_sequence = list(range(self.n_domain))
        batch_variant = "batch" in self.model_config["name"]
        rank, world = parallel.world()
        if world > 1:
            sizes = [self.dataset.train_dataset[d]["n_data"] for d in range(self.n_domain)]
            owner = parallel.lpt_partition(sizes, world)
            acc = self.model.new_vector(meta=True)
            zero = self.model.new_vector(meta=True)
        self.trace = []
        for epoch in range(tc["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            self.rng.shuffle(train_sequence)
            if world > 1 and batch_variant:
                # one process per GPU: the batch variant's sum of displacements is a sum over ranks
                self.trace += parallel.reptile_batch_epoch_sharded(
                    self.model, meta, meta_weights, [d for d in train_sequence if owner[d] == rank], self.shuffler,
                    self.batch_size, self.learning_rate, tc["meta_learning_rate"], acc, tc["meta_train_step"],
                    target=tc["target_domain"])
            elif world > 1:
                # per-domain variant: every rank runs the recurrence over its domains, displacements summed
                self.trace += parallel.reptile_epoch_sharded(
                    self.model, meta, meta_weights, [d for d in train_sequence if owner[d] == rank], self.shuffler,
                    self.batch_size, self.learning_rate, tc["meta_learning_rate"], acc, zero, tc["meta_train_step"],
                    target=tc["target_domain"])
            if world > 1 and tc["target_domain"] >= 0:
                # reptile.py:98-102: the epoch ends with a full pass of the model (theta, identical on every rank) over
                # the target domain
                meta.run_pass(self.model, tc["target_domain"], self.shuffler, self.batch_size, self.learning_rate, self.trace,
                              "target")
                # ... whose result differs per rank (own shuffle stream, own Adam slots): rank 0's is everybody's
                parallel.broadcast_live(self.model, src=0)
            if world == 1:
                self.trace += meta.reptile_epoch(self.model, meta_weights, list(train_sequence), self.shuffler,
                                                 self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                                 batch_variant, tc["meta_train_step"], target=tc["target_domain"])
            if epoch % tc["val_every_step"] == 0:
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                self.val_and_test("test")
                self._set_model_meta_parms(meta_weights)
