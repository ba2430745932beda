"""PCGrad wrapper -- mirror of model_zoo/pcgrad.py (a comparison baseline of the paper).

Per epoch (pcgrad.py:62-124): shuffle the domains; for each query domain accumulate the gradient of its train
pass at the CURRENT model weights (no reset to a stored theta), sample `sample_num` auxiliary domains, project
each one's pass-gradient onto the running gradient (`mamdr_pcgrad_project`, bit-identical to the reference's
numpy) and take one outer-Adam step (lr = meta_learning_rate) of the model with the result.  The passes run
the step kernels in accumulate mode (dropout off, no update).
"""
from .. import meta
from .maml import MAML


class PCGrad(MAML):
    def train(self):
        print("Start PCGrad training on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        target = tc["target_domain"]      # skipped as a query domain (pcgrad.py:67-68), still a candidate auxiliary
        avg = tc["average_meta_grad"]
        if avg == "mean" and tc["meta_train_step"] > 0:
            grad_scale = 1.0 / float(self.n_domain * tc["meta_train_step"])
        else:
            # "moving_mean": K.moving_average_update(ag, g, 0.999) per batch (pcgrad.py:229-230); "drop": the identity
            # (Dropout(0.2) in a K.function that never feeds the learning phase, see maml.py)
            if avg == "moving_mean":
                self.model.set_moving_average(0.999)
            grad_scale = 1.0
        windows = self.build_meta_windows()
        self._get_model_meta_parms()
        self.model.optimizer_reset()
        outer = meta.OuterAdamState(self.model)
        cur, aux = self.model.new_vector(), self.model.new_vector()
        train_sequence = list(range(self.n_domain))
        self.trace = []
        for epoch in range(tc["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            self.rng.shuffle(train_sequence)
            aux_plan = {}
            for idx in train_sequence:             # pcgrad.py:112-115
                if idx == target:
                    continue
                cand = list(train_sequence)
                cand.remove(idx)
                aux_plan[idx] = self.rng.sample(cand, k=min(tc["sample_num"], len(cand)))
            self.trace += meta.pcgrad_epoch(self.model, outer, cur, aux, [d for d in train_sequence if d != target], aux_plan,
                                            self.shuffler,
                                            self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                            tc["meta_train_step"], grad_scale, windows)
            if epoch % tc["val_every_step"] == 0:
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                weights = self.model.get_weights()
                self.val_and_test("test")          # loads the best checkpoint
                self.model.set_weights(weights)
