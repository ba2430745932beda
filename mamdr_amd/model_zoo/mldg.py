"""MLDG wrapper -- mirror of model_zoo/mldg.py (a comparison baseline of the paper).

As the reference implements it (mldg.py:62-125): per domain the model is reset to theta; the meta-train pass
only accumulates gradients (dropout off, no inner optimiser step); the outer Adam moves the live model by the
accumulated gradient; the meta-val pass adds the gradients at the moved weights; the model is reset to theta
and the outer Adam applies the sum, giving the new theta (once per epoch for "batch" names).  Both passes run
the step kernels in accumulate mode; the outer Adam is `mamdr_adam_apply` (one optimiser, two applies per domain).
"""
from .. import meta
from .maml import MAML


class MLDG(MAML):
    def train(self):
        print("Start MLDG training on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        target = tc["target_domain"]
        avg = tc["average_meta_grad"]
        if avg == "mean" and tc["meta_train_step"] > 0:
            grad_scale = 1.0 / float(self.n_domain * tc["meta_train_step"])          # mldg.py:211-213
        else:
            # "moving_mean": K.moving_average_update(ag, g, 0.999) per batch (mldg.py:222-223); "drop": Dropout(0.2) on
            # the rank-1 gradients in a K.function that never feeds the learning phase = the identity (see maml.py)
            if avg == "moving_mean":
                self.model.set_moving_average(0.999)
            grad_scale = 1.0
        windows = self.build_meta_windows()
        self._get_model_meta_parms()
        meta_weights = self._get_meta_weights()
        self.model.optimizer_reset()
        outer = meta.OuterAdamState(self.model)
        acc = self.model.new_vector()
        self.model.bind_accumulator(acc)
        train_sequence = list(range(self.n_domain))
        batch_variant = "batch" in self.model_config["name"]
        self.trace = []
        for epoch in range(tc["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            self.rng.shuffle(train_sequence)
            # (the shuffle covers every domain, the target is skipped inside the loop: mldg.py:65-68)
            seq = [d for d in train_sequence if d != target]
            self.trace += meta.mldg_epoch(self.model, meta_weights, outer, acc, seq, self.shuffler,
                                          self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                          batch_variant, tc["meta_train_step"], grad_scale, windows, target)
            if target >= 0:         # mldg.py:127-131: the model (left at theta) takes a full pass over the target domain
                meta.run_pass(self.model, target, self.shuffler, self.batch_size, self.learning_rate, self.trace, "target")
            if epoch % tc["val_every_step"] == 0:
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                self.val_and_test("test")
                self._set_model_meta_parms(meta_weights)
