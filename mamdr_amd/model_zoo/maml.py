"""MAML base wrapper -- the helpers every meta-learning wrapper inherits.

Mirror of model_zoo/maml.py for what the hot path uses: delegation to the wrapped
tower (maml.py:27-33), meta-parameter selection (maml.py:153-179), bulk weight read /
write (maml.py:181-194) and `val()` (maml.py:343-353).  In the reference these move
numpy copies through the host; here `_get_meta_weights` returns a device snapshot of
the flat trainable vector and `_set_model_meta_parms` is a device copy.

The first-order MAML loop itself (maml.py:35-151,196-243: meta-gradient accumulation
on a second data split + an outer Adam) is the "second wave" of SURVEY.md section 8
(a8b): named by the north star, used by no BASELINE config, not built in this round.
"""
import random


class MAML(object):
    def __init__(self, base_model):
        self.base_model = base_model
        self.rng = random.Random(base_model.dataset.seed)   # the reference uses the unseeded global RNG

    def __getattr__(self, item):
        return getattr(self.base_model, item)

    # ------------------------------------------------------------------ meta parameters
    def _get_model_meta_parms(self):
        """maml.py:153-179.  `["all"]` = every trainable weight = the flat vector.  Name
        filters select segments of it; all BASELINE MLP/DeepFM configs use ["all"]."""
        names = self.train_config["meta_parms"]
        if names[0] == "all":
            self.model_meta_parms = list(self.model.segments.keys())
        elif names[0] == "all_hidden":
            self.model_meta_parms = [s for s in self.model.segments if "emb" not in s]
        else:
            chosen = []
            for name in names:
                hit = [s for s in self.model.segments if name in s]
                if not hit:
                    raise ValueError("meta parms: {} not found in the model".format(name))
                chosen += hit
            self.model_meta_parms = chosen
        if set(self.model_meta_parms) != set(self.model.segments.keys()):
            raise NotImplementedError("meta_parms subsets (%s) are not built in this round: only ['all']"
                                      % (self.train_config["meta_parms"],))

    def _set_model_meta_parms(self, meta_weights):
        self.model.set_weights(meta_weights)

    def _get_meta_weights(self):
        return self.model.get_weights()

    # ------------------------------------------------------------------ validation
    def val(self):
        if self.train_config["meta_finetune_step"] > 0:
            raise NotImplementedError("meta_finetune_step > 0 (maml.py:245-287) is not built in this round")
        print("Val Result: ")
        return self.val_and_test("val")

    def _val_metric(self, val_avg_auc, val_domain_auc):
        t = self.train_config["target_domain"]
        return val_domain_auc[t] if t >= 0 else val_avg_auc

    def train(self):
        raise NotImplementedError("the first-order MAML loop (maml.py:35-151) is second-wave scope "
                                  "(SURVEY.md section 8 a8b) and is not built in this round")
