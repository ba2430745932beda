"""UncertaintyWeight wrapper -- mirror of model_zoo/uncertainty_weight/uncertainty_weight.py (a comparison
baseline of the paper).

The reference appends a `WeightedLoss` layer to the compiled model (weighted_loss.py:30-43): the training loss
of a batch of domain d is mean(BCE) / var_d^2 + log var_d (+ the tower's regularisers) with one trainable
scalar var_d per domain, initial value 1; it then trains with the plain alternate loop
(uncertainty_weight.py:62-93) and evaluates through the BASE model, i.e. unweighted.  Here the engine is
created with `uncertainty_weight=True` by the tower (the scalars are the `log_var` segment of the flat vector and
the step kernels scale d loss / d logit by 1 / var_d^2), so training is the base model's own alternate loop.
"""


class UncertaintyWeight(object):
    def __init__(self, base_model):
        self.base_model = base_model
        if "log_var" not in base_model.model.segments:
            if hasattr(base_model.model, "task_ranges"):
                # uncertainty_weight.py:41-45 takes `self.model.inputs` / `outputs[0]` of ONE compiled Keras model;
                # DeepMTLCTR keeps a dict of per-domain models (deep_mtl_ctr.py:51-67): nothing to wrap
                raise NotImplementedError("uncertainty weighting wraps a single-output tower; the multi-task towers are a "
                                          "dict of per-domain models (deep_mtl_ctr.py:51-67)")
            raise ValueError("the tower was not built with the weighted loss (model name lacks 'uncertainty_weight')")

    def __getattr__(self, item):
        return getattr(self.base_model, item)

    def train(self):
        return self.base_model.train()
