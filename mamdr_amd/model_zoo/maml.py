"""MAML base wrapper -- the helpers every meta-learning wrapper inherits.

Mirror of model_zoo/maml.py for what the hot path uses: delegation to the wrapped
tower (maml.py:27-33), meta-parameter selection (maml.py:153-179), bulk weight read /
write (maml.py:181-194) and `val()` (maml.py:343-353).  In the reference these move
numpy copies through the host; here `_get_meta_weights` returns a device snapshot of
the flat trainable vector and `_set_model_meta_parms` is a device copy.

`train()` is the first-order MAML loop itself (maml.py:35-151,196-243): the meta pass runs the
step kernels in accumulate mode (no update, dropout off) and the outer tf.train.AdamOptimizer
is `mamdr_adam_apply` on the flat vector with its own slots.
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
            # maml.py:160-166: every variable whose NAME lacks "emb" (deepctr names the linear tables "linear...sparse_emb_*" too)
            self.model_meta_parms = [s for s in self.model.segments if "emb" not in self.model.keras_name(s)]
        else:
            chosen = []
            for name in names:
                hit = [s for s in self.model.segments if name in self.model.keras_name(s) and s not in chosen]
                if not hit:
                    raise ValueError("meta parms: {} not found in the model".format(name))
                chosen += hit
            self.model_meta_parms = chosen
        # theta / phi cover ONE contiguous range of the flat vector, from the first selected tensor to the last: everything
        # ("all"), the reference's Star filter (tables, kernel_shared, bias_shared; config/Taobao-10/star_taobao.json:37-41:
        # the prefix the Star block is laid out for), "all_hidden" (everything behind the embedding tables), any name list
        # of neighbouring tensors.  A list that skips tensors in between (maml.py:167-177 allows any) keeps them as HOLES
        # of the range: the outer updates run over them too, `assign_meta` never lets those slots reach the model, so
        # the skipped tensors train on undisturbed -- as variables outside `model_meta_parms` do in the reference.
        segs = self.model.segments
        chosen = sorted(self.model_meta_parms, key=lambda n: segs[n][0])
        lo, hi = segs[chosen[0]][0], segs[chosen[0]][0] + segs[chosen[0]][1]
        holes = []
        for n in chosen[1:]:
            off, cnt = segs[n]
            # (a gap that holds no other tensor is layout: alignment padding, or the rows of the step kernels' W0 that the
            # NFM tower leaves unused -- zeros that the outer updates keep at zero)
            between = sorted((o, c) for k, (o, c) in segs.items() if hi <= o < off)
            if between:
                holes.append((between[0][0] - lo, between[-1][0] + between[-1][1] - between[0][0]))
            hi = off + cnt
        if lo == 0 and hi >= self.model.n_params - 3:
            hi = self.model.n_params
        if lo == 0 and not holes and getattr(self.model, "tower", "") == "star" and hi == self.model.n_meta:
            return                                  # the engine's own prefix (its default)
        if holes and not hasattr(self.model, "assign_meta"):
            raise NotImplementedError("meta_parms %s select tensors that are not neighbours in the flat vector; this engine "
                                      "has no assign_meta" % (self.train_config["meta_parms"],))
        if holes:
            from .. import parallel
            if parallel.world()[1] > 1:     # the skipped tensors would train apart on every rank (the all-reduce covers theta)
                raise NotImplementedError("meta_parms %s skip tensors of the flat vector: not built for several processes"
                                          % (self.train_config["meta_parms"],))
            self.model.set_meta_range(lo, hi - lo, holes)
        else:
            self.model.set_meta_range(lo, hi - lo)

    def _set_model_meta_parms(self, meta_weights):
        self.model.assign_meta(meta_weights)

    def _get_meta_weights(self):
        return self.model.meta_weights.clone()

    # ------------------------------------------------------------------ validation
    def val(self):
        print("Val Result: ")
        if self.train_config["meta_finetune_step"] > 0:
            return self.meta_finetune_val()
        return self.val_and_test("val")

    def meta_finetune_val(self):
        """maml.py:245-287: every domain is evaluated after `meta_finetune_step` full passes over its own train
        split, each starting from the current model weights (the optimiser slots are NOT restored between
        domains: Keras' set_weights leaves them alone); the model is put back afterwards."""
        from .. import meta, parallel
        rank, world = parallel.world()
        weights = self.model.get_weights().clone()
        if world > 1:
            # several processes: every rank starts from the SAME weights -- the ones the single-process loop would hold
            # here: theta for the wrappers that leave theta in the model, the model of the rank that ran the plan's last
            # pass otherwise (`live_src`, set by the sharded MAMDR loop) -- and the domains are dealt round-robin
            parallel.broadcast(weights, int(getattr(self, "live_src", 0)))
        aux = getattr(self.model, "aux", None)      # Star: PartitionedNorm moving statistics move while training
        aux = aux.clone() if aux is not None else None
        domain_loss, domain_auc = {}, {}
        for i, idx in enumerate(self.dataset.train_dataset):
            if world > 1 and i % world != rank:
                continue
            self.model.set_weights(weights)
            print("Finetune on domain: {}".format(idx))
            for _ in range(self.train_config["meta_finetune_step"]):
                meta.run_pass(self.model, idx, self.shuffler, self.batch_size, self.learning_rate, self.trace, "meta_finetune")
            p_loss, p_auc = self.evaluate_domain(idx, "val")
            domain_loss[idx], domain_auc[idx] = float(p_loss), float(p_auc)
        self.model.set_weights(weights)
        if aux is not None:
            self.model.aux.copy_(aux)
        if world > 1:
            local = {d: (domain_loss[d], domain_auc[d]) for d in domain_loss}
            domain_loss, domain_auc = parallel.gather_domain_scalars(local, self.n_domain, self.model.device)
        return self._summarise("val", domain_loss, domain_auc)

    def build_meta_windows(self):
        """maml.py:289-341 / mldg.py:296-341 `build_meta_data_split`: "meta-train/val" takes the first
        int(n * meta_split_ratio) rows (file order) as the meta-train set and the rest as the meta-val set,
        each shuffled on its own; "meta-train/val-no-exclusive" shuffles first and takes / skips afterwards; any
        other value is the train-train split (both iterators over the whole train set) -> None."""
        tc = self.train_config
        if tc["meta_split"] not in ("meta-train/val", "meta-train/val-no-exclusive"):
            return None
        # the non-exclusive variant shuffles the whole split first and then takes / skips (maml.py:316-323):
        # window tag "stream" (plan.PassShuffler)
        tag = ("stream",) if tc["meta_split"] == "meta-train/val-no-exclusive" else ()
        windows = {}
        for d, v in self.dataset.train_dataset.items():
            n = v["n_data"]
            n_train = int(n * tc["meta_split_ratio"])
            if n_train <= 0 or n_train >= n:
                raise ValueError("domain %s: meta_split_ratio %s leaves an empty meta-train or meta-val set"
                                 % (d, tc["meta_split_ratio"]))
            windows[d] = ((0, n_train) + tag, (n_train, n) + tag)
        return windows

    def _val_metric(self, val_avg_auc, val_domain_auc):
        t = self.train_config["target_domain"]
        return val_domain_auc[t] if t >= 0 else val_avg_auc

    def train(self):
        """first-order MAML (maml.py:35-151): per domain reset to theta, inner Adam pass on the train
        iterator, meta pass accumulating d total_loss / d theta at the adapted weights, outer Adam
        (lr = meta_learning_rate) on theta -- per domain, or once per epoch for "batch" names."""
        from .. import meta
        print("Start MAML training on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        target = tc["target_domain"]
        windows = self.build_meta_windows()
        avg = tc["average_meta_grad"]
        if avg == "mean" and tc["meta_train_step"] > 0:
            grad_scale = 1.0 / float(self.n_domain * tc["meta_train_step"])          # maml.py:208-210
        else:
            if avg == "moving_mean":      # maml.py:219-220: K.moving_average_update(ag, g, 0.999) per meta batch
                self.model.set_moving_average(0.999)
            # "drop" (maml.py:220-226) passes the rank-1 gradients through layers.Dropout(0.2) inside a K.function
            # whose inputs do not include the learning phase (maml.py:231-234): the phase keeps its default, False,
            # and the layer is the identity -- the accumulation is the plain sum, exactly as for "none"
            grad_scale = 1.0
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
            # (the shuffle covers every domain, the target is skipped inside the loop: maml.py:65-68)
            seq = [d for d in train_sequence if d != target]
            self.trace += meta.maml_epoch(self.model, meta_weights, outer, acc, seq, self.shuffler,
                                          self.batch_size, self.learning_rate, tc["meta_learning_rate"],
                                          batch_variant, tc["meta_train_step"], grad_scale, windows, target)
            if target >= 0:         # maml.py:124-128: the model (left at theta) takes a full pass over the target domain
                meta.run_pass(self.model, target, self.shuffler, self.batch_size, self.learning_rate, self.trace, "target")
            if epoch % tc["val_every_step"] == 0:
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                self.val_and_test("test")
                self._set_model_meta_parms(meta_weights)
