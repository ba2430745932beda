"""MAMDR wrapper: Domain Negotiation + Domain Regularization (mirror of model_zoo/mamdr.py).

Per epoch (mamdr.py:41-166):
  DN  set model := theta; one pass per domain in shuffled order, no reset;
      theta += beta * (theta~ - theta)
  DR  for every query domain i (same order): sample `sample_num` support domains
      (+ i itself if add_query_domain); merged = theta (+|*) phi_i; for each support j:
      model := merged, pass over j, pass over i (capped by domain_regulation_step),
      phi_i += gamma * (theta~ - merged), merged recomputed  -- gamma = meta_learning_rate
      (`domain_meta_learning_rate` is dead in the reference)
  val with merged weights, early stop on the average val AUC, test with the best theta / phi.
phi_d starts as a fresh random initialisation of the whole model (mamdr.py:31-33), theta as
the model's own initial weights (mamdr.py:29).  The reference draws order and samples from
the unseeded global `random`; here they come from a `random.Random(dataset.seed)`.
"""
from .. import meta, parallel
from ..plan import EpochPlanner
from .specific_base_model import SpecificBase


class MAMDR(SpecificBase):
    def __init__(self, base_model):
        super(MAMDR, self).__init__(base_model)

    def train(self):
        print("Start MAMDR on model: {}".format(self.model_config["name"]))
        tc = self.train_config
        # (train.target_domain only selects the early-stopping metric here: mamdr.py:153-154)
        self._get_model_meta_parms()
        self.meta_weights = self._get_meta_weights()
        # one process per GPU: every rank draws all D initialisations (the streams stay aligned with the
        # single-process run) and keeps a slot for every phi; the DR queries and DN passes of an epoch are dealt by
        # longest-processing-time on the cost THAT epoch's sampled plan will execute, one all-reduce per epoch carries
        # the DN displacement (+ the displacement of the tensors outside theta / phi, Star tower), a phi whose owner
        # changes travels point to point (parallel.BalancedMAMDR); validation, test and finetune of domain d run on the
        # rank that holds the current phi_d.  train.dn_mode "replicated" (not in the reference's configs; default
        # "sharded"): every rank runs the whole DN chain, the reference's sequential update, DR stays sharded.
        rank, world = parallel.world()
        steps = [self.dataset.train_dataset[d]["n_step"] for d in range(self.n_domain)]
        phis = {}
        for domain_idx in range(self.n_domain):
            full = self.model.pack(self.base_model.draw_initial_tensors())
            phis[domain_idx] = full[self.model.meta_off:self.model.meta_off + self.model.n_meta].clone()
        self.balanced = parallel.BalancedMAMDR(self.model, meta, self.meta_weights, phis, steps,
                                               dn_mode=tc.get("dn_mode", "sharded"))
        self.domain_weights = self.balanced.phis
        self.model.optimizer_reset()
        planner = EpochPlanner(self.build_meta_sequence(), tc["sample_num"], tc["add_query_domain"],
                               tc["shuffle_sequence"], seed=self.dataset.seed)
        planner.rng = self.rng
        batch_variant = "batch" in self.model_config["name"]
        if tc["merged_method"] not in ("plus", "times"):
            raise ValueError("merged_method must be 'plus' or 'times', not: {}".format(tc["merged_method"]))
        self.trace = []
        # On the GPU the shuffles of a whole epoch are drawn by one C call and uploaded in one copy, and the NEXT epoch's
        # are drawn on a worker thread while the stream runs this one (plan.EpochShuffles: the same stream of seeds in the
        # same order as one draw per pass).  Only where nothing else draws from the shuffler between two epochs.
        from ..plan import EpochShuffles
        shuffles, next_plan = None, None
        dev = getattr(self.model, "device", None)
        if (dev is not None and getattr(dev, "type", "cpu") == "cuda" and self.shuffler.shuffle
                and not tc["finetune_every_epoch"] and not tc.get("meta_finetune_step")
                and getattr(self.shuffler.shuffle_fn, "__module__", "") == "mamdr_amd.engine"):
            shuffles = EpochShuffles(self.shuffler, dev)
        for epoch in range(tc["epoch"]):
            print("Epoch: {}".format(epoch), "-" * 30)
            plan = next_plan if next_plan is not None else planner.next_epoch()
            next_plan = None
            self.trace += self.balanced.epoch(plan, shuffles.prepare if shuffles is not None else None,
                                              shuffles if shuffles is not None else self.shuffler, self.batch_size,
                                              self.learning_rate, tc["meta_learning_rate"], tc["merged_method"],
                                              tc["domain_regulation_step"], batch_variant, tc["sample_num"],
                                              bool(tc["finetune_every_epoch"]))
            if shuffles is not None and epoch + 1 < tc["epoch"]:
                next_plan = planner.next_epoch()
                shuffles.prefetch(self.balanced.local_passes(next_plan, tc["domain_regulation_step"]))
            if plan["dr"]:                       # (meta_finetune_val: the model of the rank that ran the plan's last query)
                self.live_src = self.balanced.owner(plan["dr"][-1][0])
            if epoch % tc["val_every_step"] == 0:
                self.balanced.sync_tail()           # Star: one model outside theta / phi again before it is scored
                _, val_avg_auc, _, val_domain_auc = self.val()
                if self.early_stop_step(self._val_metric(val_avg_auc, val_domain_auc)):
                    break
                print("Test Result: ")
                self.val_and_test("test")
        if shuffles is not None:
            shuffles.cancel()               # a prefetched epoch that will not run gives its seeds back (the finetune stage draws next)
        self.balanced.sync_tail()
        self.balanced.sync_phis()           # every slot current on every rank once training is over
