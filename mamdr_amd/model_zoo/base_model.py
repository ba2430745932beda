"""BaseModel -- host-side mirror of the reference's model_zoo/base_model.py.

Keeps the control flow and the files a user of the reference relies on:
checkpoint / result paths (base_model.py:23-28), `val_and_test` (base_model.py:111-144),
the weighted AUC print (base_model.py:157-175), the patience counter that treats
`metric <= best` as no improvement (base_model.py:202-224), per-domain finetuning with
Keras EarlyStopping(val_AUC, min_delta=1e-4) + best-only checkpoint semantics
(base_model.py:41-109) and `save_result` (base_model.py:183-200).  Numerics go through
`self.model`, a `TowerEngine` (the compiled Keras model's stand-in).  Weights are saved
as .npz (h5py is not available): the flat trainable vector plus its segment table.
"""
import json
import os
import os.path as osp
import time

import numpy as np

from .. import meta, parallel
from .. import plan as mplan


class BaseModel(object):
    def __init__(self, dataset, config, engine_factory=None):
        self.n_uid = dataset.n_uid
        self.n_pid = dataset.n_pid
        self.n_domain = dataset.n_domain
        self.dataset = dataset
        self.config = config
        self.model_config = config["model"]
        self.train_config = config["train"]
        self.engine_factory = engine_factory
        stamp = time.strftime("%a-%b-%d-%H-%M-%S", time.localtime())
        self.checkpoint_path = osp.join(self.train_config["checkpoint_path"], self.model_config["name"],
                                        dataset.conf["name"], dataset.conf["domain_split_path"], stamp,
                                        "model_parameters.npz")
        self.result_path = osp.join(self.train_config["result_save_path"], self.model_config["name"],
                                    dataset.conf["name"], dataset.conf["domain_split_path"])
        self.learning_rate = self.train_config["learning_rate"]
        self.batch_size = dataset.batch_size
        sizes = {d: v["n_data"] for d, v in dataset.train_dataset.items()}
        self.shuffler = mplan.PassShuffler(sizes, dataset.shuffle_buffer_size, dataset.seed,
                                           shuffle=getattr(dataset, "shuffle_train", True))
        self.model = self.build_model()
        self._build_early_stop()

    def build_model(self):
        raise NotImplementedError("You must implement build model")

    def train(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ step / eval primitives
    def fit_domain(self, idx, max_steps=0, optimizer="adam", lr=None, trace=None, phase="fit"):
        """model.fit(iter, steps_per_epoch=n_step) / n_step x train_on_batch on one domain."""
        return meta.run_pass(self.model, idx, self.shuffler, self.batch_size,
                             self.learning_rate if lr is None else lr, trace if trace is not None else [],
                             phase, max_steps, optimizer)

    def evaluate_domain(self, idx, mode):
        """model.evaluate(d['data'], steps=d['n_step']) -> (loss, auc)."""
        return self.model.evaluate(idx, mode)

    # ------------------------------------------------------------------ finetune / separate training
    def separate_train_val_test(self, init_parms=True):
        """base_model.py:41-109.  init_parms=False is the finetune stage: plain SGD with
        `learning_rate` (base_model.py:69), restarted from the same weights for every domain.
        Under several processes (every rank holds the same weights) the domains are dealt round-robin."""
        weights = self.model.get_weights()
        if init_parms:
            self.model.optimizer_reset()
        rank, world = parallel.world()
        if world > 1:
            mine = [d for i, d in enumerate(self.dataset.train_dataset) if i % world == rank]
            _, _, dl, da = self._finetune_domains(lambda d: weights, "adam" if init_parms else "sgd", self.learning_rate,
                                                  domains=mine, summarise=False)
            dl, da = parallel.gather_domain_scalars({d: (dl[d], da[d]) for d in dl}, self.n_domain, self.model.device)
            return self._summarise("test", dl, da)
        return self._finetune_domains(lambda d: weights, "adam" if init_parms else "sgd", self.learning_rate)

    def _finetune_domains(self, start_weights, optimizer, lr, domains=None, summarise=True, per_domain_reset=False):
        domain_loss, domain_auc = {}, {}
        keep = self.model.get_weights()
        best = self.model.new_vector()
        self.finetune_log = {}        # {domain: epochs run, epoch of the kept checkpoint, val AUC per epoch}
        hook = getattr(self, "finetune_epoch_hook", None)       # tests: hook(domain, epoch, engine) after every epoch's pass
        for d in (self.dataset.train_dataset if domains is None else domains):
            self.model.set_weights(start_weights(d))
            if per_domain_reset:          # a freshly compiled optimiser per domain (deep_mtl_ctr.py:139-148)
                self.model.optimizer_reset()
            print("Train on domain: {}".format(d))
            # Keras EarlyStopping(monitor=val_AUC, mode=max, min_delta=1e-4) + ModelCheckpoint(best only)
            es_best, wait, ck_best = -np.inf, 0, -np.inf
            log = self.finetune_log[d] = {"epochs": 0, "best_epoch": -1, "val_auc": []}
            for epoch in range(self.train_config["epoch"]):
                self.fit_domain(d, optimizer=optimizer, lr=lr, phase="finetune")
                if hook is not None:
                    hook(d, epoch, self.model)
                _, val_auc = self.evaluate_domain(d, "val")
                log["epochs"] = epoch + 1
                log["val_auc"].append(float(val_auc))
                if val_auc > ck_best:
                    ck_best = val_auc
                    log["best_epoch"] = epoch
                    self.model.get_weights(out=best)
                if val_auc - 1e-4 > es_best:
                    es_best, wait = val_auc, 0
                else:
                    wait += 1
                    if wait >= self.train_config["patience"]:
                        break
            self.model.set_weights(best)
            p_loss, p_auc = self.evaluate_domain(d, "test")
            domain_loss[d], domain_auc[d] = float(p_loss), float(p_auc)
        self.model.set_weights(keep)
        if not summarise:
            return None, None, domain_loss, domain_auc
        return self._summarise("test", domain_loss, domain_auc)

    def val_and_test(self, mode):
        if mode not in ("val", "test"):
            raise ValueError("Mode can be either val or test, not: {}".format(mode))
        if mode == "test":
            self.load_model(self.checkpoint_path)      # best weights so far (base_model.py:121)
        domain_loss, domain_auc = {}, {}
        rank, world = parallel.world()
        for i, idx in enumerate(self.dataset.val_dataset if mode == "val" else self.dataset.test_dataset):
            if world > 1 and i % world != rank:        # every rank holds the same weights: deal the domains out
                continue
            p_loss, p_auc = self.evaluate_domain(idx, mode)
            domain_loss[idx], domain_auc[idx] = float(p_loss), float(p_auc)
        if world > 1:
            local = {d: (domain_loss[d], domain_auc[d]) for d in domain_loss}
            domain_loss, domain_auc = parallel.gather_domain_scalars(local, self.n_domain, self.model.device)
        return self._summarise(mode, domain_loss, domain_auc)

    def _summarise(self, mode, domain_loss, domain_auc):
        avg_loss = sum(domain_loss.values()) / len(domain_loss)
        avg_auc = sum(domain_auc.values()) / len(domain_auc)       # plain mean over domains
        print("Loss: ", domain_loss)
        self._format_print_domain_metric("AUC", domain_auc)
        print("Overall {} Loss: {}, AUC: {}, Weighted AUC: {}".format(mode, avg_loss, avg_auc,
                                                                     self._weighted_auc(mode, domain_auc)))
        return avg_loss, avg_auc, domain_loss, domain_auc

    def _format_print_domain_metric(self, name, domain_metric):
        print("{}: ".format(name))
        for key, value in domain_metric.items():
            print("{}: {}".format(key, value))

    def _weighted_auc(self, mode, domain_auc):
        info = self.dataset.dataset_info
        tag = "n_val" if "val" in mode else ("n_test" if "test" in mode else "n_train")
        total = sum(info[k][tag] for k in domain_auc)
        return sum(info[k][tag] * v for k, v in domain_auc.items()) / total

    # ------------------------------------------------------------------ persistence
    def save_model(self, path):
        aux = getattr(self.model, "aux", None)      # Star: PartitionedNorm moving statistics
        if path == self.checkpoint_path:            # the best-so-far checkpoint is also kept on the device
            self._best_in_memory = (self.model.get_weights().clone(), aux.clone() if aux is not None else None)
        if parallel.world()[0] != 0:                # one set of files per run
            return
        os.makedirs(osp.dirname(path) or ".", exist_ok=True)
        seg = self.model.segments
        np.savez(path, weights=self.model.get_weights().cpu().numpy(),
                 aux=aux.cpu().numpy() if aux is not None else np.zeros(0, np.float32),
                 segment_names=np.array(list(seg.keys())),
                 segment_offsets=np.array([v[0] for v in seg.values()], np.int64),
                 segment_counts=np.array([v[1] for v in seg.values()], np.int64))

    def load_model(self, path):
        import torch
        best = getattr(self, "_best_in_memory", None)
        if path == self.checkpoint_path and best is not None:
            self.model.set_weights(best[0])
            if best[1] is not None:
                self.model.aux.copy_(best[1])
            return
        with np.load(path) as z:
            w = z["weights"]
            aux = z["aux"] if "aux" in z.files else np.zeros(0, np.float32)
        if aux.size and getattr(self.model, "aux", None) is not None:
            self.model.aux.copy_(torch.from_numpy(aux).to(self.model.device))
        if w.shape[0] != self.model.n_params:
            raise ValueError("checkpoint has %d parameters, model has %d" % (w.shape[0], self.model.n_params))
        self.model.set_weights(torch.from_numpy(w).to(self.model.device))

    def save_result(self, avg_loss, avg_auc, domain_loss, domain_auc):
        folder = "loss_{:.3f}_auc_{:.3f}_{}".format(avg_loss, avg_auc, time.strftime("%a-%b-%d-%H-%M-%S",
                                                                                     time.localtime()))
        result_path = osp.join(self.result_path, folder)
        os.makedirs(result_path, exist_ok=True)
        with open(osp.join(result_path, "dataset_info.json"), "w") as f:
            json.dump(self.dataset.dataset_info, f)
        with open(osp.join(result_path, "config.json.example"), "w") as f:
            json.dump(self.config, f)
        with open(osp.join(result_path, "result.json"), "w") as f:
            json.dump({"avg_loss": avg_loss, "avg_auc": avg_auc, "domain_loss": domain_loss,
                       "domain_auc": domain_auc}, f)
        self.save_model(osp.join(result_path, "model_parameters.npz"))
        return result_path

    # ------------------------------------------------------------------ early stopping
    def _build_early_stop(self):
        self.patience = self.train_config["patience"]
        self.counter = 0
        self.best_metric = None
        self.early_stop = False

    def early_stop_step(self, metric):
        if self.best_metric is None:
            self.best_metric = metric
            self.save_model(self.checkpoint_path)
        elif metric <= self.best_metric:                    # ties count as "no improvement"
            self.counter += 1
            print("EarlyStopping counter: {} out of {}, Best AUC: {}".format(self.counter, self.patience,
                                                                           self.best_metric))
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.save_model(self.checkpoint_path)
            self.best_metric = metric
            self.counter = 0
        return self.early_stop
