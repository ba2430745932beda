"""MultiDomainDataset -- host-side mirror of the reference's utils/dataset.py.

Same constructor argument (the `dataset` section of a config) and the same
attributes the towers and wrappers read (utils/dataset.py:41-130): n_uid, n_pid,
n_domain, train_dataset / val_dataset / test_dataset (OrderedDict: domain ->
{"data", "n_step", "n_data"}), ctr_ratio, dataset_info, user_emb / item_emb, conf,
batch_size, shuffle_buffer_size.

Differences, by design:
* "data" is a dict of int32/fp32 numpy columns (uid, pid, domain, label) instead
  of a tf.data pipeline; batching, the kept final partial batch
  (utils/dataset.py:25) and the per-pass shuffle buffer (utils/dataset.py:27-37)
  are applied on the device side (mamdr_train_steps + PassShuffler).
* the reference's CSV/JSON layout (dataset/*/split.py output) is parsed once and
  cached as one .npz next to it; `dataset.synthetic` (a shape name of
  mamdr_amd/synthetic.py) generates Taobao-/Amazon-shaped logs instead, because
  the real datasets are not redistributable and there is no network.
* pretrained tables are fp32 arrays; the JSON dict of space-separated strings
  (model_zoo/DeepCTR/deepctr.py:104-113) is converted at load time.
"""
import collections
import glob
import json
import math
import os
import os.path as osp
import zipfile

import numpy as np

from .. import synthetic

COLUMNS = ("uid", "pid", "domain", "label")


def read_csv_columns(path):
    """csv with header uid,pid,domain,label (dataset/Taobao/split.py:21) -> columns."""
    with open(path, "r") as f:
        header = f.readline().strip().split(",")
    if not set(COLUMNS) <= set(header):
        raise ValueError("%s: header %s lacks %s" % (path, header, COLUMNS))
    raw = np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.int64, ndmin=2)
    cols = {}
    for name in COLUMNS:
        c = raw[:, header.index(name)] if raw.size else np.zeros(0, np.int64)
        cols[name] = c.astype(np.float32 if name == "label" else np.int32)
    return cols


def emb_dict_to_array(emb_dict, n, dim):
    """deepctr.py:104-113: rows missing from the dict stay zero."""
    out = np.zeros((n, dim), np.float32)
    for key in sorted(emb_dict.keys()):
        out[int(key)] = np.asarray(emb_dict[key].split(" "), dtype="float32")
    return out


def write_reference_layout(gen, root, domain_split_path):
    """dump generated logs in the reference's on-disk format (split.py output)."""
    base = osp.join(root, domain_split_path)
    os.makedirs(osp.join(base, "processed_data"), exist_ok=True)
    with open(osp.join(base, "processed_data", "uid2id.json"), "w") as f:
        json.dump({"id": gen["n_user"], "raw_id2id": {}}, f)
    with open(osp.join(base, "processed_data", "pid2id.json"), "w") as f:
        json.dump({"id": gen["n_item"], "raw_id2id": {}}, f)
    for name, key in (("user_emb.json", "user_emb"), ("item_emb.json", "item_emb")):
        tab = gen["tables"][key]
        with open(osp.join(base, "processed_data", name), "w") as f:
            json.dump({str(i): " ".join(repr(float(v)) for v in tab[i]) for i in range(tab.shape[0])}, f)
    for d in range(gen["n_domain"]):
        dpath = osp.join(base, "domain_%d" % d)
        os.makedirs(dpath, exist_ok=True)
        for split in ("train", "val", "test"):
            c = gen["data"][split][d]
            arr = np.stack([c["uid"], c["pid"], c["domain"], c["label"].astype(np.int64)], axis=1)
            np.savetxt(osp.join(dpath, split + ".csv"), arr, fmt="%d", delimiter=",", header="uid,pid,domain,label",
                       comments="")
        with open(osp.join(dpath, "domain_property.json"), "w") as f:
            json.dump({"ctr_ratio": gen["info"][d]["ctr_ratio"]}, f)


class MultiDomainDataset(object):
    def __init__(self, conf):
        self.conf = conf
        self.dataset_path = conf["dataset_path"]
        self.domain_split_path = osp.join(self.dataset_path, conf["domain_split_path"])
        self.seed = conf["seed"]
        self.batch_size = conf["batch_size"]
        self.shuffle_buffer_size = conf["shuffle_buffer_size"]
        self.shuffle_train = not ("fixed_train" in conf and conf["fixed_train"])     # utils/dataset.py:76
        self.train_dataset = collections.OrderedDict()
        self.val_dataset = collections.OrderedDict()
        self.test_dataset = collections.OrderedDict()
        self.ctr_ratio = collections.OrderedDict()
        self.user_emb = None
        self.item_emb = None
        if conf.get("synthetic"):
            # (synthetic_seed: the generated logs' own seed, so that runs with different `seed` -- planner, shuffles,
            # initial tensors -- can share one data set; default: the run's seed)
            self._from_generated(synthetic.generate(conf["synthetic"], batch_size=self.batch_size,
                                                    seed=int(conf.get("synthetic_seed", self.seed)),
                                                    scale=float(conf.get("synthetic_scale", 1.0))))
        else:
            self._from_files()
        print("Found {} domain, in: {}".format(self.n_domain, self.domain_split_path))

    # ------------------------------------------------------------------ sources
    def _add(self, split_dict, idx, cols):
        n = int(cols["uid"].shape[0])
        split_dict[idx] = {"data": cols, "n_step": int(math.ceil(n / float(self.batch_size))), "n_data": n}

    def _from_generated(self, gen):
        self.n_uid, self.n_pid, self.n_domain = gen["n_user"], gen["n_item"], gen["n_domain"]
        if gen["spec"].get("pretrained", True):
            self.user_emb, self.item_emb = gen["tables"]["user_emb"], gen["tables"]["item_emb"]
        for d in range(self.n_domain):
            self.ctr_ratio[d] = gen["info"][d]["ctr_ratio"]
            self._add(self.train_dataset, d, gen["data"]["train"][d])
            self._add(self.val_dataset, d, gen["data"]["val"][d])
            self._add(self.test_dataset, d, gen["data"]["test"][d])

    def _from_files(self):
        base = self.domain_split_path
        if not osp.isdir(base):
            raise FileNotFoundError("%s not found; set dataset.synthetic (e.g. \"taobao10\") to generate "
                                    "Taobao-/Amazon-shaped logs instead" % base)
        cache = osp.join(base, "mamdr_amd_cache.npz")          # (the columns do not depend on the batch size)
        with open(osp.join(base, "processed_data/uid2id.json"), "r") as f:
            self.n_uid = json.load(f)["id"]                                   # utils/dataset.py:50-52
        with open(osp.join(base, "processed_data/pid2id.json"), "r") as f:
            self.n_pid = json.load(f)["id"]
        domains = glob.glob(osp.join(base, "domain_*"))
        domains.sort(key=lambda x: int(x.split("_")[-1]))                    # utils/dataset.py:63-64
        self.n_domain = len(domains)
        # the cache is valid only for the files it was built from: (name, size, mtime) of every source file
        sources = [osp.join(base, "processed_data", n) for n in ("uid2id.json", "pid2id.json", "item_emb.json",
                                                                  "user_emb.json")]
        for d_path in domains:
            sources += [osp.join(d_path, n) for n in ("train.csv", "val.csv", "test.csv", "domain_property.json")]
        stamp = json.dumps([[osp.relpath(f, base), os.stat(f).st_size, int(os.stat(f).st_mtime_ns)]
                            for f in sources if osp.exists(f)])
        cached = None
        if osp.exists(cache):
            try:
                z = np.load(cache)
                if "source_stamp" in z.files and str(z["source_stamp"]) == stamp:
                    cached = z
            except (OSError, ValueError, zipfile.BadZipFile):
                cached = None
        store = {"source_stamp": np.array(stamp)}
        if self.conf["name"] == "Taobao":                                     # utils/dataset.py:57-61
            if cached is not None:
                self.user_emb, self.item_emb = cached["user_emb"], cached["item_emb"]
            else:
                with open(osp.join(base, "processed_data/item_emb.json"), "r") as f:
                    item = json.load(f)
                with open(osp.join(base, "processed_data/user_emb.json"), "r") as f:
                    user = json.load(f)
                dim = len(next(iter(user.values())).split(" "))
                self.user_emb = emb_dict_to_array(user, self.n_uid, dim)
                self.item_emb = emb_dict_to_array(item, self.n_pid, dim)
            store["user_emb"], store["item_emb"] = self.user_emb, self.item_emb
        for d_path in domains:
            idx = int(osp.split(d_path)[-1].split("_")[-1])
            for split, target in (("train", self.train_dataset), ("val", self.val_dataset),
                                  ("test", self.test_dataset)):
                if cached is not None:
                    cols = {c: cached["d%d_%s_%s" % (idx, split, c)] for c in COLUMNS}
                else:
                    cols = read_csv_columns(osp.join(d_path, split + ".csv"))
                for c in COLUMNS:
                    store["d%d_%s_%s" % (idx, split, c)] = cols[c]
                self._add(target, idx, cols)
            with open(osp.join(d_path, "domain_property.json")) as f:
                self.ctr_ratio[idx] = json.load(f)["ctr_ratio"]
        # one writer (rank 0), written beside the target and renamed into place: readers never see a torn file
        if cached is None and int(os.environ.get("RANK", "0")) == 0:
            tmp = "%s.tmp.%d.npz" % (cache, os.getpid())
            try:
                np.savez(tmp, **store)
                os.replace(tmp, cache)
            except OSError:
                try:
                    os.remove(tmp)
                except OSError:
                    pass

    # ------------------------------------------------------------------ reference API
    def get_train_dataset(self, domain_idx):
        return self.train_dataset[domain_idx]

    def get_val_dataset(self, domain_idx):
        return self.val_dataset[domain_idx]

    def get_test_dataset(self, domain_idx):
        return self.test_dataset[domain_idx]

    @property
    def dataset_info(self):
        """utils/dataset.py:110-130 (same keys; domain keys are ints and become strings in JSON)."""
        info = {"n_user": self.n_uid, "n_item": self.n_pid}
        tt, tv, te = 0, 0, 0
        for i in self.train_dataset:
            info[i] = {"n_train": self.train_dataset[i]["n_data"], "n_val": self.val_dataset[i]["n_data"],
                       "n_test": self.test_dataset[i]["n_data"], "ctr_ratio": self.ctr_ratio[i]}
            tt += self.train_dataset[i]["n_data"]
            tv += self.val_dataset[i]["n_data"]
            te += self.test_dataset[i]["n_data"]
        info["total_train"], info["total_val"], info["total_test"] = tt, tv, te
        return info
