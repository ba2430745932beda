"""Derive the dataset variants of the run configurations from this repo's Taobao-10 templates.

The reference ships one directory of JSON files per dataset split (config/{Taobao-10,Taobao_20,Taobao_30,Amazon_6,
Amazon_13}): the files of a split differ from the Taobao-10 ones only in the dataset section (name, paths), in
`train.sample_num`, and -- for Amazon, which has no pretrained embeddings -- in `load_pretrain_emb: false,
emb_trainable: true`.  This script writes those variants for the towers built here (mlp, star; the plain,
Domain Negotiation and MAMDR entries), keeping every other value of the templates.  The multi-task baselines
(shared_bottom / mmoe / ple) get their per-split hyperparameters from the table MTL below (the values of the
reference's config/*/{shared_bottom,mmoe,ple}.json: layer widths, expert counts, learning rate); the Amazon ones
train their tables.  The files mirror the reference's VALUE FOR VALUE, quirks included (tests/test_host_logic.py::
test_configs_mirror_the_reference pins that where /root/reference is present): batch 1,024 everywhere (BASELINE.json's
Taobao-30 bs 4,096 case is the extra file Taobao_30/deepctr_DN+DR_bs4096.json), Taobao_30/ple.json reading the 20-domain
split, the Amazon shared_bottom files asking for pretrained tables the Amazon datasets do not have (run.py then says so,
as the reference's deepctr.py:104-116 would fail on `None` tables).

usage: python tools/make_configs.py    (idempotent; writes under config/)"""
import copy
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config")


def load(name):
    with open(os.path.join(CFG, "Taobao-10", name)) as f:
        return json.load(f)


SPLITS = {
    # directory: dataset overrides, train overrides, MAMDR sample_num, star file name
    "Taobao_20": (dict(domain_split_path="split_by_theme_20", synthetic="taobao20"), {}, 19, "star_taobao.json"),
    "Taobao_30": (dict(domain_split_path="split_by_theme_30", synthetic="taobao30"), {}, 5, "star_taobao.json"),
    "Amazon_6": (dict(name="Amazon", dataset_path="dataset/Amazon", domain_split_path="split_by_category_6",
                      synthetic="amazon6"), dict(load_pretrain_emb=False, emb_trainable=True), 3, "star.json"),
    "Amazon_13": (dict(name="Amazon", dataset_path="dataset/Amazon", domain_split_path="split_by_category_13",
                       synthetic="amazon13"), dict(load_pretrain_emb=False, emb_trainable=True), 5, "star.json"),
}
# output file -> (template in config/Taobao-10, per-file train overrides)
FILES = {
    "deepctr.json": ("deepctr_taobao_10.json", {}),
    "deepctr_DN.json": ("deepctr_DN_taobao_10.json", {}),
    "deepctr_DN+DR.json": ("deepctr_DN+DR.json", "sample_num"),
    "STAR": ("star_taobao.json", "star"),
}


# multi-task baselines: split -> file -> (model overrides, learning rate)
MTL = {
    "Taobao-10": {"shared_bottom": (dict(hidden_dim=[512, 256, 128], tower_hidden_dim=[64]), 1e-4),
                  "mmoe": (dict(hidden_dim=[512, 256, 128], tower_hidden_dim=[64], num_experts=2, gate_dnn_hidden_units=[64]), 1e-4),
                  "ple": (dict(hidden_dim=[256], tower_hidden_dim=[64], specific_expert_num=10, shared_expert_num=2,
                               gate_dnn_hidden_units=[64], num_levels=1), 1e-4)},
    "Taobao_20": {"shared_bottom": (dict(hidden_dim=[512, 256], tower_hidden_dim=[128]), 1e-4),
                  "mmoe": (dict(hidden_dim=[512, 256], tower_hidden_dim=[128], num_experts=2, gate_dnn_hidden_units=[64]), 1e-4),
                  "ple": (dict(hidden_dim=[256], tower_hidden_dim=[64], specific_expert_num=15, shared_expert_num=2,
                               gate_dnn_hidden_units=[64], num_levels=1), 1e-4)},
    "Taobao_30": {"shared_bottom": (dict(hidden_dim=[512, 256], tower_hidden_dim=[128]), 1e-4),
                  "mmoe": (dict(hidden_dim=[512, 256], tower_hidden_dim=[128], num_experts=2, gate_dnn_hidden_units=[64]), 1e-4),
                  "ple": (dict(hidden_dim=[512, 256], tower_hidden_dim=[64], specific_expert_num=3, shared_expert_num=2,
                               gate_dnn_hidden_units=[64], num_levels=1), 1e-4)},
    "Amazon_6": {"shared_bottom": (dict(hidden_dim=[256, 128], tower_hidden_dim=[64]), 1e-3),
                 "mmoe": (dict(hidden_dim=[256, 128], tower_hidden_dim=[64], num_experts=5, gate_dnn_hidden_units=[64]), 1e-4),
                 "ple": (dict(hidden_dim=[512, 256], tower_hidden_dim=[64], specific_expert_num=5, shared_expert_num=2,
                              gate_dnn_hidden_units=[64], num_levels=1), 1e-4)},
    "Amazon_13": {"shared_bottom": (dict(hidden_dim=[256, 128], tower_hidden_dim=[64]), 1e-3),
                  "mmoe": (dict(hidden_dim=[256, 128], tower_hidden_dim=[64], num_experts=5, gate_dnn_hidden_units=[64]), 1e-4),
                  "ple": (dict(hidden_dim=[512, 256], tower_hidden_dim=[64], specific_expert_num=5, shared_expert_num=2,
                               gate_dnn_hidden_units=[64], num_levels=1), 1e-4)},
}
# the train section of the plain (non-meta) entries: the reference's mmoe.json keys
PLAIN_TRAIN_KEYS = ("load_pretrain_emb", "emb_trainable", "epoch", "learning_rate", "result_save_path", "checkpoint_path",
                    "loss", "optimizer", "patience", "histogram_freq", "shuffle_buff_size")


def write_mtl(written):
    base = load("deepctr_taobao_10.json")
    for split, files in MTL.items():
        ds_over, tr_over = ({}, {}) if split == "Taobao-10" else SPLITS[split][:2]
        for name, (model_over, lr) in files.items():
            cfg = copy.deepcopy(base)
            cfg["model"]["name"] = name
            cfg["model"].update(model_over)
            cfg["train"] = {k: cfg["train"][k] for k in PLAIN_TRAIN_KEYS if k in cfg["train"]}
            cfg["train"].update(tr_over)
            cfg["train"]["learning_rate"] = lr
            cfg["dataset"].update(ds_over)
            cfg["dataset"]["batch_size"] = 1024
            if name == "shared_bottom" and split.startswith("Amazon"):      # as the reference's files
                cfg["train"].update(load_pretrain_emb=True, emb_trainable=False)
            if name == "ple" and split == "Taobao_30":                      # as the reference's file
                cfg["dataset"].update(domain_split_path="split_by_theme_20", synthetic="taobao20")
            path = os.path.join(CFG, split, name + ".json")
            if os.path.exists(path):
                continue
            with open(path, "w") as f:
                json.dump(cfg, f, indent=2)
                f.write("\n")
            written.append(os.path.relpath(path, ROOT))


def main():
    written = []
    write_mtl(written)
    for split, (ds_over, tr_over, sample_num, star_name) in SPLITS.items():
        os.makedirs(os.path.join(CFG, split), exist_ok=True)
        for out, (template, extra) in FILES.items():
            cfg = copy.deepcopy(load(template))
            cfg["dataset"].update(ds_over)
            cfg["train"].update(tr_over)
            if extra == "sample_num":
                cfg["train"]["sample_num"] = sample_num
            name = out
            if extra == "star":
                name = star_name
                cfg["model"]["name"] = "star"              # the reference's per-split star files train the plain tower
                for k in ("meta_parms",):
                    cfg["train"].pop(k, None)
            path = os.path.join(CFG, split, name)
            if os.path.exists(path):                       # hand-written entries (the BASELINE configs) stay
                continue
            with open(path, "w") as f:
                json.dump(cfg, f, indent=2)
                f.write("\n")
            written.append(os.path.relpath(path, ROOT))
    # BASELINE.json configs[3]: mlp_meta_mamdr on Taobao-30 at batch 4,096 (the reference's file says 1,024 and _finetune)
    path = os.path.join(CFG, "Taobao_30", "deepctr_DN+DR_bs4096.json")
    if not os.path.exists(path):
        with open(os.path.join(CFG, "Taobao_30", "deepctr_DN+DR.json")) as f:
            cfg = json.load(f)
        cfg["model"]["name"] = "mlp_meta_mamdr"
        cfg["dataset"]["batch_size"] = 4096
        with open(path, "w") as f:
            json.dump(cfg, f, indent=2)
            f.write("\n")
        written.append(os.path.relpath(path, ROOT))
    print("\n".join(written) if written else "nothing to write")


if __name__ == "__main__":
    main()
