"""run.py --config end to end on the GPU (real HIP engine), small synthetic Taobao-10 slice."""
import copy
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["mlp_meta_mamdr_finetune", "mlp_meta_domain_negotiation", "mlp_meta_reptile", "mlp",
                                  "mlp_meta_maml", "deepfm_meta_domain_negotiation_finetune", "mlp_meta_mldg",
                                  "mlp_uncertainty_weight", "mlp_pcgrad"])
def test_run_config_on_gpu(tmp_path, name):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import cli
    with open(os.path.join(ROOT, "config", "Taobao-10", "deepctr_DN+DR.json")) as f:
        cfg = copy.deepcopy(json.load(f))
    cfg["model"]["name"] = name
    cfg["train"].update(epoch=3, patience=1, sample_num=2, meta_learning_rate=0.5,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    if "maml" in name or "mldg" in name or "pcgrad" in name:   # meta_learning_rate is the step of an outer ADAM, not an interpolation weight
        cfg["train"]["meta_learning_rate"] = 0.003
    if "mldg" in name:      # the reference's MLDG config splits every domain 80 / 20 into meta-train / meta-val
        cfg["train"].update(meta_split="meta-train/val", meta_split_ratio=0.8)
    cfg["dataset"].update(batch_size=256, synthetic="taobao10", synthetic_scale=0.1)
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg)
    assert len(domain_auc) == 10 and np.isfinite(avg_loss)
    assert avg_auc > 0.6, (name, avg_auc)          # the tower learns the planted signal
    rdir = os.path.join(cfg["train"]["result_save_path"], name, "Taobao", "split_by_theme_10")
    run = os.listdir(rdir)[0]
    with open(os.path.join(rdir, run, "result.json")) as f:
        res = json.load(f)
    assert abs(res["avg_auc"] - avg_auc) < 1e-12
    z = np.load(os.path.join(rdir, run, "model_parameters.npz"))
    # dense block: 139777 tower weights + domain table (+ DeepFM's linear domain table), padded to 4 floats
    n_dense = 139777 + 128 * 10 + (10 if "deepfm" in name else 0) + (10 if "uncertainty_weight" in name else 0)
    assert z["weights"].shape[0] == (n_dense + 3) // 4 * 4 and np.isfinite(z["weights"]).all()


def test_run_amazon6_deepfm_config_trainable_tables(tmp_path):
    """BASELINE config 3 (deepfm_meta_domain_negotiation, trainable 128-d tables, no pretraining) through
    run.py's entry, on a small synthetic slice instead of the 79 M-parameter Amazon-6 tables."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import cli
    with open(os.path.join(ROOT, "config", "Amazon_6", "deepfm_DN.json")) as f:
        cfg = json.load(f)
    assert cfg["model"]["name"] == "deepfm_meta_domain_negotiation" and cfg["train"]["emb_trainable"]
    cfg["train"].update(epoch=2, patience=1, meta_learning_rate=0.5,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    cfg["dataset"].update(batch_size=256, synthetic="taobao10", synthetic_scale=0.1)
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg)
    assert len(domain_auc) == 10 and np.isfinite(avg_loss) and 0.0 <= avg_auc <= 1.0
    rdir = os.path.join(cfg["train"]["result_save_path"], cfg["model"]["name"], "Amazon", "split_by_category_6")
    z = np.load(os.path.join(rdir, os.listdir(rdir)[0], "model_parameters.npz"))
    assert np.isfinite(z["weights"]).all() and z["weights"].shape[0] > 128 * 1000


@pytest.mark.parametrize("name,trainable", [("star", False), ("star_meta_mamdr_finetune", False),
                                            ("star_meta_domain_negotiation", True)])
def test_run_star_configs_on_gpu(tmp_path, name, trainable):
    """Star tower (PartitionedNorm + StarFCN) through run.py's entry: plain alternate training, MAMDR over
    the name-filtered meta parameters (theta / phi = tables, shared kernels, shared biases) + finetune,
    and DN with trainable tables."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import cli
    with open(os.path.join(ROOT, "config", "Taobao-10", "star_taobao.json")) as f:
        cfg = json.load(f)
    cfg["model"]["name"] = name
    cfg["train"].update(epoch=3, patience=1, sample_num=2, meta_learning_rate=0.5, emb_trainable=trainable,
                        load_pretrain_emb=not trainable,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    cfg["dataset"].update(batch_size=256, synthetic="taobao10", synthetic_scale=0.1)
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg)
    assert len(domain_auc) == 10 and np.isfinite(avg_loss)
    if not trainable:
        assert avg_auc > 0.6, (name, avg_auc)      # pretrained tables: the tower learns the planted signal
    rdir = os.path.join(cfg["train"]["result_save_path"], name, "Taobao", "split_by_theme_10")
    z = np.load(os.path.join(rdir, os.listdir(rdir)[0], "model_parameters.npz"))
    assert np.isfinite(z["weights"]).all() and z["aux"].shape[0] == 4 * 10 * 384 + 12
    assert (z["aux"][4 * 10 * 384:4 * 10 * 384 + 10] > 0).all()      # every domain's moving statistics were updated


def test_amazon13_star_config_parses_and_selects_meta_prefix():
    """BASELINE config 5 file: star_meta_mamdr, bs 8192, trainable tables, the reference's meta filter."""
    with open(os.path.join(ROOT, "config", "Amazon_13", "star_DN+DR.json")) as f:
        cfg = json.load(f)
    assert cfg["model"]["name"] == "star_meta_mamdr" and cfg["dataset"]["batch_size"] == 8192
    assert cfg["train"]["meta_parms"] == ["emb", "kernel_shared", "bias_shared"] and cfg["train"]["emb_trainable"]
