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


def test_bench_gpus2_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` without a rendezvous starts two ranks itself (child torch.distributed.run, before
    touching the GPU) and relays ONE JSON line with n_gpus 2.  On this 1-GPU box both ranks share device 0 over
    gloo (MAMDR_BENCH_SHARE_GPU=1); with one GPU per rank the same code path runs RCCL."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    env = dict(os.environ, MAMDR_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--cpu-budget", "0", "--no-profile", "--no-targets"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["backend"] == "gloo" and rec["value"] > 0
    assert rec["config"]["domain_steps_per_epoch"] > 1000 and 1.0 < rec["partition_speedup_bound"] <= 2.0


def test_epoch_shuffles_equal_per_pass_shuffles():
    """plan.EpochShuffles (every permutation of an epoch from one C call, one upload) hands out exactly the
    permutations PassShuffler would have produced pass by pass."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import plan as mplan
    sizes = [700, 12000, 31, 2500]
    plan = {"seq": [2, 0, 3, 1], "dr": [(2, [0, 2]), (0, [1, 0]), (3, [2, 3]), (1, [3, 1])]}
    passes = mplan.epoch_passes(plan)
    assert [d for d, _ in passes][:4] == [2, 0, 3, 1] and len(passes) == 4 + 2 * 8
    a = mplan.PassShuffler(sizes, 10000, 91)
    es = mplan.EpochShuffles(mplan.PassShuffler(sizes, 10000, 91), torch.device("cuda", 0))
    for ep in range(3):                          # both staging buffers and a reuse
        es.prepare(passes)
        for d, _ in passes:
            got = es(d).cpu().numpy()
            assert np.array_equal(got, a(d)) and sorted(got.tolist()) == list(range(sizes[d]))
    with pytest.raises(RuntimeError):
        es.prepare(passes)
        es(0)                                    # the epoch's first pass is over domain 2
