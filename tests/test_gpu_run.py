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
                                  "mlp_meta_maml"])
def test_run_config_on_gpu(tmp_path, name):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from mamdr_amd import cli
    with open(os.path.join(ROOT, "config", "Taobao-10", "deepctr_DN+DR.json")) as f:
        cfg = copy.deepcopy(json.load(f))
    cfg["model"]["name"] = name
    cfg["train"].update(epoch=3, patience=1, sample_num=2, meta_learning_rate=0.5,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    if "maml" in name:      # MAML's meta_learning_rate is the step of an outer ADAM, not an interpolation weight
        cfg["train"]["meta_learning_rate"] = 0.003
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
    assert z["weights"].shape[0] == 139777 + 128 * 10 + 3 and np.isfinite(z["weights"]).all()
