"""AUC of a sharded MAMDR run (run.py's entry under torch.distributed.run) for a given number of ranks.
Used to measure what the per-rank DN sub-sequences + one all-reduce cost in AUC against the
single-process loop; on a 1-GPU box: MAMDR_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node N
tools/dist_auc.py [config] [epochs]"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamdr_amd import cli  # noqa: E402

cfg = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "config/Taobao-10/deepctr_DN+DR.json"))
cfg["train"].update(epoch=int(sys.argv[2]) if len(sys.argv) > 2 else 6, patience=100,
                    result_save_path="/tmp/dist_auc/result", checkpoint_path="/tmp/dist_auc/ckpt")
cfg["model"]["name"] = "mlp_meta_mamdr"
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    avg_loss, avg_auc, dl, da = cli.main(cfg)
if int(os.environ.get("RANK", "0")) == 0:
    print("world %s: avg test AUC %.5f avg loss %.5f per-domain %s" % (os.environ.get("WORLD_SIZE", "1"), avg_auc, avg_loss,
                                                                      " ".join("%.4f" % da[k] for k in sorted(da))))
