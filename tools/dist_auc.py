"""AUC of a sharded MAMDR run (run.py's entry under torch.distributed.run) for a given number of ranks.
Used to measure what the per-rank DN sub-sequences + one all-reduce cost in AUC against the
single-process loop, for both DN modes, next to the run-to-run spread of the single-process loop itself;
on a 1-GPU box: MAMDR_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node N
tools/dist_auc.py [config] [epochs] [dn_mode] [run seed] [model name]   (the generated logs keep seed 123 whatever the
run seed; model name default mlp_meta_mamdr, e.g. star_meta_mamdr with config/Taobao-10/star_taobao.json)"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamdr_amd import cli  # noqa: E402

cfg = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "config/Taobao-10/deepctr_DN+DR.json"))
dn_mode = sys.argv[3] if len(sys.argv) > 3 else "sharded"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else cfg["dataset"]["seed"]
cfg["train"].update(epoch=int(sys.argv[2]) if len(sys.argv) > 2 else 6, patience=100, dn_mode=dn_mode,
                    result_save_path="/tmp/dist_auc/result", checkpoint_path="/tmp/dist_auc/ckpt")
cfg["dataset"]["synthetic_seed"] = cfg["dataset"]["seed"]
if os.environ.get("MAMDR_DIST_AUC_SCALE"):        # fewer rows per domain (quick runs on the big shapes)
    cfg["dataset"]["synthetic_scale"] = float(os.environ["MAMDR_DIST_AUC_SCALE"])
cfg["dataset"]["seed"] = seed
cfg["model"]["name"] = sys.argv[5] if len(sys.argv) > 5 else "mlp_meta_mamdr"
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    avg_loss, avg_auc, dl, da = cli.main(cfg)
if int(os.environ.get("RANK", "0")) == 0:
    print("DISTAUC " + json.dumps({"world": int(os.environ.get("WORLD_SIZE", "1")), "dn_mode": dn_mode, "seed": seed, "model": cfg["model"]["name"],
                                   "avg_auc": avg_auc, "avg_loss": avg_loss, "domain_auc": [da[k] for k in sorted(da)]}))
