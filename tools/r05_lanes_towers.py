"""round 5: lanes for the other towers -- run.py's pipeline (config/Taobao-10/deepctr_DN+DR.json as configured, full rows, bs 1,024,
name <tower>_meta_mamdr) on 1 and 4 lanes; domain-steps/s of the meta epochs from the difference of a 40-epoch and a 10-epoch run
(val_every_step 10^6: only epoch 0 validates; the fixed costs -- data, tables, the one validation, the final test -- cancel).  python tools/r05_lanes_towers.py [tower,...]"""
import contextlib
import copy
import io
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamdr_amd import cli        # noqa: E402
import torch                     # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = json.load(open(os.path.join(root, "config", "Taobao-10", "deepctr_DN+DR.json")))
towers = sys.argv[1].split(",") if len(sys.argv) > 1 else ["mlp", "deepfm", "nfm", "ccpm", "autoint"]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
for tower in towers:
    out = {}
    for lanes in (1, 1, 4):          # (the first pass is a warm-up: dataset cache, code objects, allocator)
        t = {}
        for epochs in (10, 40):
            cfg = copy.deepcopy(base)
            tmp = tempfile.mkdtemp()
            cfg["model"]["name"] = tower + "_meta_mamdr"
            cfg["train"].update(epoch=epochs, patience=99, val_every_step=10 ** 6, lanes=lanes, result_save_path=tmp + "/r", checkpoint_path=tmp + "/c")
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                res = cli.main(cfg)
            torch.cuda.synchronize()
            t[epochs] = time.time() - t0
        out[lanes] = ((t[40] - t[10]) / 30.0, res[1])
    steps = 1255.3          # domain-steps of an average MAMDR epoch on Taobao-10 bs 1,024 (bench.py: domain_steps_per_epoch)
    print("%-8s 1 lane %.1f ms / epoch = %.0f domain-steps/s | 4 lanes %.1f ms / epoch = %.0f domain-steps/s | x%.2f   (avg test AUC %.4f / %.4f)" % (
        tower, out[1][0] * 1e3, steps / out[1][0], out[4][0] * 1e3, steps / out[4][0], out[1][0] / out[4][0], out[1][1], out[4][1]), flush=True)
