"""Throughput of the generic-layer engine (csrc/graph_engine.hip) on the reference's Taobao-10 multi-task configurations
and on the deepctr single-output towers it hosts: domain-steps/s of the alternate training loop (deep_mtl_ctr.py:69-96 /
deepctr.py:63-93: one full pass per domain per epoch), synthetic Taobao-10 logs, batch 1024, inputs resident in HBM.
usage: python tools/graph_bench.py [epochs [tower,tower...|all [inproc]]]    -> one JSON line per tower
(`inproc`: every tower in THIS process -- what a profiler needs, which must not see a process that has initialised the GPU
start another program)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mamdr_amd import cli, synthetic  # noqa: E402
from mamdr_amd.plan import PassShuffler  # noqa: E402
from mamdr_amd.utils import MultiDomainDataset  # noqa: E402

EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None
INPROC = len(sys.argv) > 3 and sys.argv[3] == "inproc"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def step_macs_per_row(eng, domain=0):
    """forward multiply-adds per batch row of the dense layers on ONE step's path (a step of task d runs the shared
    experts and task d's own experts / gate / tower); a training step costs 6 flop per such MAC (forward, d input,
    d weights).  AutoInt's attention projections run on 3 token rows per batch row."""
    if hasattr(eng, "shapes"):
        ranges = eng.task_ranges(domain) if hasattr(eng, "task_ranges") else [(0, eng.n_params)]
        macs = 0
        for name, (rows, cols) in eng.shapes.items():
            off = eng.segments[name][0]
            if not any(o <= off < o + c for o, c in ranges) or rows <= 1 or name.endswith("_emb") or name.startswith("lin_"):
                continue
            macs += rows * cols * (3 if name.startswith("att") else 1)
        return macs
    return sum(c for n, (_, c) in eng.segments.items() if n in ("W0", "W0x", "W1", "W2", "wo"))


PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md (bench.py uses the same figure)


TOWERS = ("shared_bottom", "mmoe", "ple", "nfm", "pnn", "ccpm", "autoint")
if not INPROC and (ONLY is None or len(ONLY) > 1):
    # one process per tower (started before this one touches the GPU): in one process a tower measured after a bigger one
    # now and then runs 3x slower than on its own (nfm after ple: 113 us, 113 us, 367 us, ... in five identical runs)
    import subprocess
    for name in TOWERS:
        if ONLY is None or name in ONLY:
            subprocess.run([sys.executable, os.path.abspath(__file__), str(EPOCHS), name], check=False)
    sys.exit(0)

for cfg_name in TOWERS:
    if ONLY is not None and cfg_name not in ONLY:
        continue
    path = os.path.join(ROOT, "config", "Taobao-10", cfg_name + ".json")
    if os.path.exists(path):
        cfg = json.load(open(path))
    else:
        cfg = json.load(open(os.path.join(ROOT, "config", "Taobao-10", "deepctr_taobao_10.json")))
        cfg["model"]["name"] = cfg_name
    cfg["train"].update(result_save_path="/tmp/graph_bench/result", checkpoint_path="/tmp/graph_bench/ckpt")
    ds = MultiDomainDataset(cfg["dataset"])
    if os.environ.get("MAMDR_GRAPH_DIAG_REPLAY"):      # stream capture needs a stream of its own (not the legacy default one)
        torch.cuda.set_stream(torch.cuda.Stream())
    model = cli.build_model(cfg, ds)
    eng = model.model
    D = ds.n_domain
    sizes = {d: v["n_data"] for d, v in ds.train_dataset.items()}
    sh = PassShuffler(sizes, 10000, 123)
    perms = {d: torch.from_numpy(sh(d)).to(eng.device) for d in range(D)}
    steps_per_epoch = sum(-(-sizes[d] // ds.batch_size) for d in range(D))

    def epoch():
        for d in range(D):
            eng.train_steps(d, perm=perms[d], lr=cfg["train"]["learning_rate"])
    epoch()
    torch.cuda.synchronize()
    graph = hasattr(eng, "shapes")
    l0 = int(eng.lib.mamdr_graph_launch_count())
    t0 = time.perf_counter()
    for _ in range(EPOCHS):
        epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches = (int(eng.lib.mamdr_graph_launch_count()) - l0) / float(steps_per_epoch * EPOCHS) if graph else None
    n_path = 0
    if hasattr(eng, "task_ranges") and cfg_name in ("shared_bottom", "mmoe", "ple"):
        n_path = sum(c for _, c in eng.task_ranges(0))
    rows = sum(sizes.values()) * EPOCHS
    macs = step_macs_per_row(eng)
    ach = 6.0 * macs * rows / dt / 1e12
    roofline = {"bound": "mfma", "scope": "whole step (every launch of a domain-step; the engine has no single dominant kernel)"
                if graph else "whole step (k_tower4<DX, FM> + k_wgrad + k_update)",
                "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "flops_per_row": 6 * macs, "launches_per_step": launches if graph else 3}
    print(json.dumps({"tower": cfg_name, "engine": "generic-layer (mamdr_graph_*)" if graph else "step kernels (mamdr_*)",
                      "value": steps_per_epoch * EPOCHS / dt, "unit": "domain-steps/s",
                      "us_per_domain_step": dt / (steps_per_epoch * EPOCHS) * 1e6, "batch": ds.batch_size, "roofline": roofline,
                      "params": int(eng.n_params), "params_on_a_step_path": int(n_path) or int(eng.n_params),
                      "model": {k: cfg["model"][k] for k in cfg["model"] if "hidden" in k or "expert" in k}}))
    eng.close()
