"""The embedding gather WHERE IT RUNS inside a training step: s_memtime stamps of the tower kernels'
gather phase -- kernel start -> every row of the tile in LDS -- from a -DMAMDR_STAMPS build (never shipped), median over the
workgroups of the last stamped launch, for the workloads whose gather is a phase of the tower kernel:

    amazon6   deepfm, trainable FULL tables (316 MB: beyond the 256 MiB infinity cache), bs 1,024   k_tower4<DX, FM, W1L>
    taobao30  mlp, frozen tables (66 MB), bs 4,096                                                   k_tower (16-row tiles)
    amazon13  star, trainable FULL tables (367 MB), bs 8,192                                         k_tower<train, 384>

(the k_wgrad_adam path of Taobao-10 gathers in k_pass_prep_multi, a launch of its own that bench.py times live).
Writes gpurun_out/<tag>/gather_in_step.json = {shape: {kernel, gather_phase_cycles, clock_ghz, tiles, note}}; copy it to
profiles/gather_in_step_latest.json (+ a dated copy) -- bench.py turns the phase time and ITS run's rows per launch into GB/s.

Usage (GPU box): python tools/r06_gather_in_step.py <out.json> [shape ...]
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from mamdr_amd import build as B      # noqa: E402

so = os.path.join(ROOT, "mamdr_amd", "build", "variants", "libstamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not (os.environ.get("MAMDR_STAMPS_PREBUILT") and os.path.exists(so)):
    procs, objs = [], []
    for src, extra in B.SOURCES:           # (parallel compiles: ~50 s instead of minutes)
        o = os.path.join("/tmp", "stamps_" + src.replace(".hip", ".o"))
        objs.append(o)
        procs.append(subprocess.Popen([B._hipcc()] + B.COMMON + extra + ["-DMAMDR_STAMPS", "-c", os.path.join(B.CSRC, src), "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so] + objs)
from mamdr_amd import _lib      # noqa: E402
_lib.LIB_PATH = so
import bench                    # noqa: E402  (workload table + initialisers; after LIB_PATH so that it loads the stamp build)
from mamdr_amd import engine, synthetic      # noqa: E402

CASES = {
    # shape: (tower, trainable, batch, row_scale, kernel as bench.py names it, rows per tile)
    "amazon6": ("deepfm", True, 1024, 0.01, "k_tower4<true, true, true, false, false>", 4),
    "taobao30": ("mlp", False, 4096, 1.0, "k_tower<true, 0, false, false>", 16),
    "amazon13": ("star", True, 8192, 0.04, "k_tower<true, 384, false, false>", 16),
}


def measure(shape):
    tower, trainable, bs, row_scale, kname, tile_rows = CASES[shape]
    g = synthetic.generate(shape, batch_size=bs, seed=123, row_scale=row_scale, splits=("train",))
    eng = bench.setup_engine(g, bs, trainable, tower)
    rs = np.random.RandomState(0)
    w = torch.from_numpy((rs.standard_normal(1 << 20) * 0.05).astype(np.float32)).to(eng.device)
    full = eng.new_vector()
    full.copy_(w.repeat(-(-full.numel() // w.numel()))[:full.numel()])       # any finite weights: the gather's timing does not care
    if tower == "star":
        from mamdr_amd.model_zoo.star import initial_tensors
        n_u, n_i = g["n_user"], g["n_item"]
        # (Keras initial values for everything but the two tables, which keep the random fill above)
        small = initial_tensors(np.random.RandomState(1), 8, 8, g["n_domain"], 128, (256, 128, 64), None, None)
        for name, (off, cnt) in eng.segments.items():
            if name not in ("user_emb", "item_emb"):
                full[off:off + cnt] = torch.from_numpy(np.asarray(small[name], np.float32).ravel()).to(eng.device)
    eng.set_weights(full)
    d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    n = eng.n_rows(d, "train")
    assert n >= 4 * bs, (shape, n)
    stamps = torch.zeros(65536 + 8192 + 4096, dtype=torch.int64, device=eng.device)
    eng.lib.mamdr_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.mamdr_debug_set_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
    perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
    phases, totals = [], []
    for rep in range(6):
        eng.train_steps(d, perm=perm, first_step=0, n_steps=3)
        torch.cuda.synchronize()
        tiles = -(-bs // 16) * (16 // tile_rows)
        for half in (0, 1):           # (the stamp buffer alternates between two steps: both halves hold full launches)
            st = stamps.cpu().numpy()[half * 16384:half * 16384 + tiles * 16].reshape(tiles, 16).astype(np.float64)
            if st[:, 0].min() > 0 and st[:, 1].min() > 0:
                phases.append(float(np.median(st[:, 1] - st[:, 0])))
                totals.append(float(np.median(st[:, 9] - st[:, 0])))
    clock = torch.cuda.get_device_properties(eng.device).clock_rate / 1e6 if hasattr(torch.cuda.get_device_properties(eng.device), "clock_rate") else 2.4
    eng.close()
    ph = float(np.median(phases[2:]))         # (the first launches run with cold code and tables)
    return {"kernel": kname, "gather_phase_cycles": ph, "kernel_cycles": float(np.median(totals[2:])), "clock_ghz": clock,
            "tiles": tiles, "rows_per_launch_here": bs, "table_bytes": (g["n_user"] + g["n_item"]) * 512,
            "note": "kernel start -> rows in LDS, %d-row tiles, %d launches stamped, phase %.0f of %.0f kernel cycles" % (
                tile_rows, len(phases), ph, float(np.median(totals[2:])))}


if __name__ == "__main__":
    out_path = sys.argv[1]
    shapes = sys.argv[2:] or list(CASES)
    out = {}
    for s in shapes:
        try:
            out[s] = measure(s)
        except Exception as e:      # noqa: BLE001
            out[s] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(s, json.dumps(out[s]), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
