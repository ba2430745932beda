"""A handful of training steps on one domain, for hardware-counter passes (rocprofv3 --pmc ... -- python3 tools/pmc_steps.py).
usage: python tools/pmc_steps.py [shape] [batch] [steps]
  taobao10 / taobao30  mlp tower, frozen tables
  amazon6              deepfm tower, trainable tables (1 % of the rows: the counters of a launch do not depend on how many
                       rows an epoch has; the tables keep their full size)
  amazon13             star tower, trainable tables (4 % of the rows), Keras initial values for the Star block"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mamdr_amd import _lib
if os.environ.get("MAMDR_LIB_PATH"):          # a diagnostic build of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.environ["MAMDR_LIB_PATH"]
from mamdr_amd import engine, synthetic
shape = sys.argv[1] if len(sys.argv) > 1 else "taobao30"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rs = np.random.RandomState(0)
if shape in ("amazon6", "amazon13"):
    import bench
    tower, row_scale = ("deepfm", 0.01) if shape == "amazon6" else ("star", 0.04)
    g = synthetic.generate(shape, batch_size=bs, seed=123, row_scale=row_scale, splits=("train",))
    eng = bench.setup_engine(g, bs, True, tower)
    w = torch.from_numpy((rs.standard_normal(1 << 20) * 0.05).astype(np.float32)).to(eng.device)
    full = eng.new_vector()
    full.copy_(w.repeat(-(-full.numel() // w.numel()))[:full.numel()])
    if tower == "star":
        from mamdr_amd.model_zoo.star import initial_tensors
        small = initial_tensors(np.random.RandomState(1), 8, 8, g["n_domain"], 128, (256, 128, 64), None, None)
        for name, (off, cnt) in eng.segments.items():
            if name not in ("user_emb", "item_emb"):
                full[off:off + cnt] = torch.from_numpy(np.asarray(small[name], np.float32).ravel()).to(eng.device)
    eng.set_weights(full)
    d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
else:
    g = synthetic.generate(shape, batch_size=bs, seed=123)
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], bs, dropout=0.5)
    eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
    d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    eng.set_weights(torch.from_numpy((rs.standard_normal(eng.n_params) * 0.05).astype(np.float32)).to(eng.device))
n = eng.n_rows(d, "train")
perm = torch.from_numpy(engine.shuffle_perm(n, 10000, 1)).to(eng.device)
eng.train_steps(d, perm=perm, first_step=0, n_steps=min(steps, -(-n // bs)))
torch.cuda.synchronize()
print("done", shape, bs, min(steps, -(-n // bs)))
