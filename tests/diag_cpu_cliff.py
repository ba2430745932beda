"""Why the CPU baseline's thread sweep falls off a cliff at the pinned core count (BENCH_r03: 4.3 ms / step at 32 threads,
903 ms at 64 on 64 pinned physical cores).  Times the same torch-CPU step (oracle/torch_ref.TorchCpuModel, Taobao-10 bs
1024) in fresh processes:  thread count x {OpenMP wait policy default / passive} x {GPU runtime loaded or not}.
usage: python tests/diag_cpu_cliff.py            (driver: spawns the cases)
       python tests/diag_cpu_cliff.py case <threads> <touch_gpu>
(a diagnostic run by hand; it times the oracle, so it lives under tests/)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(nt, touch_gpu):
    import numpy as np
    import torch
    import bench
    from mamdr_amd import synthetic
    if touch_gpu:
        torch.zeros(1, device="cuda")           # the HIP runtime's own threads exist, as in bench.py
    from oracle import rng as orng
    from oracle import torch_ref as tref
    from oracle import tower as otower
    g = synthetic.generate("taobao10", batch_size=1024, seed=123, splits=("train",))
    params = bench.init_params(g)
    params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
    d, cols, n, perm = bench._cpu_sample(g, 1024, orng.shuffle_perm)
    pinned, restore = bench._pin_to_one_socket()
    names = otower.param_names(False, 0, False)
    model = tref.TorchCpuModel(dict(params), names, tower="mlp", dropout=0.5, lr=1e-3)
    nt = len(pinned) + nt if nt <= 0 else nt    # 0 = every pinned core, -2 = two fewer
    torch.set_num_threads(nt)
    ts = []
    for s in range(8):
        idx = perm[(s % (n // 1024)) * 1024:(s % (n // 1024) + 1) * 1024]
        t = time.time()
        model.train_on_batch(cols["uid"][idx], cols["pid"][idx], cols["domain"][idx], cols["label"][idx])
        ts.append(time.time() - t)
    restore()
    print(json.dumps({"threads": nt, "pinned_cores": len(pinned), "touch_gpu": bool(touch_gpu),
                      "omp_wait_policy": os.environ.get("OMP_WAIT_POLICY", "default"),
                      "process_threads": len(os.listdir("/proc/self/task")),
                      "ms_per_step_median": round(float(np.median(ts[2:])) * 1e3, 2), "ms_max": round(max(ts[2:]) * 1e3, 2)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "case":
        case(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for touch in (1, 0):
            for policy in (None, "passive"):
                for nt in (32, -2, 0):
                    env = dict(os.environ)
                    if policy:
                        env["OMP_WAIT_POLICY"] = policy
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "case", str(nt), str(touch)], env=env,
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=600)
                    print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED %s %s %s" % (touch, policy, nt), flush=True)
