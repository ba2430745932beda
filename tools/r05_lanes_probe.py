"""round 5 probe: independent step chains (the query domains of the DR phase, the ranks' shares of a sharded epoch) run as LANES
on one GPU -- one TowerEngine per lane, each on its own HIP stream, passes issued round-robin in chunks from one host thread.
Question: how many domain-steps/s do L concurrent lanes reach together, against the one dependent chain of the headline?
  python tools/r05_lanes_probe.py [workload=taobao10] [batch=1024] [lanes=1,2,3,4] [chunk=8]
MAMDR_T4 (0/1) etc. select the tower as for any run.  No oracle, no parity: a rate measurement."""
import sys
import time

import torch

sys.path.insert(0, ".")
from mamdr_amd import engine, synthetic        # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "taobao10"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    lane_counts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4").split(",")]
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    g = synthetic.generate(workload, batch_size=batch, seed=123, scale=1.0)
    D = g["n_domain"]
    dev = torch.device("cuda:0")
    order = sorted(range(D), key=lambda d: -g["data"]["train"][d]["uid"].shape[0])
    tables = g["tables"]
    cols = {d: g["data"]["train"][d] for d in range(D)}
    for L in lane_counts:
        lanes = []
        for l in range(L):
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                eng = engine.TowerEngine(g["n_user"], g["n_item"], D, batch, dropout=0.5, emb_trainable=False)
                eng.bind_table("user_emb", tables["user_emb"])
                eng.bind_table("item_emb", tables["item_emb"])
                for d in range(D):
                    c = cols[d]
                    eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
                w = torch.randn(eng.n_params, device=dev) * 0.05
                eng.set_weights(w)
            lanes.append((s, eng))
        torch.cuda.synchronize()
        # pass by pass, `chunk` steps per call, lanes in turn
        share = [order[l:] + order[:l] for l in range(L)]       # every lane walks every domain: equal work, different order
        plans = []
        for l in range(L):
            calls = []
            for d in share[l]:
                n = -(-g["data"]["train"][d]["uid"].shape[0] // batch)
                calls += [(d, k, min(chunk, n - k)) for k in range(0, n, chunk)]
            plans.append(calls)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 0
            for i in range(max(len(p) for p in plans)):
                for l, (s, eng) in enumerate(lanes):
                    if i < len(plans[l]):
                        d, k, n = plans[l][i]
                        with torch.cuda.stream(s):
                            eng.train_steps(d, perm=None, first_step=k, n_steps=n, lr=1e-3)
                        steps += n
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                print("%s bs %d lanes %d chunk %d: %d steps in %.2f ms (host issue %.2f ms) = %.0f domain-steps/s, %.2f us/step"
                      % (workload, batch, L, chunk, steps, dt * 1e3, t_host * 1e3, steps / dt, dt / steps * 1e6), flush=True)
        for s, eng in lanes:
            eng.close()


if __name__ == "__main__":
    main()
