"""bench.py -- domain-steps/sec of the MAMDR hot path on N MI355X GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): mlp_meta_mamdr, Taobao-10 shaped synthetic
click logs, batch 1024, pretrained 128-d tables frozen, Adam lr 1e-3, dropout 0.5,
meta lr 0.1, 5 sampled support domains + the query domain (config/Taobao-10/
deepctr_DN+DR.json).  ONE bench "step" = one full MAMDR meta-epoch (DN phase over all
domains, then the DR phase: for every query domain and every support domain, a pass
over the support domain and a pass over the query domain, with all outer updates),
i.e. `domain_steps_per_epoch` inner optimisation steps (gather + MLP fwd/bwd + BCE +
Adam).  value = inner domain-steps executed by all ranks / wall time of the K epochs
(max over ranks), inputs resident in HBM, per-pass shuffles generated and uploaded
inside the timed region, no eval inside it.

N > 1: query domains (DR) and the DN sub-sequences are sharded over the ranks with a
single all-reduce of the DN displacement per epoch (mamdr_amd/parallel.py); total
work is fixed -> "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32-input MFMA (= vector peak)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec
# algorithmic flops per batch row of k_tower<train> (kernel table):
#   forward 384*256 + 256*128 + 128*64 + 64 MACs, backward chain dz3->dz2->dz1: 64*128 + 128*256 MACs
#   (the per-row contraction with W0[256:384] is gone: the domain-table gradient uses linearity)
TOWER_TRAIN_FLOPS_PER_ROW = 2 * ((384 * 256 + 256 * 128 + 128 * 64 + 64) + (64 * 128 + 128 * 256))


def tower_flops_per_row(dx_width):
    """+ the input-gradient contraction dz1 . W0[0:dx_width, :]^T of the towers whose tables train: 256 columns
    ([user | item] rows, deepctr towers) or all 384 (Star: PartitionedNorm's backward needs d loss / d x)."""
    return TOWER_TRAIN_FLOPS_PER_ROW + 2 * 256 * dx_width
GATHER_BYTES_PER_ROW = 3 * 128 * 4 * 2 + 16   # read 3 rows + write 384 floats + 4 index/label words

WORKLOADS = {
    "taobao10": dict(shape="taobao10", batch=1024, name="mlp_meta_mamdr Taobao-10 bs=1024 (frozen pretrained tables)"),
    "taobao30": dict(shape="taobao30", batch=4096, name="mlp_meta_mamdr Taobao-30 bs=4096 (frozen pretrained tables)"),
    # trainable 128-d tables (79 M parameters): every step ends with TF1's dense Adam over all rows
    # (BASELINE.json configs[2]: DeepFM tower under Domain Negotiation)
    "amazon6": dict(shape="amazon6", batch=1024, emb_trainable=True, wrapper="dn", row_scale=0.1, tower="deepfm",
                    name="deepfm_meta_domain_negotiation Amazon-6 bs=1024 (trainable tables, full-size tables, "
                         "10% of the rows per epoch)"),
}
# (BASELINE.json configs[4]: Star tower under MAMDR, theta / phi over the tables + shared kernels / biases)
WORKLOADS["amazon13"] = dict(shape="amazon13", batch=8192, emb_trainable=True, wrapper="mamdr", row_scale=0.1,
                             tower="star", name="star_meta_mamdr Amazon-13 bs=8192 (PartitionedNorm + StarFCN, "
                                                "trainable tables, full-size tables, 10% of the rows per epoch)")
TRAIN = dict(learning_rate=1e-3, meta_learning_rate=0.1, sample_num=5, add_query_domain=True, dropout=0.5,
             merged_method="plus", shuffle_buffer_size=10000, seed=123)


def init_params(g, seed=1024):
    """random-init weights of the reference architecture (deepctr.py:118-136 initialisers)."""
    rs = np.random.RandomState(seed)
    p = {"domain_emb": (rs.standard_normal((g["n_domain"], 128)) * 1e-4).astype(np.float32)}
    dims = (384, 256, 128, 64)
    for l in range(3):
        s = np.sqrt(2.0 / (dims[l] + dims[l + 1]))
        p["W%d" % l] = (np.clip(rs.standard_normal((dims[l], dims[l + 1])), -2, 2) * s).astype(np.float32)
        p["b%d" % l] = np.zeros(dims[l + 1], np.float32)
    p["wo"] = (np.clip(rs.standard_normal((64, 1)), -2, 2) * np.sqrt(2.0 / 65)).astype(np.float32)
    p["gb"] = np.zeros(1, np.float32)
    # DeepFM 1-d linear tables start at zero (deepctr get_linear_logit); ignored by the mlp tower
    p["lin_user"] = np.zeros(g["n_user"], np.float32)
    p["lin_item"] = np.zeros(g["n_item"], np.float32)
    p["lin_domain"] = np.zeros(g["n_domain"], np.float32)
    return p


def setup_engine(g, batch, emb_trainable=False, tower="mlp"):
    from mamdr_amd import engine
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], batch, dropout=TRAIN["dropout"],
                             emb_trainable=emb_trainable, tower=tower)
    if not emb_trainable:
        eng.bind_table("user_emb", g["tables"]["user_emb"])
        eng.bind_table("item_emb", g["tables"]["item_emb"])
    for d in range(g["n_domain"]):
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    return eng


def cpu_baseline(g, batch, budget_s=15.0, params=None, emb_trainable=False, tower="mlp"):
    """the oracle (numpy restatement of the TF1.12 path) timed on this box's host cores on
    the first domain-steps of the same workload; TF itself is not installable."""
    from oracle import rng as orng
    from oracle import tower as otower
    if params is None:
        params = init_params(g)
        params["user_emb"], params["item_emb"] = g["tables"]["user_emb"], g["tables"]["item_emb"]
    if tower == "star":
        from oracle import star as ostar
        model = ostar.OracleStar(params, emb_trainable=emb_trainable, lr=TRAIN["learning_rate"])
    else:
        model = otower.OracleModel(params, emb_trainable=emb_trainable, dropout=TRAIN["dropout"],
                                   lr=TRAIN["learning_rate"], tower=tower)
    d = max(range(g["n_domain"]), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    cols = g["data"]["train"][d]
    n = cols["uid"].shape[0]
    perm = orng.shuffle_perm(n, TRAIN["shuffle_buffer_size"], 1)
    steps, t0 = 0, time.time()
    while time.time() - t0 < budget_s:
        for s in range(-(-n // batch)):
            idx = perm[s * batch:(s + 1) * batch]
            model.train_on_batch(cols["uid"][idx], cols["pid"][idx], cols["domain"][idx], cols["label"][idx])
            steps += 1
            if time.time() - t0 >= budget_s:
                break
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": steps / dt, "unit": "domain-steps/s", "cores": int(cores), "kind": "port",
            "sample": "%d inner steps (bs=%d, domain %d of the same synthetic workload) of the numpy fp32 oracle "
                      "in %.1f s; restatement of the TF1.12 CPU path, not TF" % (steps, batch, d, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed MAMDR meta-epochs")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="taobao10", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # MAMDR_BENCH_SHARE_GPU=1 (testing on a 1-GPU box only): all ranks use device 0 and gloo
        share = os.environ.get("MAMDR_BENCH_SHARE_GPU") == "1"
        torch.cuda.set_device(0 if share else local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:
                dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from mamdr_amd import _lib as L
    from mamdr_amd import meta, parallel, plan as mplan, synthetic

    wl = WORKLOADS[args.workload]
    batch = int(os.environ.get("MAMDR_BENCH_BATCH", wl["batch"]))      # (exploration only: the named config fixes it)
    trainable = bool(wl.get("emb_trainable"))
    g = synthetic.generate(wl["shape"], batch_size=batch, seed=TRAIN["seed"], row_scale=wl.get("row_scale", 1.0))
    D = g["n_domain"]
    tower = wl.get("tower", "mlp")
    eng = setup_engine(g, batch, trainable, tower)

    def full_params(seed=1024):
        if tower == "star":      # Keras defaults of the Star layers (mamdr_amd/model_zoo/star.py)
            from mamdr_amd.model_zoo.star import initial_tensors
            return initial_tensors(np.random.RandomState(seed), g["n_user"], g["n_item"], D, 128, (256, 128, 64),
                                   None if trainable else g["tables"]["user_emb"],
                                   None if trainable else g["tables"]["item_emb"])
        p = init_params(g, seed)
        if trainable:      # Amazon: no pretraining, N(0, 1e-4^2) tables (deepctr.py:115 SparseFeat default)
            rs_ = np.random.RandomState(seed + 7)
            p["user_emb"] = (rs_.standard_normal((g["n_user"], 128)) * 1e-4).astype(np.float32)
            p["item_emb"] = (rs_.standard_normal((g["n_item"], 128)) * 1e-4).astype(np.float32)
        return p
    sizes = [eng.n_rows(d, "train") for d in range(D)]
    steps_per_domain = [-(-n // batch) for n in sizes]
    full0 = eng.pack(full_params())
    eng.set_weights(full0)                 # tensors outside theta (Star: PN, specific kernels, output unit)
    theta = full0[:eng.n_meta].clone()
    del full0
    owner = parallel.lpt_partition(sizes, world)
    # phi_d starts as a second random init of the whole model (mamdr.py:31-33)
    wrapper = wl.get("wrapper", "mamdr")
    phis = {d: eng.pack(full_params(seed=2000 + d))[:eng.n_meta].clone() for d in range(D) if owner[d] == rank} \
        if wrapper == "mamdr" else {}
    bufs = {"delta": eng.new_vector(meta=True), "zero": eng.new_vector(meta=True), "merged": eng.new_vector(meta=True)}
    planner = mplan.EpochPlanner(range(D), TRAIN["sample_num"], TRAIN["add_query_domain"], True, TRAIN["seed"])
    shuffler = mplan.PassShuffler(sizes, TRAIN["shuffle_buffer_size"], TRAIN["seed"] + rank)

    def epoch():
        if wrapper == "dn":               # Domain Negotiation only (domain_negotiation.py:37-88)
            p = planner.next_epoch(with_dr=False)
            tr = []
            parallel.dn_phase_sharded(eng, meta, theta, parallel.shard_plan(p, owner, rank)["seq"], shuffler, batch,
                                      TRAIN["learning_rate"], TRAIN["meta_learning_rate"], tr, bufs["delta"],
                                      bufs["zero"])
            eng.set_weights(theta)
            return tr, mplan.plan_steps(p, steps_per_domain)
        p = planner.next_epoch()          # same seed on every rank -> same global plan
        tr = parallel.mamdr_epoch_sharded(eng, meta, theta, phis, p, owner, shuffler, batch,
                                          TRAIN["learning_rate"], TRAIN["meta_learning_rate"], bufs,
                                          TRAIN["merged_method"])
        return tr, mplan.plan_steps(p, steps_per_domain)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        epoch()
    barrier()
    t0 = time.perf_counter()
    local_steps, global_steps, local_passes = 0, 0, 0
    for _ in range(args.steps):
        tr, b = epoch()
        local_steps += sum(t[2] for t in tr)
        local_passes += len(tr)
        global_steps += b
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, float(local_steps), float(local_passes)], dtype=torch.float64, device=eng.device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        assert int(round(float(t[1]))) == global_steps, (float(t[1]), global_steps)
        local_passes = int(round(float(t[2])))

    # ---- per-kernel device time (HIP events on the launch stream) over one more epoch of
    #      the same workload; reported for the dominant kernel, k_tower<train>
    roofline, gather_info, kernels, sweep_info, table_info = None, None, {}, None, None
    if not args.no_profile:
        eng.profile(True)
        eng.profile_reset()
        epoch()                       # fills the library's event pool: the measured epoch below creates no events
        eng.profile_reset()
        prof_trace, _ = epoch()
        for k in (L.KERNEL_FWD_BWD, L.KERNEL_WGRAD, L.KERNEL_UPDATE, L.KERNEL_EMB_SWEEP):
            ms, cnt = eng.profile_read(k)
            if cnt:
                kernels[L.KERNEL_NAMES[k]] = {"launches": cnt, "avg_us": ms / max(cnt, 1) * 1e3}
        dense_adam = os.environ.get("MAMDR_DENSE_ADAM", "0") not in ("", "0")
        if trainable and not dense_adam:
            # default: lazy replay of TF1's dense table Adam (csrc/emb_kernels.hip) -- per step only the rows of
            # the batch move through HBM; MAMDR_DENSE_ADAM=1 measures the per-step sweep instead
            ms, cnt = eng.profile_read(L.KERNEL_EMB_SWEEP)
            table_info = {"mode": "lazy (bit-identical to the per-step dense sweep)", "kernel": "k_emb_reduce (+ Adam step of the touched rows)",
                          "avg_us": ms / max(cnt, 1) * 1e3, "launches": cnt}
        if trainable and dense_adam:
            # HBM-bound dense optimiser pass: 24 B per table element (read p, m, v; write p, m, v) + 4 B
            # of row map per 512-B row, one launch per step over both tables
            ms, cnt = eng.profile_read(L.KERNEL_EMB_SWEEP)
            # (DeepFM's 1-d linear tables ride in the same launches: 24 more bytes per row, < 1 %, not counted)
            sweep_bytes = (g["n_user"] + g["n_item"]) * (128 * 24 + 4) * cnt
            ach = sweep_bytes / (ms * 1e-3) / 1e9
            sweep_info = {"kernel": "k_emb_sweep", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                          "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": pmc_traffic("k_emb_sweep<0>"),
                          "avg_us": ms / max(cnt, 1) * 1e3, "launches": cnt,
                          "bytes_per_step": (g["n_user"] + g["n_item"]) * (128 * 24 + 4)}
        roofline_ms, cnt = eng.profile_read(L.KERNEL_FWD_BWD)
        eng.profile(False)
        eng.profile_reset()
        # every launch is one batch; a pass of n_steps launches covers min(rows of the domain, n_steps*batch) rows
        prof_rows = sum(min(sizes[d], n * batch) for (_, d, n) in prof_trace)
        assert cnt == sum(n for (_, _, n) in prof_trace)
        # batches <= 2048 rows launch the 4-row-tile kernel (mamdr_api.hip: use4), template <DX, FM>
        use4 = batch <= 2048 and tower != "star" and os.environ.get("MAMDR_TOWER_TILE", "") != "16"
        fm = ", true>" if tower == "deepfm" else ", false>"
        kname = ("k_tower4<true" if trainable else "k_tower4<false") + fm if use4 else \
            ("k_tower<true, 256" if trainable else "k_tower<true, 0") + fm
        if tower == "star":
            kname = "k_tower<true, 384, false>"
        roofline = finish_roofline(kname, roofline_ms, cnt, prof_rows,
                                   tower_flops_per_row(384 if tower == "star" else (256 if trainable else 0)))
        roofline["rocprofv3_avg_us"] = rocprof_avg_us(kname, wl["shape"])
        # gather kernel on a pass-sized batch (largest domain, shuffled order)
        dbig = max(range(D), key=lambda k: sizes[k])
        perm = torch.from_numpy(shuffler(dbig)).to(eng.device)
        out = torch.empty((sizes[dbig], 384), dtype=torch.float32, device=eng.device)
        for _ in range(3):
            eng.gather(dbig, "train", perm=perm, out=out)
        eng.profile(True)
        eng.profile_reset()
        for _ in range(20):
            eng.gather(dbig, "train", perm=perm, out=out)
        gms, gcnt = eng.profile_read(L.KERNEL_GATHER)
        eng.profile(False)
        eng.profile_reset()
        gbytes = sizes[dbig] * GATHER_BYTES_PER_ROW
        gach = gbytes / (gms / gcnt * 1e-3) / 1e9
        gather_info = {"kernel": "k_gather", "bound": "hbm", "achieved": gach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": gach / PEAK_HBM_GBS, "traffic": None, "rows_per_launch": sizes[dbig],
                       "bytes_per_row": GATHER_BYTES_PER_ROW,
                       "note": "standalone pass-sized gather of the same tile code the step kernel uses; "
                               "Taobao tables (15.7 MB) are cache-resident, the 48 MB output is not"}
    result = None
    if rank == 0:
        cpu = cpu_baseline(g, batch, args.cpu_budget, full_params() if trainable else None, trainable, tower) \
            if args.cpu_budget > 0 else None
        result = {
            "metric": "domain-steps/sec", "value": global_steps / dt, "unit": "domain-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "global_batch": batch, "domains": D,
                       "domain_steps_per_epoch": global_steps / args.steps,
                       "step_definition": "one MAMDR meta-epoch (DN + DR phases, all outer updates)",
                       "parallelism": "domain-sharded x%d, 1 all-reduce of the DN displacement per epoch" % world
                       if world > 1 else "single GPU"},
            "us_per_domain_step": dt / global_steps * 1e6 * world,
            # also domain-passes/sec (one pass = one domain's re-initialised iterator run to its end or
            # cap) and the epoch time (= ms_per_step: a bench step is one meta-epoch)
            "domain_passes_per_sec": local_passes / dt, "epoch_time_ms": dt / args.steps * 1e3,
            "roofline": sweep_info if (sweep_info and sweep_info["avg_us"] * 2 > (roofline or {}).get("avg_us", 0))
            else roofline, "tower": roofline, "table_update": table_info or sweep_info, "gather": gather_info, "kernels_avg_us": kernels, "cpu_baseline": cpu,
        }
        if cpu:
            result["gpu_over_cpu"] = result["value"] / cpu["value"]
        print(json.dumps(result))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return result


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (tools/summarize_pmc.py, gfx950 corrections applied); None if no summary is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")
    try:
        with open(path) as f:
            return json.load(f)[kernel_key]["hbm_bytes_per_launch"]
    except Exception:
        return None


def rocprof_avg_us(kernel, shape):
    """average duration of `kernel` in the newest committed rocprofv3 summary of this workload
    (profiles/r*_kernel_stats_<shape>.csv), for comparison with the HIP-event average measured live
    (events bracket single launches and add ~2 us of inter-command gap); None if no summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_%s.csv" % shape)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if kernel in row["Name"]:
                    return float(row["AverageNs"]) / 1e3
    except Exception:
        pass
    return None


def finish_roofline(kernel, total_ms, launches, rows, flops_per_row=TOWER_TRAIN_FLOPS_PER_ROW):
    """achieved = algorithmic flops of all profiled launches / their summed device time
    (= flops per average launch / average launch duration)."""
    flops = rows * flops_per_row
    ach = flops / (total_ms * 1e-3) / 1e12
    return {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(kernel),
            "traffic_unit": "HBM bytes per launch (profiles/pmc_hbm_latest.json: separate rocprofv3 --pmc passes)",
            "launches": launches,
            "avg_us": total_ms / max(launches, 1) * 1e3, "rows_per_launch": rows / max(launches, 1),
            "flops_per_row": flops_per_row}


if __name__ == "__main__":
    main()
