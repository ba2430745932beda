"""Lanes on the GPU (mamdr_amd/parallel.LaneGroup, round 5): the ranks of the sharded epoch as host threads of
one process, one TowerEngine on one HIP stream each, their kernels overlapping on the device.  What must hold for that to
be the L-rank run: (i) the collectives between lanes are ordered by events on the lanes' streams -- no lane reads a
buffer before its owner's stream has produced it, no owner overwrites one before every reader's stream has read it;
(ii) contexts that run side by side share nothing on the device -- a lane's pass ends in the bits it ends in alone;
(iii) the tower tile a lane chooses (mamdr_set_tower_tile) is a parity-green step path of its own.  The end-to-end lane
run against its oracle twin is tests/test_gpu_e2e.py's `taobao10_mamdr_finetune_lanes2`; lane run == gloo 2-process run
bit for bit is tests/test_abi_and_parallel.py (CPU stand-in)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import rng as orng          # noqa: E402

F32 = np.float32


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _engine(g, batch, tile=None, seed=1):
    from mamdr_amd import engine
    eng = engine.TowerEngine(g["n_user"], g["n_item"], g["n_domain"], batch, dropout=0.5, tower_tile=tile)
    eng.bind_table("user_emb", g["tables"]["user_emb"])
    eng.bind_table("item_emb", g["tables"]["item_emb"])
    for d in range(g["n_domain"]):
        c = g["data"]["train"][d]
        eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    w = torch.from_numpy((np.random.RandomState(seed).standard_normal(eng.n_params) * 0.05).astype(F32)).to(eng.device)
    eng.set_weights(w)
    return eng


def test_lane_collectives_are_ordered_on_the_lanes_streams():
    """4 lanes, 40 rounds: every lane rewrites an 8 M-float vector on its stream (value = f(lane, round)), all-reduces it,
    broadcasts another, hands a third to its neighbour -- and checks the results ON THE DEVICE, with no host
    synchronisation in between (a missing event wait shows up as a stale round's value)."""
    _need_gpu()
    from mamdr_amd import parallel, synthetic
    g = synthetic.generate("taobao10", batch_size=256, seed=7, scale=0.02, splits=("train",))
    L, N, R = 4, 8 << 20, 40

    def fn(lane):
        eng = _engine(g, 256)
        parallel.lane_adder(eng)                  # sums through the library's elementwise kernel (mamdr_merge, plus)
        dev = eng.device
        a, b = torch.empty(N, device=dev), torch.empty(N, device=dev)
        vec = {k: torch.empty(N, device=dev) for k in range(L)}
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        for r in range(R):
            a.fill_(float(lane + 1 + 10 * r))
            parallel.all_reduce(a)
            bad += (a != float(sum(k + 1 + 10 * r for k in range(L)))).sum()
            b.fill_(float(100 * lane + r))
            parallel.broadcast(b, r % L)
            bad += (b != float(100 * (r % L) + r)).sum()
            vec[lane].fill_(float(1000 * lane + r))
            parallel.lanes().transfer(lane, vec, [(k, k, (k + 1) % L) for k in range(L)])
            src = (lane - 1) % L
            bad += (vec[src] != float(1000 * src + r)).sum()
        n_bad = int(bad.item())
        eng.close()
        return n_bad
    assert parallel.LaneGroup(L).run(fn) == [0] * L


@pytest.mark.parametrize("tile", [None, 16, 4])
def test_lanes_running_side_by_side_end_in_the_bits_of_their_solo_runs(tile):
    """three lanes train three different domains from three different models at the same time (their towers, weight
    gradients and optimiser steps overlap on the device), then each pass is repeated ALONE on a fresh engine: the same bits.
    tile None = what an engine picks on its own (4-row towers in a group of three lanes; 16-row from four lanes on)."""
    _need_gpu()
    from mamdr_amd import parallel, synthetic
    g = synthetic.generate("taobao10", batch_size=1024, seed=7, scale=0.5, splits=("train",))
    order = sorted(range(10), key=lambda d: -g["data"]["train"][d]["uid"].shape[0])
    L = 3

    def one(lane, solo):
        d = order[lane]
        n = g["data"]["train"][d]["uid"].shape[0]
        eng = _engine(g, 1024, tile, seed=10 + lane)
        perm = torch.from_numpy(orng.shuffle_perm(n, 10000, seed=lane)).to(eng.device)
        if not solo:
            parallel.barrier()                    # (start together)
        for _ in range(3):
            eng.train_steps(d, perm=perm, lr=1e-3)
        # ... the finetune stage's steps (SGD: the slab path's k_update) and an evaluation (AUC histogram, loss) as well
        eng.train_steps(d, perm=perm, lr=1e-3, optimizer="sgd")
        ev = eng.evaluate(d, "train")
        w = eng.get_weights().cpu().numpy().copy()
        eng.close()
        return w, ev
    together = parallel.LaneGroup(L).run(lambda lane: one(lane, False))
    for lane in range(L):
        alone, ev = one(lane, True)
        assert np.array_equal(together[lane][0].view(np.uint32), alone.view(np.uint32)), (tile, lane)
        assert together[lane][1] == ev, (tile, lane, together[lane][1], ev)
        assert not np.array_equal(together[lane][0], together[(lane + 1) % L][0])


def test_tower_tile_of_an_engine():
    """mamdr_set_tower_tile: 4 / 16 / automatic are accepted, anything else is an error; an engine built on a lane of a group
    of four takes the 16-row tower, whose pass stays within rounding of the 4-row tower's (two orders of the same split-K
    sums: every weight within 2 % of the pass's k * lr, the typical one far closer)."""
    _need_gpu()
    from mamdr_amd import _lib as L, parallel, synthetic
    g = synthetic.generate("taobao10", batch_size=1024, seed=7, scale=0.2, splits=("train",))
    d = max(range(10), key=lambda k: g["data"]["train"][k]["uid"].shape[0])
    n = g["data"]["train"][d]["uid"].shape[0]
    out = {}
    for tile in (4, 16):
        eng = _engine(g, 1024, tile)
        with pytest.raises(L.MamdrError):
            eng.set_tower_tile(8)
        eng.set_tower_tile(tile)
        w0 = eng.get_weights().cpu().numpy().copy()
        k = eng.train_steps(d, perm=torch.from_numpy(orng.shuffle_perm(n, 10000, seed=3)).to(eng.device), lr=1e-3)
        out[tile] = (eng.get_weights().cpu().numpy().copy(), w0, k, eng.tower_tile(1024))
        eng.close()
    (w4, w0, k, t4), (w16, _, _, t16) = out[4], out[16]
    assert (t4, t16) == (4, 16)
    diff = np.abs(w4 - w16)
    assert k >= 5 and np.abs(w4 - w0).max() > 0.5 * k * 1e-3
    assert float(np.mean(diff > 0.02 * k * 1e-3)) < 2e-3 and float(np.median(diff)) < 1e-3 * k * 1e-3, (diff.max(), np.median(diff))

    def fn(lane):
        eng = _engine(g, 1024)
        t = eng.tower_tile(1024)
        eng.close()
        return t
    assert parallel.LaneGroup(4).run(fn) == [16] * 4 and parallel.LaneGroup(2).run(fn) == [4] * 2


@pytest.mark.parametrize("cfg_file,name,lanes", [
    ("Taobao-10/deepctr_DN+DR.json", "ccpm_meta_reptile", 2),                            # generic-layer engine on lanes
    ("Taobao-10/star_taobao.json", "star_meta_mamdr", 2),                                # Star: TailSync through the lanes' collectives
    ("Taobao-10/deepctr_DN+DR.json", "deepfm_meta_domain_negotiation_finetune", 3),
    ("Taobao-10/deepctr_DN+DR.json", "nfm_meta_mamdr", 4),                               # (four lanes: the nfm tower keeps its four-row tile)
    ("Taobao-10/deepctr_DN+DR.json", "wdl_meta_mamdr", 4),                               # (... wdl takes the 16-row tower)
])
def test_run_entry_on_lanes_with_the_other_towers(tmp_path, cfg_file, name, lanes):
    """run.py's entry with train.lanes for towers beyond the frozen-table mlp: every lane builds its own engine of that kind
    on its own stream, the sharded wrappers' collectives (and, for the Star tower, the step-weighted combination of the
    tensors outside theta / phi) run between the lanes, every domain is reported and the model has learnt."""
    _need_gpu()
    import copy
    import json
    import os
    from mamdr_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", cfg_file)) as f:
        cfg = copy.deepcopy(json.load(f))
    cfg["model"]["name"] = name
    cfg["train"].update(epoch=3, patience=2, sample_num=2, meta_learning_rate=0.5, lanes=lanes,
                        result_save_path=str(tmp_path / "result"), checkpoint_path=str(tmp_path / "ckpt"))
    cfg["dataset"].update(batch_size=256, synthetic="taobao10", synthetic_scale=0.1)
    built = []
    avg_loss, avg_auc, domain_loss, domain_auc = cli.main(cfg, on_model=built.append)
    assert len(built) == lanes and len(domain_auc) == 10 and np.isfinite(avg_loss)
    assert avg_auc > 0.52, (name, avg_auc)
