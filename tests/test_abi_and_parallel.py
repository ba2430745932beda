"""C-ABI export check (no GPU calls) and the N>1 path on CPU (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    from mamdr_amd import _lib
    header = open(os.path.join(ROOT, "include", "mamdr_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mamdr_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mamdr_abi_version() == _lib.ABI_VERSION
    # error path without a device: bad config is rejected before any HIP call
    import ctypes as C
    cfg = _lib.Config(_lib.ABI_VERSION, _lib.TOWER_MLP, 10, 10, 2, 64, (C.c_int32 * 3)(256, 128, 64), 1024, 0, 0.5,
                      1e-5, 1e-5, 0.9, 0.999, 1e-8)
    h = C.c_void_p()
    assert lib.mamdr_create(C.byref(cfg), None, C.byref(h)) == _lib.EINVAL
    assert b"emb_dim 128" in lib.mamdr_last_error()
    cfg.emb_dim, cfg.tower = 128, 7
    assert lib.mamdr_create(C.byref(cfg), None, C.byref(h)) == _lib.EINVAL
    assert b"unknown tower kind" in lib.mamdr_last_error()
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.ENOTBUILT)


def test_no_kernel_of_the_shipped_library_spills_to_scratch():
    """Every gfx950 kernel of libmamdr_hip.so has a private segment of 0 bytes (k_pcgrad's 80-byte local table excepted):
    read from the code objects' own metadata (the clang offload bundles inside the .so -> llvm-readelf --notes).  Round 6 met
    the failure this guards against: two call sites of one device body made the compiler index the by-value argument
    structs dynamically -- 296 B of scratch per lane, Amazon-13's step 165 -> 207 us, every parity test still green."""
    import struct
    import tempfile
    from mamdr_amd import _lib
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    _lib.load()
    data = open(_lib.LIB_PATH, "rb").read()
    sizes, vgprs = {}, {}
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        i = m.start()
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" not in triple or not sz:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[i + o:i + o + sz])
                f.flush()
                notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for name, size in re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", notes):
                sizes[name] = int(size)
            for blk in notes.split("- .agpr_count:")[1:]:
                vgprs[re.search(r"\.name:\s+(\S+)", blk).group(1)] = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    assert len(sizes) >= 100, len(sizes)          # (111 kernels at the end of round 6)
    spilled = {k: v for k, v in sizes.items() if v and "k_pcgrad" not in k}
    assert not spilled, spilled
    assert all(v <= 128 for v in sizes.values()), {k: v for k, v in sizes.items() if v > 128}
    # the 384-wide Star tower runs TWO 8-wave tiles per CU (512 VGPRs per SIMD lane / 4 waves): it must stay within 128
    # (64.9 -> 72 us at 8,192 rows when it did not)
    star = [v for k, v in vgprs.items() if "k_towerILb1ELi384ELb0ELb0E" in k]
    assert len(star) == 1 and star[0] <= 128, star


def test_environment_switch_registry_is_complete():
    """ONE table of environment switches (csrc/env_registry.h, mamdr_env_switches): every MAMDR_* name that any source file
    reads from the environment is listed, a name nobody reads is reported (mamdr_env_unknown), and the library never
    aborts the process."""
    from mamdr_amd import _lib
    table = _lib.env_switches()
    names = [t[0] for t in table]
    assert len(names) == len(set(names)) and all(len(t) == 3 and t[2] for t in table)
    exact = {n for n in names if not n.endswith("*")}
    prefixes = tuple(n[:-1] for n in names if n.endswith("*"))
    read = set()
    pats = (r'getenv\(\s*"(MAMDR_[A-Z0-9_]+)"', r'environ(?:\.get|\.setdefault)?[\(\[]\s*"(MAMDR_[A-Z0-9_]+)"',
            r'\$\{?(MAMDR_[A-Z0-9_]+)')
    for top in ("mamdr_amd", "tools", "tests", "bench.py", "run.py", "__graft_entry__.py"):
        path = os.path.join(ROOT, top)
        files = [path] if os.path.isfile(path) else [os.path.join(b, f) for b, _, fs in os.walk(path) for f in fs
                                                      if f.endswith((".py", ".hip", ".h", ".sh"))]
        for f in files:
            if os.path.basename(f) == "test_abi_and_parallel.py":
                continue
            src = open(f, errors="replace").read()
            for pat in pats:
                read.update(re.findall(pat, src))
    read = {n for n in read if not n.endswith("_")} | {"MAMDR_NFM_ENGINE"}     # ("MAMDR_%s_ENGINE" % tower: pnn, nfm)
    missing = sorted(n for n in read if n not in exact and not n.startswith(prefixes))
    assert not missing, "environment switches read but not in csrc/env_registry.h: %s" % missing
    # an unknown name is counted (and reported on stderr) by a fresh process; a known one and a prefixed one are not
    code = "from mamdr_amd import _lib; print('unknown', _lib.load().mamdr_env_unknown())"
    env = dict(os.environ, MAMDR_NO_SUCH_SWITCH="1", MAMDR_TOWER_TILE="0", MAMDR_BENCH_SKIP_AMAZON6="1")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "unknown 1" in out.stdout and "MAMDR_NO_SUCH_SWITCH" in out.stderr and "MAMDR_TOWER_TILE" not in out.stderr
    # no abort() / exit() in the library, no unsynchronised one-time flags (function-local statics carry initialisers)
    for f in os.listdir(os.path.join(ROOT, "mamdr_amd", "csrc")):
        src = re.sub(r"//.*", "", open(os.path.join(ROOT, "mamdr_amd", "csrc", f)).read())
        assert not re.search(r"\babort\s*\(|\bexit\s*\(", src), f
        assert not re.search(r"^\s+static\s+(int|bool)\s+\w+\s*=\s*(0|-1|false)\s*;", src, flags=re.M), f


def test_product_path_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under mamdr_amd/ or tools/, nor run.py, may import it;
    bench.py only inside its cpu_baseline leg, __graft_entry__ only inside smoke()."""
    bad = []
    for top in ("mamdr_amd", "tools"):
        for base, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(base, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(top, f))
    assert not bad, bad
    assert "oracle" not in open(os.path.join(ROOT, "run.py")).read()
    # bench.py / __graft_entry__.py: every oracle import sits inside cpu_baseline() / smoke()
    for fname, func in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        src = open(os.path.join(ROOT, fname)).read()
        assert not re.search(r"^(from|import)\s+oracle\b", src, flags=re.M), fname      # none at module level
        body = src[src.index("def %s(" % func):]
        nxt = re.search(r"^def \w+\(", body[4:], flags=re.M)
        body = body[:nxt.start() + 4] if nxt else body
        n_inside = len(re.findall(r"^\s+(from|import)\s+oracle\b", body, flags=re.M))
        n_total = len(re.findall(r"^\s*(from|import)\s+oracle\b", src, flags=re.M))
        assert n_inside == n_total > 0, (fname, n_inside, n_total)


def test_bench_spawns_ranks_as_a_child_process(monkeypatch, capsys):
    """`python bench.py --gpus N` without WORLD_SIZE: the parent starts `python -m torch.distributed.run
    --nproc-per-node N ... bench.py <same arguments>` as a child (never exec), relays the one JSON line and the
    exit code, and does not import torch.cuda-touching code itself."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    seen = {}

    class Done(object):
        returncode = 0
        stdout = "noise\n" + '{"metric": "domain-steps/sec", "n_gpus": 4}' + "\n"

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        return Done()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    rc = bench.spawn_ranks(argparse.Namespace(gpus=4))
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "domain-steps/sec", "n_gpus": 4}' and "noise" in out.err
    Done.returncode, Done.stdout = 3, ""
    assert bench.spawn_ranks(argparse.Namespace(gpus=4)) == 3


def test_lpt_and_plan_sharding():
    from mamdr_amd import parallel
    sizes = [4493, 3495, 7864, 5242, 6291, 31462, 3145, 10485, 3932, 15728]
    for n in (1, 2, 4, 8):
        owner = parallel.lpt_partition(sizes, n)
        assert set(owner) <= set(range(n)) and len(owner) == 10
        loads = [sum(s for s, o in zip(sizes, owner) if o == r) for r in range(n)]
        assert max(loads) <= max(max(sizes), sum(sizes) / n * 4 / 3 + 1)       # LPT bound
    plan = {"seq": [3, 1, 2, 0], "dr": [(3, [1, 3]), (1, [0, 1]), (2, [3, 2]), (0, [2, 0])]}
    owner = [0, 1, 0, 1]
    s0, s1 = parallel.shard_plan(plan, owner, 0), parallel.shard_plan(plan, owner, 1)
    assert s0["seq"] == [2, 0] and s1["seq"] == [3, 1]                           # order preserved
    assert [q for q, _ in s0["dr"]] == [2, 0] and [q for q, _ in s1["dr"]] == [3, 1]
    assert sorted(s0["dr"] + s1["dr"]) == sorted(plan["dr"])                     # a partition


WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np, torch, torch.distributed as dist
from fake_engine import FakeEngine
from mamdr_amd import meta, parallel, plan as mplan, synthetic
from oracle import rng as orng
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = synthetic.generate({{"name": "Taobao", "split": "s", "n_domain": 4, "n_user": 300, "n_item": 200, "n_train": 1200,
                        "n_val": 400, "n_test": 400, "pretrained": True}}, batch_size=64, seed=5, emb_dim=8)
eng = FakeEngine(g["n_user"], g["n_item"], 4, 64, emb_dim=8, hidden=(16, 8, 4))
eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
for d in range(4):
    c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
sizes = [eng.n_rows(d, "train") for d in range(4)]
owner = parallel.lpt_partition(sizes, world)
theta0 = eng.oracle.get_flat().copy()
theta = torch.from_numpy(theta0.copy())
phis = {{d: torch.zeros(eng.n_params) for d in range(4) if owner[d] == rank}}
bufs = {{"delta": eng.new_vector(), "zero": eng.new_vector(), "merged": eng.new_vector()}}
plan = {{"seq": [2, 0, 3, 1], "dr": [(2, [0, 2]), (0, [1, 0]), (3, [2, 3]), (1, [3, 1])]}}
shuf = mplan.PassShuffler(sizes, 10000, 77 + rank, shuffle_fn=orng.shuffle_perm)
# reference point: what this rank's DN sub-sequence alone produces
local = parallel.shard_plan(plan, owner, rank)
eng.set_weights(theta)
probe = mplan.PassShuffler(sizes, 10000, 77 + rank, shuffle_fn=orng.shuffle_perm)
for d in local["seq"]:
    meta.run_pass(eng, d, probe, 64, 1e-3, [], "dn")
my_delta = eng.oracle.get_flat() - theta0
eng.optimizer_reset(); eng.oracle.step = 0
trace = parallel.mamdr_epoch_sharded(eng, meta, theta, phis, plan, owner, shuf, 64, 1e-3, 0.1, bufs)
# every rank must hold the same theta = theta0 + 0.1 * sum_g delta_g
deltas = [torch.zeros(eng.n_params) for _ in range(world)]
dist.all_gather(deltas, torch.from_numpy(my_delta.astype(np.float32)))
total = deltas[0].numpy().copy()
for t in deltas[1:]:
    total = (total + t.numpy()).astype(np.float32)
want = (theta0 + (total * np.float32(0.1)).astype(np.float32)).astype(np.float32)
assert np.array_equal(theta.numpy(), want), np.abs(theta.numpy() - want).max()
thetas = [torch.zeros(eng.n_params) for _ in range(world)]
dist.all_gather(thetas, theta)
assert all(torch.equal(thetas[0], t) for t in thetas)
# DR ran only for the owned query domains, and only those phis moved
assert sorted(set(t[1] for t in trace if t[0] == "dr_query")) == sorted(d for d in range(4) if owner[d] == rank)
assert all(float(p.abs().max()) > 0 for p in phis.values())
steps = torch.tensor([float(sum(t[2] for t in trace))])
dist.all_reduce(steps)
spd = [-(-n // 64) for n in sizes]
assert int(steps.item()) == mplan.plan_steps(plan, spd)
losses, aucs = parallel.gather_domain_scalars({{d: (0.5 + d, 0.6 + 0.01 * d) for d in phis}}, 4, torch.device("cpu"))
assert sorted(aucs) == [0, 1, 2, 3] and abs(aucs[3] - 0.63) < 1e-12
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_mamdr_epoch_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, here=HERE))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in out, out[-3000:]


BAL_SETUP = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np, torch
from fake_engine import FakeEngine
from mamdr_amd import meta, parallel, plan as mplan, synthetic
from oracle import rng as orng
D = 5
def make_engine():
    g = synthetic.generate({{"name": "Taobao", "split": "s", "n_domain": D, "n_user": 300, "n_item": 200, "n_train": 1900,
                            "n_val": 400, "n_test": 400, "pretrained": True}}, batch_size=64, seed=5, emb_dim=8)
    eng = FakeEngine(g["n_user"], g["n_item"], D, 64, emb_dim=8, hidden=(16, 8, 4))
    eng.bind_table("user_emb", g["tables"]["user_emb"]); eng.bind_table("item_emb", g["tables"]["item_emb"])
    for d in range(D):
        c = g["data"]["train"][d]; eng.bind_domain_data(d, "train", c["uid"], c["pid"], c["domain"], c["label"])
    return eng
def initial(eng):
    theta = torch.from_numpy(eng.oracle.get_flat().copy())
    rs = np.random.RandomState(3)
    phis = {{d: torch.from_numpy((rs.standard_normal(eng.n_params) * 0.01).astype(np.float32)) for d in range(D)}}
    return theta, phis
"""

BAL_WORKER = BAL_SETUP + r"""
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
eng = make_engine()
sizes = [eng.n_rows(d, "train") for d in range(D)]
spd = [-(-n // 64) for n in sizes]
theta, phis = initial(eng)
bal = parallel.BalancedMAMDR(eng, meta, theta, phis, spd, dn_mode={dn_mode!r})
planner = mplan.EpochPlanner(range(D), 2, True, True, 11)
shuf = mplan.PassShuffler(sizes, 10000, 77 + rank, shuffle_fn=orng.shuffle_perm)
total, owners = 0, []
for ep in range(3):
    p = planner.next_epoch()
    tr = bal.epoch(p, None, shuf, 64, 1e-3, 0.1)
    total += sum(t[2] for t in tr)
    owners.append(sorted(bal.mine))
    want = mplan.plan_steps(p, spd) + ((world - 1) * sum(spd[d] for d in p["seq"]) if {dn_mode!r} == "replicated" else 0)
    st = torch.tensor([float(sum(t[2] for t in tr))]); dist.all_reduce(st)
    assert int(st.item()) == want, (st.item(), want)
moved = sum(b > 4 * eng.n_params for b in bal.wire_bytes)       # epochs in which this rank SENT a phi slot
where = dict(bal.where)
bal.sync_phis()
np.savez({out!r} % rank, theta=theta.numpy(), owners=np.array([str(o) for o in owners]), moved=np.array(moved),
         where=np.array([where[d] for d in range(D)]), **{{"phi%d" % d: bal.phis[d].numpy() for d in range(D)}})
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("dn_mode", ["sharded", "replicated"])
def test_balanced_mamdr_epochs_gloo_world2(tmp_path, dn_mode):
    """BalancedMAMDR: per-epoch LPT of queries / DN passes, ONE all-reduce per epoch (the DN displacement), a phi
    whose owner changes travels point to point.  Two gloo ranks against an in-process emulation of the same two
    ranks (two engines, sums in numpy): theta and every phi identical on both ranks after sync_phis() and bit-equal
    to the emulation; ownership really moves between epochs.  dn_mode "replicated": every rank runs the whole DN
    chain and rank 0's result is everybody's -- theta follows the reference's sequential DN update exactly."""
    script = tmp_path / "bal_worker.py"
    script.write_text(BAL_WORKER.format(root=ROOT, here=HERE, out=str(tmp_path / "bal_%d.npz"), dn_mode=dn_mode))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and ("rank %d ok" % r) in out.decode(), out.decode()[-3000:]
    res = [np.load(str(tmp_path / ("bal_%d.npz" % r))) for r in range(2)]
    for k in res[0].files:
        if k not in ("owners", "moved"):
            assert np.array_equal(res[0][k], res[1][k]), k
    assert len(set(res[0]["owners"])) > 1 or len(set(res[1]["owners"])) > 1      # the assignment moved
    assert int(res[0]["moved"]) + int(res[1]["moved"]) > 0                      # ... and slots travelled point to point
    assert set(res[0]["where"]) <= {0, 1}
    # ---- emulation of the two ranks in this process
    ns = {}
    exec(BAL_SETUP.format(root=ROOT, here=HERE), ns)
    import torch
    from mamdr_amd import meta, parallel, plan as mplan
    from oracle import rng as orng
    D = ns["D"]
    engs = [ns["make_engine"](), ns["make_engine"]()]
    sizes = [engs[0].n_rows(d, "train") for d in range(D)]
    spd = [-(-n // 64) for n in sizes]
    theta, phi0 = ns["initial"](engs[0])
    theta = theta.clone()
    phis = {d: v.clone() for d, v in phi0.items()}
    planner = mplan.EpochPlanner(range(D), 2, True, True, 11)
    shufs = [mplan.PassShuffler(sizes, 10000, 77 + r, shuffle_fn=orng.shuffle_perm) for r in range(2)]
    for ep in range(3):
        p = planner.next_epoch()
        if dn_mode == "replicated":
            dr_owner, _, _ = parallel.epoch_assignment({"seq": [], "dr": p["dr"]}, spd, 2)
        else:
            dr_owner, dn_owner, _ = parallel.epoch_assignment(p, spd, 2)
        deltas = []
        for r in range(2):
            engs[r].set_weights(theta)
            for d in p["seq"]:
                if dn_mode == "replicated" or dn_owner[d] == r:
                    meta.run_pass(engs[r], d, shufs[r], 64, 1e-3, [], "dn")
            deltas.append((engs[r].weights.numpy() - theta.numpy()).astype(np.float32))
        # replicated: rank 0's chain is everybody's (== the single-process DN update on rank 0's state)
        tot = deltas[0] if dn_mode == "replicated" else (deltas[0] + deltas[1]).astype(np.float32)
        engs[0].interp(theta, torch.from_numpy(tot), torch.zeros_like(theta), 0.1)
        for r in range(2):
            merged = torch.empty_like(theta)
            for q, sup in p["dr"]:
                if dr_owner[q] == r:
                    meta.dr_query(engs[r], theta, phis[q], q, sup, shufs[r], 64, 1e-3, 0.1, [], merged)
    assert np.array_equal(theta.numpy(), res[0]["theta"])
    for d in range(D):
        assert np.array_equal(phis[d].numpy(), res[0]["phi%d" % d]), d


def test_epoch_assignment_balances_the_sampled_plan():
    from mamdr_amd import parallel, plan as mplan
    spd = [5, 4, 8, 6, 7, 31, 4, 11, 4, 16, 3, 9, 2, 14, 6, 5, 7, 3, 20, 4, 6, 8, 5, 3, 9, 4, 12, 5, 3, 6]
    planner = mplan.EpochPlanner(range(30), 5, True, True, 123)
    for _ in range(5):
        p = planner.next_epoch()
        for n in (1, 2, 4, 8):
            dr_owner, dn_owner, load = parallel.epoch_assignment(p, spd, n)
            assert sorted(dr_owner) == list(range(30)) and sorted(dn_owner) == list(range(30))
            assert abs(sum(load) - mplan.plan_steps(p, spd)) < 1e-6          # the cost model IS the step count
            assert max(load) <= sum(load) / n * 1.12 + 1                     # 30 long-tailed domains: within 12 %
    # with a cap on the query pass (domain_regulation_step) the cost follows it
    p = planner.next_epoch()
    _, _, load = parallel.epoch_assignment(p, spd, 4, domain_regulation_step=2)
    assert abs(sum(load) - mplan.plan_steps(p, spd, 2)) < 1e-6


RUN_WORKER = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np
from fake_engine import FakeEngine
from mamdr_amd import cli, synthetic
real = synthetic.generate
synthetic.generate = lambda *a, **k: real(*a, **dict(k, emb_dim=8))          # tiny tables for CPU speed
cfg = json.load(open({cfg!r}))
built = []
res = cli.main(cfg, FakeEngine, on_model=built.append)   # run.py's entry; init_distributed() reads RANK / WORLD_SIZE
rank = int(os.environ.get("RANK", "0"))
import hashlib
sha = hashlib.sha1(built[0].model.weights.numpy().tobytes()).hexdigest()      # the LIVE model this rank ends with
json.dump({{"avg_loss": res[0], "avg_auc": res[1], "domain_auc": {{str(k): v for k, v in res[3].items()}},
           "weights_sha": sha}}, open({out!r} % rank, "w"))
print("rank", rank, "ok")
'''


def test_run_entry_sharded_mamdr_gloo_world2(tmp_path):
    """run.py's entry under 2 processes (gloo): query domains sharded by owner, DN all-reduce, owners evaluate
    and finetune, every rank ends with the same per-domain results; they stay close to the 1-process run."""
    _run_entry_worlds(tmp_path, "mlp_meta_mamdr_finetune")


def test_run_entry_sharded_domain_negotiation_gloo_world2(tmp_path):
    """the same for the Domain Negotiation wrapper (BASELINE config 3's wrapper): sub-sequences + one all-reduce,
    evaluation and finetune dealt round-robin over ranks that hold identical weights."""
    _run_entry_worlds(tmp_path, "mlp_meta_domain_negotiation_finetune")


def test_run_entry_sharded_reptile_batch_gloo_world2(tmp_path):
    """batch Reptile: the epoch's sum of per-domain displacements is a sum over ranks -- domains
    dealt by owner, ONE all-reduce of the accumulator per epoch, identical theta on every rank."""
    _run_entry_worlds(tmp_path, "mlp_meta_reptile_batch")


@pytest.mark.parametrize("name,extra", [
    ("mlp_meta_mamdr_batch", {}),                                   # mamdr.py:100-108 batch names: accumulated DR update
    ("mlp_meta_mamdr", {"finetune_every_epoch": True}),             # mamdr.py:110-143
    ("mlp_meta_domain_negotiation", {"meta_train_step": 2, "target_domain": 1}),   # domain_negotiation.py:44-45,67,89-93
    ("mlp_meta_reptile", {}),                                       # reptile.py:45-99, per-domain interpolation
    ("mlp_meta_reptile", {"target_domain": 1}),                     # reptile.py:47-48,82-85,98-102: target step + closing pass
    ("mlp_meta_reptile_batch", {"target_domain": 2}),
    ("mlp_meta_mamdr_finetune", {"dn_mode": "replicated"}),         # SURVEY 8e fallback: one DN chain, DR sharded
    ("mlp_meta_domain_negotiation", {"meta_finetune_step": 1}),     # maml.py:245-287: fine-tune, then validate, per domain
    ("mlp_meta_mamdr", {"meta_finetune_step": 1}),
])
def test_run_entry_sharded_variants_gloo_world2(tmp_path, name, extra):
    """the remaining multi-process variants through run.py's entry, 2 gloo ranks: identical results on every rank,
    every domain reported, close to the 1-process run."""
    _run_entry_worlds(tmp_path, name, extra)


def _run_entry_worlds(tmp_path, name, extra=None):
    import json
    sys.path.insert(0, HERE)
    from test_host_logic import tiny_config
    cfg = tiny_config(tmp_path, name, epochs=2)
    cfg["train"].update(extra or {})
    cfg_path = tmp_path / "cfg.json"
    cfg_path.write_text(json.dumps(cfg))
    script = tmp_path / "run_worker.py"
    script.write_text(RUN_WORKER.format(root=ROOT, here=HERE, cfg=str(cfg_path), out=str(tmp_path / "res_%d.json")))
    results = {}
    for world in (1, 2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE=str(world), MAMDR_SHARE_GPU="1",
                   OMP_NUM_THREADS="2")
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        for r, p in enumerate(procs):
            try:
                out, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0 and ("rank %d ok" % r) in out.decode(), out.decode()[-3000:]
        results[world] = [json.load(open(str(tmp_path / ("res_%d.json" % r)))) for r in range(world)]
    a, b = results[2]
    sha = [r.pop("weights_sha") for r in (a, b)]
    if (extra or {}).get("target_domain", -1) >= 0 and "mamdr" not in name:
        # the closing pass over the target domain ran on every rank with its own shuffle stream and Adam slots: the ranks
        # must still END with one live model (parallel.broadcast_live) -- validation deals the domains
        # round-robin on that premise and rank 0 alone writes the checkpoint
        assert sha[0] == sha[1], sha
    assert a == b and sorted(a["domain_auc"]) == ["0", "1", "2"]          # every rank holds every domain's result
    assert np.isfinite(a["avg_loss"]) and abs(a["avg_auc"] - results[1][0]["avg_auc"]) < 0.1
    # the same two ranks as LANES of this process (train.lanes = 2, parallel.LaneGroup): the 2-process run bit for bit --
    # the returned results and the live model every lane ends with
    import hashlib
    import shutil
    from fake_engine import FakeEngine
    from mamdr_amd import cli, synthetic
    for d in ("result", "checkpoint"):
        shutil.rmtree(str(tmp_path / d), ignore_errors=True)
    real = synthetic.generate
    synthetic.generate = lambda *a_, **k: real(*a_, **dict(k, emb_dim=8))
    try:
        built = []
        lane_cfg = json.loads(json.dumps(cfg))
        lane_cfg["train"]["lanes"] = 2
        res = cli.main(lane_cfg, FakeEngine, on_model=built.append)
    finally:
        synthetic.generate = real
    assert len(built) == 2
    got = {"avg_loss": res[0], "avg_auc": res[1], "domain_auc": {str(k): v for k, v in res[3].items()}}
    assert got == a, (got, a)
    lane_sha = sorted(hashlib.sha1(m.model.weights.numpy().tobytes()).hexdigest() for m in built)
    assert lane_sha == sorted(sha), (lane_sha, sha)


COMPOSED_WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np
from fake_engine import FakeEngine
from mamdr_amd import cli, synthetic
real = synthetic.generate
synthetic.generate = lambda *a, **k: real(*a, **dict(k, emb_dim=8))
cfg = json.load(open({cfg!r}))
built = []
res = cli.main(cfg, FakeEngine, on_model=built.append)
rank = int(os.environ.get("RANK", "0"))
json.dump({{"avg_loss": res[0], "avg_auc": res[1], "domain_auc": {{str(k): v for k, v in res[3].items()}},
           "weights_sha": sorted(hashlib.sha1(m.model.weights.numpy().tobytes()).hexdigest() for m in built),
           "traces": sorted([list(map(list, getattr(m, "trace", []))) for m in built])}}, open({out!r} % rank, "w"))
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("name,extra", [
    ("mlp_meta_mamdr_finetune", {}),
    ("mlp_meta_mamdr_finetune", {"dn_mode": "replicated"}),
    ("mlp_meta_domain_negotiation_finetune", {}),
    ("mlp_meta_reptile_batch", {"target_domain": 2}),
    ("mlp_meta_mamdr_batch", {}),
])
def test_run_entry_ranks_x_lanes_gloo_world2(tmp_path, name, extra):
    """RANKS x LANES: run.py's entry under 2 gloo processes with train.lanes = 2 -- ONE world of 4
    participants (rank * 2 + lane): a rank's DR queries and DN sub-sequence are dealt on to its lanes, a collective is the
    lanes' step followed by one inter-rank collective per process, a phi slot that changes hands between processes travels
    through lane 0's batched send / recv.  It must equal, bit for bit -- returned results, every participant's live model and
    trace --, the ONE-process run of 4 lanes whose all-reduce adds up in the same order (lanes 0 + 1, lanes 2 + 3, then the
    two sums: train.lane_sum_block = 2).  (A flat world of 4 gloo ranks adds in gloo's ring order, chunk by chunk: no
    hierarchical sum can equal that bit for bit; the flat 2-lane run equals the 2-process run, test above.)"""
    import json
    sys.path.insert(0, HERE)
    from test_host_logic import tiny_config
    cfg = tiny_config(tmp_path, name, epochs=2)
    cfg["dataset"]["synthetic"].update(n_domain=5, n_train=1500, n_val=500, n_test=500)
    cfg["train"].update(extra or {}, lanes=2)
    cfg_path = tmp_path / "cfg.json"
    cfg_path.write_text(json.dumps(cfg))
    script = tmp_path / "composed_worker.py"
    script.write_text(COMPOSED_WORKER.format(root=ROOT, here=HERE, cfg=str(cfg_path), out=str(tmp_path / "res_%d.json")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", MAMDR_SHARE_GPU="1", OMP_NUM_THREADS="2")
    env.pop("MAMDR_LANES", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and ("rank %d ok" % r) in out.decode(), out.decode()[-3000:]
    a, b = [json.load(open(str(tmp_path / ("res_%d.json" % r)))) for r in range(2)]
    sha, traces = a.pop("weights_sha") + b.pop("weights_sha"), a.pop("traces") + b.pop("traces")
    assert a == b and sorted(a["domain_auc"]) == ["0", "1", "2", "3", "4"]
    assert sum(1 for t in traces if t) >= 3                  # the work really was dealt over the four participants
    # the reference: 4 lanes of ONE process, sums in blocks of 2
    import hashlib
    import shutil
    from fake_engine import FakeEngine
    from mamdr_amd import cli, synthetic
    for d in ("result", "checkpoint"):
        shutil.rmtree(str(tmp_path / d), ignore_errors=True)
    real = synthetic.generate
    synthetic.generate = lambda *a_, **k: real(*a_, **dict(k, emb_dim=8))
    try:
        built = []
        lane_cfg = json.loads(json.dumps(cfg))
        lane_cfg["train"].update(lanes=4, lane_sum_block=2)
        res = cli.main(lane_cfg, FakeEngine, on_model=built.append)
    finally:
        synthetic.generate = real
    assert len(built) == 4
    got = {"avg_loss": res[0], "avg_auc": res[1], "domain_auc": {str(k): v for k, v in res[3].items()}}
    assert got == a, (got, a)
    assert sorted(hashlib.sha1(m.model.weights.numpy().tobytes()).hexdigest() for m in built) == sorted(sha)
    assert sorted([list(map(list, getattr(m, "trace", []))) for m in built]) == sorted(traces)


TAIL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from mamdr_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
D, X, NM, NT = 3, 4, 10, 6
class Eng(object):                      # what TailSync touches of a TowerEngine
    n_domain, n_meta, n_params = D, NM, NM + NT
    def __init__(self):
        self.weights = torch.arange(NM + NT, dtype=torch.float32).clone()
        self.aux = torch.zeros(4 * D * X + D)
        self.aux[D * X:2 * D * X] = 1.0
eng = Eng()
ts = parallel.TailSync(eng)
assert ts.active and ts.floats() == NT + 2 * D * X + D
# each rank trains: the tail moves by (rank + 1) * 0.5 everywhere; domain `rank` takes (rank + 2) steps with statistics
# mean = 10 * (rank + 1), var = 2 + rank; domain 2 is touched by nobody
eng.weights[NM:] += (rank + 1) * 0.5
mm, mv, bm, bv, steps = ts._aux_views(eng.aux)
mm[rank] = 10.0 * (rank + 1); mv[rank] = 2.0 + rank; steps[rank] = rank + 2
ts.sync()
# step-weighted mean of the displacements (this stand-in has no per-domain tensors: every element is shared, weights = each
# rank's total steps 2 and 3)
want_tail = torch.arange(NM, NM + NT, dtype=torch.float32) + (2 * 0.5 + 3 * 1.0) / 5
assert torch.allclose(eng.weights[NM:], want_tail), eng.weights
assert torch.equal(eng.weights[:NM], torch.arange(NM, dtype=torch.float32))          # theta's part is not the tail's business
for r in range(2):
    assert torch.allclose(mm[r], torch.full((X,), 10.0 * (r + 1))) and torch.allclose(mv[r], torch.full((X,), 2.0 + r))
    assert float(steps[r]) == r + 2
    assert torch.allclose(bm[r], mm[r] * (1 - 0.99 ** (r + 2)))
assert torch.equal(mm[2], torch.zeros(X)) and torch.equal(mv[2], torch.ones(X)) and float(steps[2]) == 0
# second round: BOTH ranks train domain 0 (1 and 3 more steps): step-weighted average of their statistics
eng.weights[NM:] -= 0.25
mm[0] = 4.0 if rank == 0 else 8.0
steps[0] += 1.0 if rank == 0 else 3.0
ts.sync()
assert torch.allclose(eng.weights[NM:], want_tail - 0.25)       # both moved by -0.25 (in 1 and 3 steps): the mean is -0.25
assert torch.allclose(mm[0], torch.full((X,), (1 * 4.0 + 3 * 8.0) / 4.0)) and float(steps[0]) == 2 + 4
# per-domain tensors: a slice only ONE rank trained keeps that rank's displacement, a slice both trained takes the mean
# weighted by the steps on that domain, a slice nobody trained the mean weighted by the total steps; shared tensors as above
class Eng2(Eng):
    segments = {{"theta": (0, NM), "Wd0": (NM, 3), "wo": (NM + 3, 3)}}      # Wd0 = [D = 3][1], wo shared
eng2 = Eng2()
ts2 = parallel.TailSync(eng2)
assert ts2.elem_dom.tolist() == [0, 1, 2, -1, -1, -1]
_, _, _, _, st2 = ts2._aux_views(eng2.aux)
# rank 0: 4 steps on domain 0, 1 step on domain 1;  rank 1: 3 steps on domain 1.  Everybody moves everything by (rank + 1)
eng2.weights[NM:] += float(rank + 1)
if rank == 0:
    st2[0] = 4.0; st2[1] = 1.0
else:
    st2[1] = 3.0
ts2.sync()
e = parallel.TailSync.EPS
got = (eng2.weights[NM:] - torch.arange(NM, NM + NT, dtype=torch.float32)).tolist()
want = [(1 * (4 + e * 5) + 2 * (0 + e * 3)) / (4 + e * 8),         # domain 0: rank 0 alone trained it
        (1 * (1 + e * 5) + 2 * (3 + e * 3)) / (4 + e * 8),         # domain 1: 1 and 3 steps
        (1 * 5 + 2 * 3) / 8.0,                                     # domain 2: nobody -> total steps decide
        (1 * 5 + 2 * 3) / 8.0, (1 * 5 + 2 * 3) / 8.0, (1 * 5 + 2 * 3) / 8.0]
assert np.allclose(got, want, rtol=1e-6), (got, want)
assert abs(got[0] - 1.0) < 2e-3 and abs(got[1] - 1.75) < 2e-3
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_tail_sync_gloo_world2(tmp_path):
    """parallel.TailSync (the tensors outside theta / phi of the Star tower under several ranks): the tail becomes
    common + the step-weighted mean of the ranks' displacements, a domain's moving statistics the step-weighted average of the ranks
    that trained it, the zero-debias slots follow the summed step count, untouched domains keep their values."""
    script = tmp_path / "tail_worker.py"
    script.write_text(TAIL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and ("rank %d ok" % r) in out.decode(), out.decode()[-3000:]


PREFLIGHT_WORKER = r"""
import os, sys, warnings
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from mamdr_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rec = parallel.preflight(torch.device("cpu"))
assert rec["ranks"] == world and rec["all_reduce"] is True and rec["p2p"] is True and rec["broadcast"] is True, rec
assert parallel.P2P_ENABLED
# a ring that RAISES on one rank: every rank switches to the broadcast path together
real = dist.batch_isend_irecv
def broken(ops):
    if rank == 1:
        raise RuntimeError("injected: no peer access")
    return real(ops)
dist.batch_isend_irecv = broken
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    if rank == 1:
        rec = parallel.preflight(torch.device("cpu"))
        assert any("broadcast" in str(x.message) for x in w)
    else:
        # (the healthy ranks' sends complete into gloo's buffers; their receive from the broken rank never arrives, so
        # they take part through the agreement only -- the same code path a rank whose own ring worked goes through)
        flag = torch.tensor([1.0]); x = torch.full((4096,), float(rank + 1)); dist.all_reduce(x)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parallel.P2P_ENABLED = bool(flag.item() > 0.5)
        y = torch.full((4096,), 7.0 if rank == world - 1 else -1.0); dist.broadcast(y, src=world - 1)
assert parallel.P2P_ENABLED is False
dist.batch_isend_irecv = real
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _spawn(tmp_path, text, world, port, timeout=300, threads="1"):
    script = tmp_path / "worker.py"
    script.write_text(text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS=threads)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in out, out[-3000:]
    return outs


def test_preflight_gloo_world3(tmp_path):
    """parallel.preflight (first contact with the communicator): values of the all-reduce, the
    send / recv ring and the broadcast checked on three gloo ranks; a ring that raises on one rank switches EVERY rank
    to the per-slot broadcast hand-over (P2P_ENABLED agreed by an all-reduce)."""
    _spawn(tmp_path, PREFLIGHT_WORKER.format(root=ROOT), 3, 29541)


WORLD8_WORKER = BAL_SETUP.replace("D = 5", "D = 30").replace('"n_train": 1900', '"n_train": 9000').replace(
    '"n_val": 400, "n_test": 400', '"n_val": 1500, "n_test": 1500') + r"""
import json
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pre = parallel.preflight(torch.device("cpu"))
eng = make_engine()
sizes = [eng.n_rows(d, "train") for d in range(D)]
spd = [-(-n // 64) for n in sizes]
theta, phis = initial(eng)
bal = parallel.BalancedMAMDR(eng, meta, theta, phis, spd)
planner = mplan.EpochPlanner(range(D), 5, True, True, 11)
shuf = mplan.PassShuffler(sizes, 10000, 77 + rank, shuffle_fn=orng.shuffle_perm)
loads = []
for ep in range(2):
    p = planner.next_epoch()
    tr = bal.epoch(p, None, shuf, 64, 1e-3, 0.1)
    loads.append(bal.last_load)
    st = torch.tensor([float(sum(t[2] for t in tr))]); dist.all_reduce(st)
    assert int(st.item()) == mplan.plan_steps(p, spd)
    mine = sum(t[2] for t in tr)
    assert mine == loads[-1][rank], (mine, loads[-1])
P = eng.n_params
# wire accounting of this rank: the DN all-reduce (P floats) + 4 bytes x P per phi slot it SENT
for wb in bal.wire_bytes:
    assert wb >= 4 * P and (wb - 4 * P) % (4 * P) == 0, (wb, P)
bal.sync_phis()
if rank == 0:
    print("LINE", json.dumps({{"preflight": pre, "bound": [sum(l) / max(l) for l in loads], "wire": bal.wire_bytes,
                               "host_prep_s": bal.host_prep_s}}))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_balanced_mamdr_gloo_world8(tmp_path):
    """the 8-rank shape of SURVEY 8e on CPU (gloo, oracle-backed engines, 30 miniature domains, sample_num 5): the
    preflight passes, every rank runs exactly the steps the per-epoch LPT planned for it, the global step count is the
    plan's, wire bytes are the DN all-reduce + whole phi slots, and the partition leaves the headroom north_star's 6x
    needs (sum of the ranks' loads / the largest >= 6.5)."""
    import json
    outs = _spawn(tmp_path, WORLD8_WORKER.format(root=ROOT, here=HERE), 8, 29547, timeout=600)
    line = [ln for ln in outs[0].splitlines() if ln.startswith("LINE ")][0]
    rec = json.loads(line[5:])
    assert rec["preflight"]["p2p"] is True and rec["preflight"]["ranks"] == 8
    assert min(rec["bound"]) >= 6.5, rec["bound"]


def test_taobao30_partition_bound_at_8_ranks():
    """BASELINE.json configs[3] (Taobao-30, bs 4096, sample_num 5 + the query domain) dealt over 8 ranks by
    parallel.epoch_assignment on the workload's own per-domain step counts: the partition allows >= 6.5x on every one
    of ten sampled epochs (north_star asks for >= 6x at 8 GPUs), and no rank is left without work."""
    from mamdr_amd import parallel, plan as mplan, synthetic
    shape = synthetic.SHAPES["taobao30"]
    sizes = synthetic._domain_sizes(shape["n_train"], shape["n_domain"], 4096, np.random.RandomState(123 + 1))
    spd = [int(-(-int(n) // 4096)) for n in sizes]
    planner = mplan.EpochPlanner(range(30), 5, True, True, seed=123)
    for ep in range(10):
        plan = planner.next_epoch()
        dr_owner, dn_owner, load = parallel.epoch_assignment(plan, spd, 8)
        assert sum(load) == mplan.plan_steps(plan, spd) and min(load) > 0
        assert sum(load) / max(load) >= 6.5, (ep, load)


# ------------------------------------------------------------------ lanes (parallel.LaneGroup): ranks as threads of one process
def test_lane_group_collectives_cpu():
    """world() / all_reduce / broadcast / barrier / the phi hand-over inside a LaneGroup: the values every rank of a process
    group would see, sums taken in rank order on every lane (bit-identical everywhere)."""
    import torch
    from mamdr_amd import parallel
    L = 3
    rs = np.random.RandomState(3)
    base = [torch.from_numpy(rs.standard_normal(1000).astype(np.float32)) for _ in range(L)]

    def fn(lane):
        assert parallel.world() == (lane, L) and parallel.lanes().n == L
        out = {}
        t = base[lane].clone()
        parallel.all_reduce(t)
        out["sum"] = t
        m = base[lane].clone()
        parallel.all_reduce(m, "max")
        out["max"] = m
        b = base[lane].clone()
        parallel.broadcast(b, 1)
        out["bcast"] = b
        vecs = {d: torch.full((4,), float(10 * lane + d)) for d in range(3)}
        parallel.lanes().transfer(lane, vecs, [(0, 0, 2), (1, 2, 1), (2, 1, 1)])
        out["vecs"] = vecs
        parallel.barrier()
        s = torch.tensor([float(lane)], dtype=torch.float64)
        parallel.all_reduce(s)
        out["scalar"] = float(s[0])
        return out
    outs = parallel.LaneGroup(L).run(fn)
    want = (base[0] + base[1]) + base[2]
    for lane, o in enumerate(outs):
        assert torch.equal(o["sum"], want) and torch.equal(o["bcast"], base[1])
        assert torch.equal(o["max"], torch.maximum(torch.maximum(base[0], base[1]), base[2]))
        assert o["scalar"] == 3.0
    assert float(outs[2]["vecs"][0][0]) == 0.0 and float(outs[1]["vecs"][1][0]) == 21.0 and float(outs[1]["vecs"][2][0]) == 12.0
    assert float(outs[0]["vecs"][0][0]) == 0.0 and float(outs[0]["vecs"][1][0]) == 1.0       # the senders keep theirs
    assert parallel.world() == (0, 1) and parallel.lanes() is None                        # nothing leaks to the caller's thread


def test_lane_group_error_ends_every_lane():
    """a lane that raises breaks the barrier: the others do not wait for it for ever, and the caller sees ITS exception."""
    import torch
    from mamdr_amd import parallel

    def fn(lane):
        if lane == 1:
            raise KeyError("lane 1 failed")
        t = torch.zeros(4)
        for _ in range(3):
            parallel.all_reduce(t)
        return lane
    with pytest.raises(KeyError, match="lane 1 failed"):
        parallel.LaneGroup(3).run(fn)


def test_run_entry_lanes_need_a_sharded_wrapper(tmp_path, monkeypatch):
    sys.path.insert(0, HERE)
    from fake_engine import FakeEngine
    from test_host_logic import patch_emb_dim, tiny_config
    from mamdr_amd import cli
    patch_emb_dim(monkeypatch)
    cfg = tiny_config(tmp_path, "mlp", epochs=1)
    cfg["train"]["lanes"] = 2
    with pytest.raises(NotImplementedError, match="multi-lane"):
        cli.main(cfg, FakeEngine)
    # MAMDR_LANES overrides the config's key; three lanes over three domains
    cfg = tiny_config(tmp_path, "mlp_meta_mamdr", epochs=1)
    monkeypatch.setenv("MAMDR_LANES", "3")
    built = []
    res = cli.main(cfg, FakeEngine, on_model=built.append)
    assert len(built) == 3 and sorted(res[3]) == [0, 1, 2] and np.isfinite(res[0])
