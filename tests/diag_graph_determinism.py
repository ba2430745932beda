"""Diagnostic (GPU box): the HIP side of an end-to-end case is run-to-run deterministic and does not depend on what ran in the
process before it -- python tests/diag_graph_determinism.py <case> [prewarm]; prints a hash of every evaluation of the run
(round 6: the generic-layer engine's MAMDR_GRAPH_TILE32_BELOW leak was found with it)."""
import sys, os, hashlib, json, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import oracle_jobs
import test_gpu_e2e as T
case = sys.argv[1] if len(sys.argv) > 1 else "taobao10_shared_bottom_as_configured"
prewarm = len(sys.argv) > 2 and sys.argv[2] == "prewarm"
if prewarm:
    # another context first: leaves its garbage in freed device memory
    x = torch.randn(64 << 20, device="cuda"); del x
    kw0 = T._job_kwargs("taobao10_mmoe_as_configured")
    tmp0 = tempfile.mkdtemp()
    cfg0 = oracle_jobs.pipeline_config(kw0["cfg_file"], kw0["model_name"], tmp0, dict(kw0["train"], epoch=1), dict(kw0["dataset"]), dict(kw0.get("model", ())))
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        oracle_jobs.run_pipeline(cfg0)
kw = T._job_kwargs(case)
tmp = tempfile.mkdtemp()
cfg = oracle_jobs.pipeline_config(kw["cfg_file"], kw["model_name"], tmp, dict(kw["train"], epoch=2), dict(kw["dataset"]), dict(kw.get("model", ())))
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    s = oracle_jobs.run_pipeline(cfg)
vals = [e for e in s["events"] if e[0] == "eval"]
h = hashlib.sha1(json.dumps([[e[1], e[4], e[5]] for e in vals], sort_keys=True).encode()).hexdigest()
print(case, "prewarm" if prewarm else "fresh", h, [round(v, 6) for v in list(vals[-1][5].values())[:4]])
