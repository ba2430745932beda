"""Build libmamdr_hip.so (gfx950) in-tree with hipcc.

Used by __graft_entry__.build().  The shared library lands next to this file
(mamdr_amd/libmamdr_hip.so): git-ignored, but it travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmamdr_hip.so")
BUILD = os.path.join(HERE, "build")

# (source, extra flags).  outer_kernels must not fuse multiply-adds (bit-exact vs numpy).
SOURCES = [
    # (kernarg preload: the leading scalar arguments of k_tower / k_tower4 arrive in SGPRs with the wave)
    ("step_kernels.hip", ["-mllvm", "-amdgpu-kernarg-preload-count=14"]),
    # emb_kernels: the lazy and the dense table updates must round identically -> no implicit fma fusion
    # (HIP's __fmul_rn / __fadd_rn are plain operators; explicit __fmaf_rn where an fma is wanted)
    ("emb_kernels.hip", ["-ffp-contract=off"]),
    ("tower4_kernels.hip", ["-mllvm", "-amdgpu-kernarg-preload-count=14"]),
    ("fused_kernels.hip", []),
    ("star_kernels.hip", []),
    ("outer_kernels.hip", ["-ffp-contract=off"]),
    ("graph_engine.hip", []),
    ("mamdr_api.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mamdr_hip.h"))
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
        # a change of flags rebuilds too: the command line is kept next to the object
        stamp = o + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or _stale(o, [s] + headers):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
