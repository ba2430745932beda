"""csrc/launch_combiner.h -- the host-side rendezvous behind mamdr_group_* (the step launches of several contexts in one
launch) -- compiled with g++ (no HIP) and hammered from threads: every submitted descriptor is delivered exactly once, in
its member's order, grouped by kind in ascending member order; members that enter and leave calls of ragged lengths never
deadlock (tests/native/combiner_test.cpp has a watchdog)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("members", [1, 2, 4, 7])
def test_launch_combiner_delivers_everything_once_and_never_hangs(tmp_path, members):
    exe = str(tmp_path / "combiner_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "native", "combiner_test.cpp"), "-o", exe])
    out = subprocess.run([exe, str(members), "120"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    fields = out.stdout.split()
    rec = dict(zip(fields[0::2], fields[1::2]))
    assert rec["bad"] == "0" and rec["submitted"] == rec["delivered"] == rec["carried"]
    if members > 1:
        assert int(rec["batched"]) > 0          # launches really were shared
