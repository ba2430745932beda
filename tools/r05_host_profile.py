"""round 5: where does the HOST spend an epoch of the headline workload?  (Taobao-10 has ~250 passes of ~5 steps per epoch: the
Python around every pass is on the critical path once it approaches the 110 us the GPU needs for them.)
  python tools/r05_host_profile.py [epochs=20]     -> cProfile, top functions by own time and by cumulative time"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench        # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import torch        # noqa: E402
torch.cuda.set_device(0)
pr = cProfile.Profile()
pr.enable()
rec = bench.run_workload("taobao10", epochs, 3, 0, 1, False, 0.0)
pr.disable()
print("value %.0f domain-steps/s, %.2f ms per epoch (under cProfile)" % (rec["value"], rec["ms_per_step"]))
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
