"""The embedding gather on Amazon-6-sized tables (316 MB: beyond the 256 MiB infinity cache), as its own program
for the rocprofv3 passes whose summaries feed bench.py's `gather.traffic`:

  cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/gather_trace -o run -- python3 $REPO/tools/gather_hbm.py
  cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/gather_fetch -o run -- python3 $REPO/tools/gather_hbm.py
  cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/gather_write -o run -- python3 $REPO/tools/gather_hbm.py
  python tools/rocpd_summary.py pmc1 <fetch.db> <write.db> k_gather k_gather@amazon6 profiles/pmc_hbm_latest.json

Same sizes and the same launches as bench.py's gather_hbm_record (it IS that function)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
rec = bench.gather_hbm_record(torch.device("cuda", 0))
print(json.dumps(rec))
