#!/bin/bash
# Star-step iteration loop (through gpurun): tools/r03_star.sh <tag>
#   Star parity tests, the Amazon-13 bench line (full rows, 1 + 1 epochs), a rocprofv3 kernel trace at 25 % of the rows
TAG=${1:-r03star}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python -m pytest tests/test_gpu_parity.py -q -k "star" 2>&1 | tail -5
python bench.py --workload amazon13 --steps 1 --warmup 1 --cpu-budget 0 --no-targets --no-profile > "$OUT/bench_amazon13.json" 2> "$OUT/bench_amazon13.err"
python - "$OUT/bench_amazon13.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("amazon13 %.0f steps/s %.2f us/step" % (d["value"], d["us_per_domain_step"]))
PY
cd /tmp
MAMDR_BENCH_ROW_SCALE=0.25 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o run -- python3 "$REPO/bench.py" --workload amazon13 --steps 1 --warmup 1 --cpu-budget 0 --no-targets --no-profile > "$OUT/prof.log" 2>&1
cd "$REPO"
python tools/rocpd_summary.py stats "$(find "$OUT/prof" -name '*.db' | head -1)" "$OUT/kernel_stats_amazon13_q.csv"
rm -rf "$OUT/prof"
python - "$OUT/kernel_stats_amazon13_q.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nst = sum(int(r["Calls"]) for r in rows if "k_tower" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("steps", nst, "GPU busy us/step %.2f" % (tot / nst / 1e3))
for r in rows[:14]:
    print("   %-52s calls %6s avg %8.2f us  per-step %7.2f us" % (r["Name"].replace("mamdr::", "").replace("void ", "")[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / nst / 1e3))
PY
