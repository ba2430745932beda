#!/bin/bash
# round-2 measurement pass (through gpurun): tools/r02_measure.sh <tag> [quick]
TAG=${1:-r02a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python bench.py > "$OUT/bench_taobao10.json" 2> "$OUT/bench_taobao10.err"
tail -c 600 "$OUT/bench_taobao10.err"
MAMDR_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-budget 0 > "$OUT/bench_taobao10_2ranks_shared.json" 2> "$OUT/bench_2ranks.err"
tail -c 400 "$OUT/bench_2ranks.err"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/gather_trace" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/gather_fetch" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/gather_write" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_write.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" | head
F=$(find "$OUT/gather_fetch" -name "*.db" | head -1); W=$(find "$OUT/gather_write" -name "*.db" | head -1); T=$(find "$OUT/gather_trace" -name "*.db" | head -1)
cp profiles/pmc_hbm_latest.json "$OUT/pmc_hbm_latest.json"
python tools/rocpd_summary.py pmc1 "$F" "$W" k_gather k_gather@amazon6 "$OUT/pmc_hbm_latest.json"
python tools/rocpd_summary.py stats "$T" "$OUT/kernel_stats_gather_amazon6.csv"
cat "$OUT/kernel_stats_gather_amazon6.csv" | head -5
python -c "
import json; d=json.load(open('$OUT/pmc_hbm_latest.json')); print(d.get('k_gather@amazon6'))"
# keep the merged databases small: drop the raw rocprof output
rm -rf "$OUT/gather_trace" "$OUT/gather_fetch" "$OUT/gather_write"
cut -c1-1500 "$OUT/bench_taobao10.json"
cut -c1-600 "$OUT/bench_taobao10_2ranks_shared.json"
