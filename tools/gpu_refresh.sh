#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun):
#   parity tests, the three bench workloads, a rocprofv3 kernel trace and two PMC passes
#   (FETCH_SIZE, WRITE_SIZE separately, no other trace domains) of the default bench command.
# Everything lands under gpurun_out/$TAG; copy the summaries into profiles/ afterwards.
TAG=${1:-r1c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
python bench.py > "$OUT/bench_taobao10.json" 2> "$OUT/bench_taobao10.err"
python bench.py --workload taobao30 --steps 10 > "$OUT/bench_taobao30.json" 2> "$OUT/bench_taobao30.err"
python bench.py --workload amazon6 --steps 3 --warmup 1 > "$OUT/bench_amazon6.json" 2> "$OUT/bench_amazon6.err"
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o run -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --cpu-budget 0 > "$OUT/prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python tools/summarize_pmc.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_hbm.json" > /dev/null 2>&1
# keep what travels back small: traces are large, the stats and counter summaries are not
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
cat "$OUT"/bench_*.json | cut -c1-600
ls -la "$OUT" "$OUT"/prof/* | head -40
