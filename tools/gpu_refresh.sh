#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun):
#   parity tests, the four bench workloads (+ the dense-sweep variant of Amazon-6), a rocprofv3 kernel
#   trace and two PMC passes (FETCH_SIZE, WRITE_SIZE separately, no other trace domains) of the default
#   bench command and of the dense Amazon-6 run (k_emb_sweep traffic).
# Everything lands under gpurun_out/$TAG; summaries: tools/rocpd_summary.py -> profiles/.
TAG=${1:-r1e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
python bench.py > "$OUT/bench_taobao10.json" 2> "$OUT/bench_taobao10.err"
python bench.py --workload taobao30 --steps 10 > "$OUT/bench_taobao30.json" 2> "$OUT/bench_taobao30.err"
python bench.py --workload amazon6 --steps 3 --warmup 1 > "$OUT/bench_amazon6.json" 2> "$OUT/bench_amazon6.err"
MAMDR_DENSE_ADAM=1 python bench.py --workload amazon6 --steps 3 --warmup 1 --cpu-budget 0 > "$OUT/bench_amazon6_dense.json" 2> "$OUT/bench_amazon6_dense.err"
python bench.py --workload amazon13 --steps 2 --warmup 1 > "$OUT/bench_amazon13.json" 2> "$OUT/bench_amazon13.err"
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o run -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --cpu-budget 0 > "$OUT/prof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile > "$OUT/pmc_write.log" 2>&1
for w in taobao30 amazon6 amazon13; do
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_$w" -o run -- python3 "$REPO/bench.py" --workload $w --steps 1 --warmup 1 --cpu-budget 0 --no-profile > "$OUT/prof_$w.log" 2>&1
done
export MAMDR_DENSE_ADAM=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch_amz" -o run -- python3 "$REPO/bench.py" --workload amazon6 --steps 1 --warmup 0 --cpu-budget 0 --no-profile > "$OUT/pmc_fetch_amz.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write_amz" -o run -- python3 "$REPO/bench.py" --workload amazon6 --steps 1 --warmup 0 --cpu-budget 0 --no-profile > "$OUT/pmc_write_amz.log" 2>&1
unset MAMDR_DENSE_ADAM
cd "$REPO"
cat "$OUT"/bench_*.json | cut -c1-300
du -sh "$OUT"
