#!/bin/bash
# round-2 measurement pass (through gpurun): tools/r02_measure.sh <tag>
# bench lines (default = Taobao-10 + targets.taobao30 + Amazon-6-sized gather + CPU baseline; 2 ranks sharing the GPU),
# rocprofv3 kernel traces of the default command and of Taobao-30, FETCH_SIZE / WRITE_SIZE passes (separate, with
# --kernel-trace only) of the default command and of the gather program.  Summaries -> profiles/ by hand.
TAG=${1:-r02m}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python bench.py > "$OUT/bench_taobao10.json" 2> "$OUT/bench_taobao10.err"
python bench.py --workload taobao30 --steps 10 --no-targets > "$OUT/bench_taobao30.json" 2> "$OUT/bench_taobao30.err"
MAMDR_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-budget 0 --no-targets > "$OUT/bench_taobao10_2ranks_shared.json" 2> "$OUT/bench_2ranks.err"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof10" -o run -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --cpu-budget 0 --no-targets > "$OUT/prof10.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/prof30" -o run -- python3 "$REPO/bench.py" --workload taobao30 --steps 2 --warmup 1 --cpu-budget 0 --no-targets --no-profile > "$OUT/prof30.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile --no-targets > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --cpu-budget 0 --no-profile --no-targets > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc30_fetch" -o run -- python3 "$REPO/bench.py" --workload taobao30 --steps 1 --warmup 1 --cpu-budget 0 --no-profile --no-targets > "$OUT/pmc30_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc30_write" -o run -- python3 "$REPO/bench.py" --workload taobao30 --steps 1 --warmup 1 --cpu-budget 0 --no-profile --no-targets > "$OUT/pmc30_write.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/gather_trace" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/gather_fetch" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/gather_write" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_write.log" 2>&1
cd "$REPO"
db() { find "$OUT/$1" -name "*.db" | head -1; }
python tools/rocpd_summary.py stats "$(db prof10)" "$OUT/kernel_stats_taobao10.csv"
python tools/rocpd_summary.py stats "$(db prof30)" "$OUT/kernel_stats_taobao30.csv"
python tools/rocpd_summary.py stats "$(db gather_trace)" "$OUT/kernel_stats_gather_amazon6.csv"
python tools/rocpd_summary.py pmc "$(db pmc_fetch)" "$(db pmc_write)" "$OUT/pmc_hbm_taobao10.json"
cp "$OUT/pmc_hbm_taobao10.json" "$OUT/pmc_hbm_latest.json"
python tools/rocpd_summary.py pmc1 "$(db gather_fetch)" "$(db gather_write)" k_gather k_gather@amazon6 "$OUT/pmc_hbm_latest.json"
# the 16-row tower of the Taobao-30 target (bench.py: targets.taobao30.tower.traffic)
python tools/rocpd_summary.py pmc1 "$(db pmc30_fetch)" "$(db pmc30_write)" "k_tower<true, 0, false, false>" "k_tower<true, 0, false, false>" "$OUT/pmc_hbm_latest.json"
rm -rf "$OUT"/prof10 "$OUT"/prof30 "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc30_fetch "$OUT"/pmc30_write "$OUT"/gather_trace "$OUT"/gather_fetch "$OUT"/gather_write
grep "mamdr::" "$OUT/kernel_stats_taobao10.csv" | cut -c1-160 | head -8
grep "mamdr::" "$OUT/kernel_stats_taobao30.csv" | cut -c1-160 | head -8
python - <<PY
import json
d = json.load(open("$OUT/bench_taobao10.json"))
print("taobao10", round(d["value"]), "steps/s", round(d["us_per_domain_step"], 2), "us/step; tower frac", round(d["roofline"]["frac"], 3),
      "| cpu", round(d["cpu_baseline"]["value"], 1), "x", round(d["gpu_over_cpu"], 1), "| gather", round(d["gather"]["frac"], 3), d["gather"]["traffic"])
t = d["targets"]["taobao30"]
print("taobao30 (in-run)", round(t["value"]), round(t["us_per_domain_step"], 2), "tower frac", round(t["tower"]["frac"], 3), "cpu", round(t["cpu_baseline"]["value"], 1), "x", round(t["gpu_over_cpu"], 1))
print(d["kernels_avg_us"])
d2 = json.load(open("$OUT/bench_taobao10_2ranks_shared.json")); print("2 ranks shared:", round(d2["value"]), d2.get("partition_speedup_bound"))
PY
