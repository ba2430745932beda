#!/bin/bash
exec < /dev/null
# round-6 measurement pass (through gpurun; round 5's stages + `sq` and `gis`): tools/r06_measure.sh <tag> [stages]
#   stages (default "test bench prof pmc gather"): test = pytest -m gpu; bench = the default bench line (Taobao-10 +
#   targets taobao30 / amazon6 / amazon13 at full rows + gather + CPU baseline) and the 2-ranks-on-one-GPU line;
#   prof = rocprofv3 --kernel-trace --stats of the four BASELINE workloads; pmc = FETCH_SIZE / WRITE_SIZE passes
#   (separate, --kernel-trace only, the program directly after `--`); gather = trace + counters of tools/gather_hbm.py;
#   graph = throughput + kernel trace of the generic-layer towers (tools/graph_bench.py);
#   sq = SQ / TCP counters of the step kernels (MFMA-busy, L2 -> L1 requests, waitcnt share) on Taobao-10 bs 1,024,
#   Taobao-30 bs 4,096, Amazon-6 bs 1,024 and Amazon-13 bs 8,192 -> pmc_sq_latest.json (bench.py puts them beside the roofline fractions); gis = the in-step gather
#   phase from a -DMAMDR_STAMPS build (tools/r06_gather_in_step.py) -> gather_in_step.json.
# Every command is bounded by `timeout` and reads /dev/null: a hung profiler must not eat the GPU budget.
# Summaries land in gpurun_out/<tag>/; copy what is to be judged into profiles/ by hand.
TAG=${1:-r06z}
STAGES=${2:-"test bench prof pmc gather graph sq gis"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
has() { [[ " $STAGES " == *" $1 "* ]]; }
db() { find "$OUT/$1" -name "*.db" | head -1; }
if has test; then
    timeout 1500 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.log" 2>&1
    tail -3 "$OUT/pytest_gpu.log"
    grep -E "worst|bs [0-9]+:" "$OUT/pytest_gpu.log"
fi
if has bench; then
    T0=$(date +%s)
    timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
    echo "default bench.py run: $(( $(date +%s) - T0 )) s wall"
    MAMDR_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-budget 0 --no-targets > "$OUT/bench_taobao10_2ranks_shared.json" 2> "$OUT/bench_2ranks.err"
fi
cd /tmp
if has prof; then
    # (--lanes 0: the lane leg stretches the kernels it shares the device with; these summaries are the single chain's)
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof10" -o run -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --cpu-budget 0 --no-targets --lanes 0 > "$OUT/prof10.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof30" -o run -- python3 "$REPO/bench.py" --workload taobao30 --steps 2 --warmup 1 --cpu-budget 0 --no-targets --no-profile --lanes 0 > "$OUT/prof30.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/profa6" -o run -- python3 "$REPO/bench.py" --workload amazon6 --steps 1 --warmup 1 --cpu-budget 0 --no-targets --no-profile --lanes 0 > "$OUT/profa6.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/profa13" -o run -- python3 "$REPO/bench.py" --workload amazon13 --steps 1 --warmup 1 --cpu-budget 0 --no-targets --no-profile --lanes 0 > "$OUT/profa13.log" 2>&1
fi
if has pmc; then
    for W in ${PMC_WORKLOADS:-taobao10 taobao30 amazon6 amazon13}; do
        for C in FETCH_SIZE WRITE_SIZE; do
            # (Amazon-13's full-row epoch is 39 K steps x 10 kernels of counter records: rocprofv3 itself crashed on it;
            # the bytes a launch moves do not depend on how many rows an epoch has -> 10 % of the rows for that pass)
            RS=1; [ $W = amazon13 ] && RS=0.1
            MAMDR_BENCH_ROW_SCALE=$RS timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_${W}_$C" -o run -- python3 "$REPO/bench.py" --workload $W --steps 1 --warmup 1 --cpu-budget 0 --no-profile --no-targets --lanes 0 > "$OUT/pmc_${W}_$C.log" 2>&1
        done
    done
fi
if has gather; then
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/gather_trace" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_trace.log" 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/gather_fetch" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_fetch.log" 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/gather_write" -o run -- python3 "$REPO/tools/gather_hbm.py" > "$OUT/gather_write.log" 2>&1
fi
if has graph; then
    timeout 900 python3 "$REPO/tools/graph_bench.py" 3 2>/dev/null | grep tower > "$OUT/graph_bench.jsonl"
    # (NFM / PNN run on the step kernels by default since round 4: their generic-layer twins for comparison)
    MAMDR_PNN_ENGINE=graph MAMDR_NFM_ENGINE=graph timeout 400 python3 "$REPO/tools/graph_bench.py" 3 nfm,pnn 2>/dev/null | grep tower >> "$OUT/graph_bench.jsonl"
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/graph_trace" -o run -- python3 "$REPO/tools/graph_bench.py" 1 all inproc > "$OUT/graph_trace.log" 2>&1
fi
if has sq; then
    for WB in ${SQ_WORKLOADS:-"taobao10 1024" "taobao30 4096" "amazon6 1024" "amazon13 8192"}; do
        set -- $WB
        i=0
        for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
                   "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
            i=$((i+1))
            timeout 200 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/sq_$1_g$i" -o run -- python3 "$REPO/tools/pmc_steps.py" $1 $2 12 > "$OUT/sq_$1_g$i.log" 2>&1
        done
    done
fi
cd "$REPO"
if has sq; then
    python - "$OUT" <<'PY'
import collections, glob, json, sqlite3, sys
out = sys.argv[1]
try:
    merged = json.load(open("profiles/pmc_sq_latest.json"))
except Exception:
    merged = {}
for w in ("taobao10", "taobao30", "amazon6", "amazon13"):
    for db in sorted(glob.glob(out + "/sq_%s_g*/**/*.db" % w, recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for k, n, v in sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection"):
            if "mamdr" in k:
                agg[k.split("(")[0].replace("void ", "").replace("mamdr::", "")][n].append(float(v))
        for k, cs in agg.items():
            # (the first launch of a kernel in a pass runs cold: the median over the launches is the steady state)
            for key in ([k] if w == "taobao10" else []) + ["%s@%s" % (k, w)]:
                ent = merged.setdefault(key, {})
                for n, v in cs.items():
                    v = sorted(v)
                    ent[n] = v[len(v) // 2]
                ent["launches_in_pass"] = len(next(iter(cs.values())))
json.dump(merged, open(out + "/pmc_sq_latest.json", "w"), indent=1)
for k in sorted(merged):
    if k.startswith(("k_tower", "k_wgrad", "k_update", "k_star", "k_emb")):
        print(k, {n: round(v) for n, v in merged[k].items()})
PY
    rm -rf "$OUT"/sq_*_g?
fi
if has gis; then
    timeout 900 python tools/r06_gather_in_step.py "$OUT/gather_in_step.json" > "$OUT/gather_in_step.log" 2>&1
    tail -4 "$OUT/gather_in_step.log"
fi
if has graph; then
    G=$(db graph_trace); [ -n "$G" ] && python tools/rocpd_summary.py stats "$G" "$OUT/kernel_stats_graph_towers.csv"
    rm -rf "$OUT/graph_trace"
    python3 - "$OUT/graph_bench.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["tower"], d["engine"][:12], round(d["value"]), "steps/s", round(d["us_per_domain_step"], 1), "us; whole-step frac", round(d["roofline"]["frac"], 3),
          "launches/step", d["roofline"]["launches_per_step"])
PY
fi
if has prof; then
    python tools/rocpd_summary.py stats "$(db prof10)" "$OUT/kernel_stats_taobao10.csv"
    python tools/rocpd_summary.py stats "$(db prof30)" "$OUT/kernel_stats_taobao30.csv"
    python tools/rocpd_summary.py stats "$(db profa6)" "$OUT/kernel_stats_amazon6.csv"
    python tools/rocpd_summary.py stats "$(db profa13)" "$OUT/kernel_stats_amazon13.csv"
    for W in taobao10 taobao30 amazon6 amazon13; do echo "== $W"; grep "mamdr::" "$OUT/kernel_stats_$W.csv" | cut -c1-150 | head -14; done
fi
if has pmc; then
    for W in ${PMC_WORKLOADS:-taobao10 taobao30 amazon6 amazon13}; do
        python tools/rocpd_summary.py pmc "$(db pmc_${W}_FETCH_SIZE)" "$(db pmc_${W}_WRITE_SIZE)" "$OUT/pmc_hbm_$W.json"
    done
    # one file keyed the way bench.py looks kernels up: Taobao-10's kernels by short name, the other workloads' towers
    # under their instance names, every other kernel of theirs as <name>@<workload>
    python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
import os
try:
    merged = json.load(open(out + "/pmc_hbm_latest.json"))
except Exception:
    merged = {}
if os.path.exists(out + "/pmc_hbm_taobao10.json"):
    merged.update(json.load(open(out + "/pmc_hbm_taobao10.json")))
for w in ("taobao30", "amazon6", "amazon13"):
    if not os.path.exists(out + "/pmc_hbm_%s.json" % w):
        continue
    for k, v in json.load(open(out + "/pmc_hbm_%s.json" % w)).items():
        if k.startswith("k_tower") and k not in merged:
            merged[k] = v
        merged["%s@%s" % (k, w)] = v
json.dump(merged, open(out + "/pmc_hbm_latest.json", "w"), indent=1)
PY
fi
if has gather; then
    python tools/rocpd_summary.py stats "$(db gather_trace)" "$OUT/kernel_stats_gather_amazon6.csv"
    [ -f "$OUT/pmc_hbm_latest.json" ] || cp profiles/pmc_hbm_latest.json "$OUT/pmc_hbm_latest.json"
    python tools/rocpd_summary.py pmc1 "$(db gather_fetch)" "$(db gather_write)" k_gather k_gather@amazon6 "$OUT/pmc_hbm_latest.json"
fi
rm -rf "$OUT"/prof10 "$OUT"/prof30 "$OUT"/profa6 "$OUT"/profa13 "$OUT"/pmc_*_FETCH_SIZE "$OUT"/pmc_*_WRITE_SIZE "$OUT"/gather_trace "$OUT"/gather_fetch "$OUT"/gather_write
if has bench; then
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("taobao10", round(d["value"]), "steps/s", round(d["us_per_domain_step"], 2), "us/step; tower frac", round(d["roofline"]["frac"], 3),
      "| cpu", round(d["cpu_baseline"]["value"], 1), d["cpu_baseline"]["spread"], "x", round(d["gpu_over_cpu"], 1), "| gather", round(d["gather"]["frac"], 3), "host ms/epoch", d["host_ms_per_epoch"])
print({k: v for k, v in d["kernels_avg_us"].items() if k != "_rated"})
for r in d["kernels_avg_us"].get("_rated", []): print("   rated:", r["kernel"], r["bound"], round(r["frac"], 3))
for w, t in d["targets"].items():
    print(w, round(t["value"]), round(t["us_per_domain_step"], 2), "us/step; tower frac", round(t["roofline"]["frac"], 3), "cpu", round(t["cpu_baseline"]["value"], 2),
          "x", round(t["gpu_over_cpu"], 1), "host ms/epoch", t["host_ms_per_epoch"])
    print("   ", {k: (round(v["us_per_domain_step"], 2) if isinstance(v, dict) else round(v, 2)) for k, v in t["kernels_avg_us"].items() if k != "_rated"})
    for r in t["kernels_avg_us"].get("_rated", []): print("    rated:", r["kernel"], r["bound"], round(r["frac"], 3))
d2 = json.load(open("$OUT/bench_taobao10_2ranks_shared.json")); print("2 ranks shared:", round(d2["value"]), d2.get("partition_speedup_bound"), d2.get("host_ms_per_epoch"))
PY
fi
