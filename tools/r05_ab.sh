#!/bin/bash
exec < /dev/null
# round-5 measurements (through gpurun): tools/r05_ab.sh <tag> [stages]
#   stamps  the per-phase s_memtime profile of the headline tower k_tower4<false,false,true,true> (9 phases; VERDICT r04
#           item 7) and of the 16-row tower at 4,096 rows, from the -DMAMDR_STAMPS build (prebuilt: tools/stamp_tower.py --build-only)
#   groups  Taobao-30 bs 4,096: k_wgrad's row-group count (= slabs written / re-read by k_update) A/B through MAMDR_RPG,
#           2 repeats interleaved
#   a13     Amazon-13 bs 8,192: the same for k_wgrad_reduce (item 6), 10 % of the rows
TAG=${1:-r05e}
STAGES=${2:-"stamps groups a13"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
has() { [[ " $STAGES " == *" $1 "* ]]; }
run() {   # file, name, workload, steps, env...
    local file=$1 name=$2 wl=$3 steps=$4; shift 4
    env "$@" timeout 400 python bench.py --workload $wl --steps $steps --warmup 1 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', round(d['value'],1), 'steps/s', round(d['us_per_domain_step'],3), 'us/step', {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})" >> "$OUT/$file"
}
if has stamps; then
    MAMDR_STAMPS_PREBUILT=1 timeout 300 python tools/stamp_tower.py taobao10 1024 > "$OUT/stamps_tower4_taobao10_bs1024.txt" 2>&1
    MAMDR_STAMPS_PREBUILT=1 MAMDR_TOWER_TILE=16 timeout 300 python tools/stamp_tower.py taobao30 4096 > "$OUT/stamps_tower16_taobao30_bs4096.txt" 2>&1
    tail -n 30 "$OUT/stamps_tower4_taobao10_bs1024.txt"
fi
if has groups; then
    for rep in 1 2; do
        run ab_groups_taobao30.txt rpg256_16groups taobao30 3 A=1
        run ab_groups_taobao30.txt rpg320_13groups taobao30 3 MAMDR_RPG=320
        run ab_groups_taobao30.txt rpg416_10groups taobao30 3 MAMDR_RPG=416
        run ab_groups_taobao30.txt rpg512_8groups taobao30 3 MAMDR_RPG=512
        run ab_groups_taobao30.txt rpg1024_4groups taobao30 3 MAMDR_RPG=1024
    done
    cat "$OUT/ab_groups_taobao30.txt"
fi
if has a13; then
    for rep in 1 2; do
        run ab_groups_amazon13.txt default amazon13 1 MAMDR_BENCH_ROW_SCALE=0.1
        run ab_groups_amazon13.txt rpg1024_8groups amazon13 1 MAMDR_BENCH_ROW_SCALE=0.1 MAMDR_RPG=1024
        run ab_groups_amazon13.txt rpg2048_4groups amazon13 1 MAMDR_BENCH_ROW_SCALE=0.1 MAMDR_RPG=2048
        run ab_groups_amazon13.txt rpg256_32groups amazon13 1 MAMDR_BENCH_ROW_SCALE=0.1 MAMDR_RPG=256 MAMDR_MAX_GROUPS=32
    done
    cat "$OUT/ab_groups_amazon13.txt"
fi
