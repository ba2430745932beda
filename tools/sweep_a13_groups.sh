mkdir -p gpurun_out/r03t
run() { env "$@" MAMDR_BENCH_ROW_SCALE=0.1 timeout 400 python bench.py --workload amazon13 --steps 2 --warmup 1 --cpu-budget 0 --no-targets --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('amazon13 $*', round(d['value']), round(d['us_per_domain_step'],2))" | tee -a gpurun_out/r03t/a13_groups.txt; }
for i in 1 2; do
run X=0
run MAMDR_RPG=1024
run MAMDR_MAX_GROUPS=32 MAMDR_RPG=256
done
