mkdir -p gpurun_out/r03p
for w in taobao10 taobao30 amazon6 amazon13; do
  S=2; [ $w = taobao10 ] && S=20; [ $w = taobao30 ] && S=8
  for m in 1 0 1 0; do
    MAMDR_BENCH_NO_PREFETCH=$m timeout 500 python bench.py --workload $w --steps $S --warmup 1 --cpu-budget 0 --no-targets --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w noprefetch=$m', round(d['value']), round(d['us_per_domain_step'],2), 'host_prep', round(d.get('host_prep_ms_per_epoch',0),2), 'epoch', round(d['ms_per_step'],1))" | tee -a gpurun_out/r03p/ab.txt
  done
done
