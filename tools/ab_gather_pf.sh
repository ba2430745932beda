# A/B of the gather riders in k_update's launch (Taobao-30, slab path): MAMDR_NO_GATHER_PF=1 = off
mkdir -p gpurun_out/r03r
for m in 1 0 1 0 1 0; do
  MAMDR_NO_GATHER_PF=$m timeout 500 python bench.py --workload taobao30 --steps 8 --warmup 2 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_avg_us']; print('taobao30 no_gather_pf=$m', round(d['value']), round(d['us_per_domain_step'],2), {n: round(v['avg_us'],2) for n,v in k.items() if isinstance(v,dict)})" | tee -a gpurun_out/r03r/ab_gather_pf.txt
done
