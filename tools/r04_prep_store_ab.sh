cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in default prepplain; do
  if [ $v = prepplain ]; then export MAMDR_LIB_PATH=$PWD/mamdr_amd/build/variants/libprepplain.so; else unset MAMDR_LIB_PATH; fi
  python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-targets 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), round(d['us_per_domain_step'],3), {k.split(' ')[0]:(v['launches'],round(v['avg_us'],2)) for k,v in d['kernels_avg_us'].items() if isinstance(v,dict)})"
done; done
