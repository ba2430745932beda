#!/bin/bash
# rocprofv3 kernel trace of ONE generic-layer tower (argument: tower name), Taobao-10 config, batch 1024
T=${1:-shared_bottom}
OUT=$PWD/gpurun_out/r04g_$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $T -- python3 $OLDPWD/tools/graph_bench.py 2 $T inproc > $OUT/bench.jsonl 2> $OUT/err.txt
cd $OLDPWD
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cat $OUT/bench.jsonl | cut -c1-300
python3 - $OUT/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print('%-70s calls %6s avg %8.2f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 100*int(r['TotalDurationNs'])/tot))
PY
