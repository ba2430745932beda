#!/bin/bash
exec < /dev/null
# WRITE_SIZE of the 16-row tower at 4,096 rows: write-through (sc1, the default) against plain workspace stores
# (variant library built with -DMAMDR_WS_PLAIN by tools/build_variant.sh wsplain -DMAMDR_WS_PLAIN, in-tree before gpurun)
TAG=${1:-r04w}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD
cd /tmp; export TMPDIR=/tmp
for V in default wsplain; do
    if [ $V = wsplain ]; then export MAMDR_LIB_PATH=$REPO/mamdr_amd/build/variants/libwsplain.so; else unset MAMDR_LIB_PATH; fi
    for C in WRITE_SIZE FETCH_SIZE; do
        timeout 200 rocprofv3 --pmc $C --kernel-trace -d $OUT/${V}_$C -o run -- python3 $REPO/tools/pmc_steps.py taobao30 4096 12 > $OUT/${V}_$C.log 2>&1
        echo "== $V $C" >> $OUT/write_ab.txt
        python3 $REPO/tools/pmc_dump.py $(find $OUT/${V}_$C -name "*.db" | head -1) k_tower >> $OUT/write_ab.txt 2>&1
        rm -rf $OUT/${V}_$C
    done
done
cat $OUT/write_ab.txt | cut -c1-400
