#!/bin/bash
# hardware counters of the step kernels over a dozen steps (separate short passes; --pmc with --kernel-trace only):
#   tools/pmc_tower.sh <shape> <batch> <tag>
S=${1:-taobao30}; B=${2:-4096}; TAG=${3:-pmc}
OUT=/tmp/$TAG; mkdir -p $OUT; REPO=$PWD; RES=$REPO/gpurun_out/$TAG.txt; : > $RES
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o run -- python3 $REPO/tools/pmc_steps.py $S $B 12 > $OUT/g$i.log 2>&1
  tail -2 $OUT/g$i.log >> $RES; python3 $REPO/tools/pmc_dump.py $OUT/g$i/run_results.db k_ >> $RES 2>&1
done
cat $RES | cut -c1-700
