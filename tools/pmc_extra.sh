#!/bin/bash
# extra SQ counter passes over tools/run_tail.py: tools/pmc_extra.sh TAG [lib.so]
TAG=${1:-extra}
export TMPDIR=/tmp
[ -n "$2" ] && export PDS_HIP_LIB=$2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
i=0
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/x_$i -- python tools/run_tail.py 2 > $OUT/x_$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/pmc_extra.txt 2>&1
