#!/bin/bash
# SQ / memory counters of one kernel of the tail (tools/run_tail.py): tools/pmc_tail_kernel.sh TAG KERNEL_SUBSTRING [lib.so]
export TMPDIR=/tmp
TAG=$1; WANT=$2
[ -n "$3" ] && export PDS_HIP_LIB=$3
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
         "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p_$i -- python tools/run_tail.py 2 > $OUT/log_$i.txt 2>&1
done
python tools/pmc_summary.py $OUT "$WANT" > $OUT/pmc.txt 2>&1
rm -rf $OUT/p_?
cat $OUT/pmc.txt
