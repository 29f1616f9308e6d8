#!/bin/bash
# SQ / LDS / memory counter passes over any command: tools/pmc_cmd.sh TAG KERNEL_SUBSTRING cmd...  -> gpurun_out/TAG/pmc.txt
TAG=$1; WANT=$2; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$i -- "$@" > $OUT/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $OUT "$WANT" > $OUT/pmc.txt 2>&1
rm -rf $OUT/pmc_?
cat $OUT/pmc.txt
