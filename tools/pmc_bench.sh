#!/bin/bash
# PMC passes over a short sequential bench run: tools/pmc_bench.sh TAG  (results gpurun_out/TAG/pmc.txt)
TAG=${1:-pmcb}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
i=0
for C in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$i -- python bench.py --steps 2 --warmup 1 --kernel-reps 1 --no-cpu-baseline --no-train-record --no-pipeline --windows 1 > $OUT/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/pmc.txt 2>&1
