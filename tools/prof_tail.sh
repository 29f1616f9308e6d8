#!/bin/bash
# rocprofv3 passes over tools/run_tail.py: kernel trace + PMC counters (separate passes, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE cannot share one).  Usage: tools/prof_tail.sh TAG [lib.so]
# Results: gpurun_out/TAG/{trace,pmc_*}; summarise with tools/prof_summary.py / tools/pmc_summary.py.
TAG=${1:-tail}
export TMPDIR=/tmp
[ -n "$2" ] && export PDS_HIP_LIB=$2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
python tools/run_tail.py 5 > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python tools/run_tail.py 5 > $OUT/trace.log 2>&1
i=0
for C in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$i -- python tools/run_tail.py 2 > $OUT/pmc_$i.log 2>&1
done
python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "tools/run_tail.py 5 (Regularization + SubpixelMap tail only, config 2)" > /dev/null 2>&1
python tools/pmc_summary.py $OUT > $OUT/pmc.txt 2>&1
