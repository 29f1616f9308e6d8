#!/bin/bash
# kernel summary of any command: tools/prof_cmd.sh NAME cmd...  -> gpurun_out/NAME/kernels.txt
export TMPDIR=/tmp
NAME=$1; shift
OUT=$PWD/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -- "$@" > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "$*" > /dev/null 2>&1
rm -rf $OUT/trace
head -40 $OUT/kernels.txt | cut -c1-100,104-150
