#!/bin/bash
# kernel durations of Matching alone at config 2 under rocprofv3: tools/prof_matching.sh [filter]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_matching
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -- python tools/run_matching.py 10 > $OUT/log.txt 2>&1
python tools/prof_summary.py $OUT/t $OUT/kernels.txt "tools/run_matching.py 10" > /dev/null 2>&1
rm -rf $OUT/t
grep -E "${1:-.}" $OUT/kernels.txt | head -30 | cut -c1-100,104-150
grep matching: $OUT/log.txt
