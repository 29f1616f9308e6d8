#!/bin/bash
# tools/timeline.sh -> gpurun_out/timeline/overlap.txt (see tools/timeline_overlap.py)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-train-record --windows 1 --kernel-reps 1"
rocprofv3 --kernel-trace -d $OUT/seq -- python bench.py $ARGS --no-pipeline > $OUT/seq.log 2>&1
rocprofv3 --kernel-trace -d $OUT/pipe -- python bench.py $ARGS > $OUT/pipe.log 2>&1
python tools/timeline_overlap.py $OUT/seq $OUT/pipe $OUT/overlap.txt
rm -rf $OUT/seq $OUT/pipe
