#!/bin/bash
# kernel summary of the full-size training step: tools/prof_train.sh NAME -> gpurun_out/NAME/kernels.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-train}
rm -rf $OUT; mkdir -p $OUT
python tools/train_step_bench.py > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python tools/train_step_bench.py > $OUT/prof.log 2>&1
python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "tools/train_step_bench.py: 3 full-size training steps (960x540, D=192, one GPU, SubpixelCrossEntropy)" > /dev/null 2>&1
rm -rf $OUT/trace
cat $OUT/plain.log | tail -3
head -45 $OUT/kernels.txt | cut -c1-100,104-150
