#!/bin/bash
# same-box per-kernel A/B of two source trees on the sequential hot path (rocprofv3 kernel trace):
#   tools/ab_trees_prof.sh build/r3tree .      -> gpurun_out/abp_{a,b}.txt + side-by-side table
export TMPDIR=/tmp
ROOT=$PWD
SEQ="bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-pipeline --windows 1"
i=0
for t in "$@"; do
  i=$((i+1)); tag=$(echo abcd | cut -c$i)
  OUT=$ROOT/gpurun_out/abp_$tag; rm -rf $OUT; mkdir -p $OUT
  (cd $t && rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $SEQ > $OUT/seq.log 2>&1)
  python $ROOT/tools/prof_summary.py $OUT/trace $ROOT/gpurun_out/abp_$tag.txt "$t: $SEQ" > /dev/null 2>&1
  rm -rf $OUT
done
python $ROOT/tools/cmp_kernels.py $(for j in $(seq 1 $i); do echo $ROOT/gpurun_out/abp_$(echo abcd | cut -c$j).txt; done) | head -60
