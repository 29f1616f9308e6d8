#!/bin/bash
# 48-plane conv2d_x3 launch times inside the sequential hot path, per library variant: tools/seq_variants.sh v1 v2 ...
export TMPDIR=/tmp
for v in base "$@"; do
  if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
  OUT=$PWD/gpurun_out/seqv_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -- python bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-pipeline --windows 1 > $OUT/seq.log 2>&1
  python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "$v" > /dev/null 2>&1
  rm -rf $OUT/trace
  echo "== $v"; grep "conv2d_x3.*131072" $OUT/kernels.txt | cut -c1-50,100-160
done
