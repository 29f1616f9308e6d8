#!/bin/bash
# A/B of library variants on the sequential hot path: per-kernel times (rocprofv3 summary) and the pipelined bench value.
#   tools/ab_seq.sh "KERNEL_REGEX" v1 v2 ...     (variants from tools/build_variant_one.sh; "base" = the tree's library)
export TMPDIR=/tmp
FILT=$1; shift
for v in base "$@"; do
  if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
  OUT=$PWD/gpurun_out/ab_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -- python bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-pipeline --windows 1 > $OUT/seq.log 2>&1
  python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "$v" > /dev/null 2>&1
  rm -rf $OUT/trace
  echo "== $v"; grep -E "$FILT" $OUT/kernels.txt | head -12 | cut -c1-60,100-150
  python bench.py --no-cpu-baseline --no-train-record --windows 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   pairs/s', round(d['value'],1), 'median', round(d['windows']['median'],1), 'sequential ms', round(d['ms_per_frame'],3), d['pipelined_equals_sequential'])"
done
