#!/bin/bash
# A/B of debug switches on the sequential hot path: tools/ab_env.sh "KERNEL_REGEX" "NAME=VAR=VALUE[,VAR=VALUE]" ...
# ("base" runs first with no switch set); per-kernel times from rocprofv3, then the pipelined bench value.
export TMPDIR=/tmp
FILT=$1; shift
for spec in base "$@"; do
  name=${spec%%=*}
  ( export PDS_DEBUG_SWITCHES=1
    if [ "$spec" != base ]; then IFS=',' read -ra kv <<< "${spec#*=}"; for a in "${kv[@]}"; do export "$a"; done; fi
    OUT=$PWD/gpurun_out/ab_$name; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --kernel-trace --stats -d $OUT/trace -- python bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-pipeline --windows 1 > $OUT/seq.log 2>&1
    python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "$spec" > /dev/null 2>&1
    rm -rf $OUT/trace
    echo "== $spec"; grep -E "$FILT" $OUT/kernels.txt | head -14 | cut -c1-70,100-150
    python bench.py --no-cpu-baseline --no-train-record --windows 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   pairs/s', round(d['value'],1), 'sequential ms', round(d['ms_per_frame'],3), d['pipelined_equals_sequential'], d.get('parity'))"
  )
done
