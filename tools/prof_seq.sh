#!/bin/bash
# sequential-pairs kernel summary of the hot path: tools/prof_seq.sh NAME  -> gpurun_out/NAME/kernels.txt + bench line
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-seq}
rm -rf $OUT; mkdir -p $OUT
SEQ="bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-sub-records --no-pipeline --windows 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $SEQ > $OUT/seq.log 2>&1
python tools/prof_summary.py $OUT/trace $OUT/kernels.txt "$SEQ" > /dev/null 2>&1
rm -rf $OUT/trace
python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-train-record --no-sub-records > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','ms_per_frame','full_forward_ms','pipelined_equals_sequential')}, d.get('parity'))"
head -45 $OUT/kernels.txt | cut -c1-100,104-150
