#!/bin/bash
# same-box A/B of two source trees (each with its own built library): tools/ab_trees.sh build/r3tree . [rounds]
#   prints the pipelined pairs/s (median of 3 windows) and the sequential ms per pair of each tree, alternating
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for t in $A $B; do
    (cd $t && python bench.py --no-cpu-baseline --windows 3 --kernel-reps 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', 'pairs/s', round(d['value'],1), 'median', round(d['windows']['median'],1), 'seq ms', round(d['ms_per_frame'],3), 'x3 launch ms', round(d['roofline']['launch_ms'],3), d['pipelined_equals_sequential'])")
  done
done
