#!/bin/bash
# A/B of conv2d_t8w's tile height (PDS_CONV2D_T8W_ROWS = 8 | 6) on Matching alone at config 2 under rocprofv3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/t8w_rows
rm -rf $OUT; mkdir -p $OUT
for rows in 8 6; do
  PDS_DEBUG_SWITCHES=1 PDS_CONV2D_T8W_ROWS=$rows rocprofv3 --kernel-trace --stats -d $OUT/t$rows -- python tools/run_matching.py 10 > $OUT/log$rows.txt 2>&1
  python tools/prof_summary.py $OUT/t$rows $OUT/k$rows.txt "rows $rows" > /dev/null 2>&1
  echo "rows $rows: $(grep conv2d_t8w $OUT/k$rows.txt | head -1 | awk '{print $(NF-3), $(NF-2), $(NF-1)}') | $(grep matching: $OUT/log$rows.txt)"
  rm -rf $OUT/t$rows
done
