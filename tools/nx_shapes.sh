#!/bin/bash
# A/B of conv3d_nx.hip's tile shapes (PDS_CONV3D_NX_CFG) on the Regularization tail at config 2: kernel durations under rocprofv3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/nx_shapes
rm -rf $OUT; mkdir -p $OUT
for cfg in ${@:-0 1 2 3 4}; do
  PDS_DEBUG_SWITCHES=1 PDS_CONV3D_NX_CFG=$cfg rocprofv3 --kernel-trace --stats -d $OUT/t$cfg -- python tools/run_tail.py 10 > $OUT/log$cfg.txt 2>&1
  python tools/prof_summary.py $OUT/t$cfg $OUT/k$cfg.txt "cfg $cfg" > /dev/null 2>&1
  echo "cfg $cfg: $(grep conv3d_nx $OUT/k$cfg.txt | head -1 | awk '{print $(NF-3), $(NF-2), $(NF-1)}') | $(tail -1 $OUT/log$cfg.txt)"
  rm -rf $OUT/t$cfg
done
