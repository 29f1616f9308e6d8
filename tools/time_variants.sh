#!/bin/bash
# kernel-trace timing of tools/run_tail.py (or RUNNER=tools/run_matching.py) under several library variants:
#   tools/time_variants.sh TAG KERNELFILTER v1 v2 ...
TAG=$1; FILT=$2; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for v in base "$@"; do
  if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/tr_$v -- python ${RUNNER:-tools/run_tail.py} 5 > gpurun_out/$TAG/tr_$v.log 2>&1
  python tools/prof_summary.py gpurun_out/$TAG/tr_$v gpurun_out/$TAG/k_$v.txt "$v" > /dev/null 2>&1
  echo "== $v"; grep "$FILT" gpurun_out/$TAG/k_$v.txt | cut -c1-70,100-150
done
