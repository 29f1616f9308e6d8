#!/bin/bash
# FETCH_SIZE calibration of upsample_full_subpixel_kernel in its own access pattern (MI355X_MICROARCH.md, HBM section):
# the product build against a build whose tiles read only their own interior (-DPDS_UPS_NOHALO: exactly the 212.3 MB of
# the input tensor at config 2, same load instructions).  tools/build_variant_one.sh ups_nohalo upsample_estimator "-DPDS_UPS_NOHALO" first.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-ups_cal}
rm -rf $OUT; mkdir -p $OUT
for v in base ups_nohalo; do
  if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
  for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/p_$v -- python tools/run_tail.py 2 > $OUT/log_$v.txt 2>&1
  done
  echo "== $v" >> $OUT/calibration.txt
  python tools/pmc_summary.py $OUT/p_$v upsample_full >> $OUT/calibration.txt 2>&1
  rm -rf $OUT/p_$v
done
cat $OUT/calibration.txt
