#!/bin/bash
# bench_conv64 under library variants: tools/time_x3_variants.sh v1 v2 ...
for v in base "$@"; do
  if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
  echo "== $v: $(python tools/bench_conv64.py 20 2>&1 | tail -1)"
done
