#!/bin/bash
# pipelined / sequential bench values of library variants, interleaved ROUNDS times: tools/ab_pipe.sh ROUNDS v1 v2 ...
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset PDS_HIP_LIB; else export PDS_HIP_LIB=$PWD/build/variants/libpds_$v.so; fi
    python bench.py --no-cpu-baseline --no-train-record --windows 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'pairs/s', round(d['value'],1), 'sequential ms', round(d['ms_per_frame'],3))"
  done
done
