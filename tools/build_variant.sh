#!/bin/bash
# A/B builds of libpds_hip.so with extra compiler flags: tools/build_variant.sh NAME "-DPDS_TW=48 ..."
# -> build/variants/libpds_NAME.so (select with PDS_HIP_LIB=... ; build/ is git-ignored but travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w $2 -o build/variants/libpds_$1.so practicaldeepstereo_nips2018_amd/csrc/*.hip
echo built build/variants/libpds_$1.so
