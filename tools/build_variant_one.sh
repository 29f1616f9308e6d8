#!/bin/bash
# A/B build that recompiles ONE source with extra flags and links it with the tree's other objects (build/obj):
#   tools/build_variant_one.sh NAME conv2d_x3 "-DPDS_X3_NOMFMA"  -> build/variants/libpds_NAME.so  (PDS_HIP_LIB=... selects it)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $3 -c practicaldeepstereo_nips2018_amd/csrc/$2.hip -o build/vobj/$2_$1.o
OBJS=$(ls build/obj/*.o | grep -v "/$2.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libpds_$1.so $OBJS build/vobj/$2_$1.o
echo built build/variants/libpds_$1.so
