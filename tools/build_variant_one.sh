#!/bin/bash
# A/B build that recompiles ONE source with extra flags and links it with the current objects of the others:
#   tools/build_variant_one.sh NAME conv3d_t8 "-DPDS_T8_NOMFMA"   -> build/variants/libpds_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants build/vobj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $3 -c practicaldeepstereo_nips2018_amd/csrc/$2.hip -o build/vobj/$1_$2.o
OBJS=$(ls build/obj/*.o | grep -v "/$2.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libpds_$1.so $OBJS build/vobj/$1_$2.o
echo built build/variants/libpds_$1.so
