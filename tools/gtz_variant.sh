#!/bin/bash
# build_ab/libdnaf_gtz_NAME.so = the library compiled with extra flags (encoder experiments; tools/measure.sh gtz-ab runs them)
set -e
n=$1; shift
cd "$(dirname "$0")/../dna-factory_amd/csrc"
mkdir -p ../../build_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../../build_ab/libdnaf_gtz_$n.so dnaf_api.hip "$@"
