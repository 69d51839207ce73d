#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of the library (device-only compile of dnaf_api.hip; extra hipcc flags as arguments)
# usage: tools/kernel_resources.sh [-DNAME=VALUE ...] | grep gtz_encode
set -e
cd "$(dirname "$0")/../dna-factory_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c dnaf_api.hip -o /tmp/dnaf_dev.o "$@"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/dnaf_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/dnaf_dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/dnaf_dev.co | awk '
/\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2}
/\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {print name, "vgpr", vg, "sgpr", sg, "lds", lds, "scratch", scr, "spills", $2}'
