#!/bin/bash
# workgroups per CU of the row kernel, capped through unused dynamic LDS: kernel time by cap, several fresh processes each
# usage: run_pad_sweep.sh "pad1 pad2 ..." [rounds]
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pad
for r in $(seq 1 ${2:-3}); do
for lib in build_ab/libdnaf_*.so; do
for pad in $1; do
  n=$(basename $lib .so)
  DNAF_LIB=$PWD/$lib DNAF_RUN_LDS_PAD=$pad python bench.py --cpu-seconds 0 --bgzf-tiles 0 > gpurun_out/pad/$n.$pad.$r.json 2>/dev/null || { echo "$n pad $pad failed"; continue; }
  python -c "import json;d=json.load(open('gpurun_out/pad/$n.$pad.$r.json'));print('$n pad %5d round $r: kernel %.3f ms frac %.3f' % ($pad, d['roofline']['kernel_ms_mean'], d['roofline']['frac']))"
done; done; done
