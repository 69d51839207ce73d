#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/spread
for i in $(seq 1 ${1:-8}); do
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 > gpurun_out/spread/s$i.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/spread/s$i.json'));print('fresh process $i: kernel %.3f ms frac %.3f  %.2f M rows/s' % (d['roofline']['kernel_ms_mean'], d['roofline']['frac'], d['value']/1e6))"
done
