#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/shapes
for S in 100 400 600 800; do
for NM in 0 1; do
  DNAF_MIX_MAX_GROUPS=100000 DNAF_NO_ROWMIX=$NM python bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $S --tile-rows 2500000 --table-rows 2500000 --steps 10 --warmup 4 > gpurun_out/shapes/x.json 2> gpurun_out/shapes/x.err || { tail -3 gpurun_out/shapes/x.err; continue; }
  python -c "import json;d=json.load(open('gpurun_out/shapes/x.json'));r=d['roofline'];print('S=$S no_rowmix=$NM kernel %.3f ms  frac %.3f' % (r['kernel_ms_mean'], r['frac']))"
done; done
