#!/bin/bash
# the short-row kernel by rows per wave (DNAF_ROWS_PER_WAVE; 0 = the library's own choice); usage: tools/rpw_sweep.sh "S ..." "R ..."
cd "$GRAFT_REPO_ROOT"
for S in ${1:-2000 200}; do for R in ${2:-0 16 32 64}; do
  T=$(( 5000000000 / S )); [ $T -gt 5000000 ] && T=5000000
  if [ $R = 0 ]; then unset DNAF_ROWS_PER_WAVE; else export DNAF_ROWS_PER_WAVE=$R; fi
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $S --tile-rows $T --table-rows 5000000 --steps 10 --warmup 4 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('S=$S rows/wave $R: kernel %.3f ms frac %.3f' % (r['kernel_ms_mean'], r['frac']))"
done; done
