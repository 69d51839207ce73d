#!/bin/bash
# A/B of encoder builds on one GPU box: every build_ab/libdnaf_gtz_*.so through tools/bgzf_experiments/gtz_bymaf.py (time and
# bits per class) and bench.py's BGZF leg.   Build variants with tools/gtz_variant.sh NAME [-Dflags...]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gtzab
for lib in build_ab/libdnaf_gtz_*.so; do
  n=$(basename $lib .so)
  DNAF_LIB=$PWD/$lib python tools/bgzf_experiments/gtz_bymaf.py > gpurun_out/gtzab/$n.bymaf 2>&1 || { echo "$n failed"; tail -3 gpurun_out/gtzab/$n.bymaf; continue; }
  DNAF_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --steps 2 --warmup 2 --bgzf-tiles 3 > gpurun_out/gtzab/$n.json 2> gpurun_out/gtzab/$n.err
  echo "== $n: $(tail -1 gpurun_out/gtzab/$n.bymaf)"
  python -c "
import json,sys
d=json.load(open('gpurun_out/gtzab/$n.json'))['bgzf']
print('   bench: %.2f ms per tile, %.4f B/genotype, ratio %.2f' % (d['kernel_ms_per_tile'], d['bytes_per_genotype'], d['ratio']))
print('   by class ms:', ' '.join(l.split()[-3] for l in open('gpurun_out/gtzab/$n.bymaf') if l.startswith('p=')))"
done
