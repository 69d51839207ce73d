#!/bin/bash
# A/B of library builds on one GPU box: every build_ab/libdnaf_*.so through bench.py (DNAF_LIB), interleaved, ROUNDS times.
# usage: tools/ab.sh [ROUNDS] [extra bench args]
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
R=${1:-2}; shift || true
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for lib in build_ab/libdnaf_*.so; do
    n=$(basename $lib .so)
    DNAF_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 20 "$@" > gpurun_out/ab/$n.$r.json 2> gpurun_out/ab/$n.$r.err || { echo "$n failed"; tail -3 gpurun_out/ab/$n.$r.err; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/$n.$r.json"))
print("%-14s round $r  kernel %.3f ms  frac %.3f  %.2f M rows/s" % ("$n", d["roofline"]["kernel_ms_mean"], d["roofline"]["frac"], d["value"]/1e6))
PY
  done
done
