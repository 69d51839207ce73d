#!/bin/bash
# the bench at other sample counts (same kernel family, other row shapes)
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/shapes
run() { # samples tile_rows table_rows
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $1 --tile-rows $2 --table-rows $3 --steps 10 --warmup 4 > gpurun_out/shapes/S$1.json 2> gpurun_out/shapes/S$1.err || { tail -3 gpurun_out/shapes/S$1.err; return; }
  python - <<PY
import json
d=json.load(open("gpurun_out/shapes/S$1.json"))
r=d["roofline"]
print("S=%-6d tile %8d rows  kernel %.3f ms  %.1f GB/s  frac %.3f  %.1f M rows/s" % ($1, $2, r["kernel_ms_mean"], r["achieved"], r["frac"], d["value"]/1e6))
PY
}
run 100 5000000 5000000
run 200 5000000 5000000
run 400 5000000 5000000
run 1000 2500000 5000000
run 2000 2500000 5000000
run 4000 1250000 5000000
run 8000 600000 3000000
run 50000 100000 1000000
