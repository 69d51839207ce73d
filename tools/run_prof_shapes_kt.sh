#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench at BASELINE config 1's and 2's sample counts (short-row kernel)
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profshapes_kt
mkdir -p $O
for S in 200 2000; do
  T=$([ $S = 200 ] && echo 5000000 || echo 2500000)
  rocprofv3 --kernel-trace --stats -d $O/S$S -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $S --tile-rows $T --table-rows 5000000 --steps 10 --warmup 4 > $O/S$S.json 2> $O/S$S.err
  head -3 $O/S$S/kt_kernel_stats.csv
done
