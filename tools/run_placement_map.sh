#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/place
mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 200 ./tools/placement_study align --k 1 --reps 3 > $O/own$i.log 2>&1
  (grep "^OWN" $O/own$i.log || true)
  echo "--- run $i done"
done
