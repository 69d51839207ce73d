#!/bin/bash
# wall time of config 3 through the CLI by tile size and number of contexts (one GPU)
cd "$GRAFT_REPO_ROOT"
O=/tmp/dnaf_sweep; rm -rf $O; mkdir -p $O gpurun_out
for dev in 0 0,0,0,0; do for tr in 0 26000 100000; do
  n=$(echo $dev | tr -cd , | wc -c); n=$((n+1))
  python pop_factory.py -s 10000 -c 10000 -x 5000000 -f 0.01 --seed 7 --devices $dev --tile-rows $tr --outdir $O/x > gpurun_out/r3_sweep.log 2>&1 || { tail -3 gpurun_out/r3_sweep.log; exit 1; }
  echo "contexts $n tile-rows $tr: $(grep -E 'Finished VCF' gpurun_out/r3_sweep.log)"
  rm -rf $O/x
done; done
DNAF_TRACE_TILES=1 python pop_factory.py -s 10000 -c 10000 -x 1000000 -f 0.01 --seed 7 --devices 0,0,0,0 --outdir $O/x 2>&1 | grep "^tile" | sed -n '40,52p'
rm -rf $O
