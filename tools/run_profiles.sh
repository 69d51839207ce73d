#!/bin/bash
# The round's measurement artefacts for the row kernel, all on one GPU box: default bench line, kernel-trace stats of the
# same command, the two PMC traffic passes (-> profiles/pmc_traffic.json), and the spread over fresh processes.
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench: $(python -c "import json;d=json.load(open('$O/bench.json'));print(d['value']/1e6,'M rows/s frac',d['roofline']['frac'],'kernel ms',d['roofline']['kernel_ms_mean'])")"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 2 > $O/profiled_bench.json 2> $O/kt.err
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
head -6 $O/kernel_stats.csv
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o pmc -f csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --bgzf-tiles 0 > $O/pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o pmc -f csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --bgzf-tiles 0 > $O/pmc_f.log 2>&1
python tools/pmc_traffic.py $O/pmc_w/pmc_counter_collection.csv $O/pmc_f/pmc_counter_collection.csv --out $O/pmc_traffic.json
for i in 1 2 3 4 5 6; do
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 > $O/spread$i.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/spread$i.json'));print('fresh process $i: kernel %.3f ms frac %.3f  %.2f M rows/s' % (d['roofline']['kernel_ms_mean'], d['roofline']['frac'], d['value']/1e6))"
done
