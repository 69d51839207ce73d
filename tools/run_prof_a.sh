#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profa
mkdir -p $O
for W in text+packed text packed; do
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 --what $W --steps 10 > $O/bench_$W.json 2>$O/bench_$W.err
  python - <<PY
import json
d=json.load(open("$O/bench_$W.json"))
print("$W", "ms/launch %.3f frac %.3f rows/s %.3e" % (d["roofline"]["kernel_ms_mean"], d["roofline"]["frac"], d["value"]))
PY
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
python - <<'PY'
import csv, collections
rows=collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/profa/pmc_sq/pmc_counter_collection.csv")):
    if 'rowrun' not in r['Kernel_Name'] and 'rowgen' not in r['Kernel_Name']: continue
    d=int(r['Dispatch_Id']); rows.setdefault(d,{})[r['Counter_Name']]=float(r['Counter_Value'])
    rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for d,v in rows.items(): print(d, {k:(round(x,3) if k=='dur' else int(x)) for k,x in v.items()})
PY
