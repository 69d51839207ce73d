#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profmt
mkdir -p $O
python bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 6 --warmup 2 > $O/bench.json 2>$O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print('mt19937: %.2f M rows/s, kernel(s) %.2f ms per tile of %d rows' % (d['value']/1e6, d['roofline']['kernel_ms_mean'], d['config']['tile_rows']))"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 6 --warmup 2 > $O/kt.log 2>&1
head -4 $O/kt/kt_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc -o pmc -f csv -- python3 bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 2 --warmup 1 > $O/pmc.log 2>&1
python - <<'PY'
import csv, collections
rows=collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/profmt/pmc/pmc_counter_collection.csv")):
    if 'mt_' not in r['Kernel_Name']: continue
    d=int(r['Dispatch_Id']); rows.setdefault(d,{'k':r['Kernel_Name'][:40]})[r['Counter_Name']]=float(r['Counter_Value'])
    rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for d,v in list(rows.items())[-4:]:
    cyc=v['GRBM_GUI_ACTIVE']/8
    print(d, v['k'], "dur %.2f ms clock %.2f GHz VALU insts %.4g busy %.1f%%  LDS insts %.4g  LDS active %.3g bank conflict %.3g" % (v['dur'], cyc/v['dur']/1e6, v['SQ_INSTS_VALU'], 100*v['SQ_ACTIVE_INST_VALU']*4/(cyc*1024), v.get('SQ_INSTS_LDS',0), v.get('SQ_ACTIVE_INST_LDS',0), v.get('SQ_LDS_BANK_CONFLICT',0)))
PY
