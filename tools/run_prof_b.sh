#!/bin/bash
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profb
mkdir -p $O
export DNAF_LIB=$PWD/build_ab/libdnaf_b_asm_nobounds.so
for W in text+packed text packed; do
  python bench.py --cpu-seconds 0 --bgzf-tiles 0 --what $W --steps 20 > $O/bench_$W.json 2>$O/bench_$W.err
  python - <<PY
import json
d=json.load(open("$O/bench_$W.json"))
print("$W", "ms/launch %.3f frac %.3f rows/s %.3e" % (d["roofline"]["kernel_ms_mean"], d["roofline"]["frac"], d["value"]))
PY
done
rocprofv3 --kernel-trace -d $O/kt -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 40 --warmup 2 > $O/kt.log 2>&1
python - <<'PY'
import csv
d=[(int(r['Start_Timestamp']),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in csv.DictReader(open("gpurun_out/profb/kt/kt_kernel_trace.csv")) if 'rowrun' in r['Kernel_Name']]
d.sort()
print("per-launch ms:", " ".join("%.2f"%x[1] for x in d))
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
python - <<'PY'
import csv, collections
rows=collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/profb/pmc_sq/pmc_counter_collection.csv")):
    if 'rowrun' not in r['Kernel_Name']: continue
    d=int(r['Dispatch_Id']); rows.setdefault(d,{})[r['Counter_Name']]=float(r['Counter_Value'])
    rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for d,v in rows.items():
    cyc=v['GRBM_GUI_ACTIVE']/8
    print(d, "dur %.3f ms clock %.2f GHz VALU insts %.4g  VALU busy %.1f%%  wave-cycles %.4g wait_any %.3g wait_inst_any %.3g" % (v['dur'], cyc/v['dur']/1e6, v['SQ_INSTS_VALU'], 100*v['SQ_ACTIVE_INST_VALU']*4/(cyc*1024), v['SQ_WAVE_CYCLES'], v['SQ_WAIT_ANY'], v['SQ_WAIT_INST_ANY']))
PY
