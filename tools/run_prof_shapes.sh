#!/bin/bash
# vector-instruction counts of the row kernels at other shapes; usage: tools/run_prof_shapes.sh [S ...]  (default 200 2000)
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profshapes
mkdir -p $O
for S in ${@:-200 2000}; do
  T=$(( 1000000000 / (S * 2) )); [ $T -gt 5000000 ] && T=5000000
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVES -d $O/S$S -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $S --tile-rows $T --table-rows 5000000 --steps 3 --warmup 2 > $O/S$S.log 2>&1
  python - $S $T <<'PY'
import csv, collections, sys
S=int(sys.argv[1]); T=int(sys.argv[2])
rows=collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/profshapes/S%d/pmc_counter_collection.csv" % S)):
    if 'row' not in r['Kernel_Name']: continue
    d=int(r['Dispatch_Id']); rows.setdefault(d,{'k':r['Kernel_Name'][:40]})[r['Counter_Name']]=float(r['Counter_Value'])
    rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for d,v in list(rows.items())[-2:]:
    cyc=v['GRBM_GUI_ACTIVE']/8
    print("S=%d" % S, v['k'], "dur %.3f ms clock %.2f GHz VALU %.4g = %.1f per row, busy %.1f%%, SALU %.1f per row, LDS %.1f per row, waves %d" % (v['dur'], cyc/v['dur']/1e6, v['SQ_INSTS_VALU'], v['SQ_INSTS_VALU']/T, 100*v['SQ_ACTIVE_INST_VALU']*4/(cyc*1024), v['SQ_INSTS_SALU']/T, v['SQ_INSTS_LDS']/T, v['SQ_WAVES']))
PY
done
