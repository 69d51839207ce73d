#!/bin/bash
# SQ counters of the BGZF encoder kernels (bench.py's BGZF leg): where gtz_encode's time goes.
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profgtz
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 2 --steps 2 --warmup 1 > $O/kt.log 2>&1
head -8 $O/kt/kt_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pmc -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 1 --steps 2 --warmup 1 > $O/pmc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES -d $O/pmc2 -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 1 --steps 2 --warmup 1 > $O/pmc2.log 2>&1 || true
python - <<'PY'
import csv, collections
for d_ in ("pmc","pmc2"):
    try: f=open("gpurun_out/profgtz/%s/pmc_counter_collection.csv" % d_)
    except OSError: continue
    rows=collections.OrderedDict()
    for r in csv.DictReader(f):
        if 'gtz_encode' not in r['Kernel_Name'] and 'bgzf_deflate' not in r['Kernel_Name']: continue
        d=int(r['Dispatch_Id']); rows.setdefault(d,{'k':r['Kernel_Name'][:48]})[r['Counter_Name']]=float(r['Counter_Value'])
        rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    for d,v in rows.items():
        if v['dur'] < 1: continue
        print(d, v['k'], "dur %.2f ms" % v['dur'], " ".join("%s=%.4g" % (k,x) for k,x in v.items() if k not in ('k','dur')))
        if 'GRBM_GUI_ACTIVE' in v:
            cyc=v['GRBM_GUI_ACTIVE']/8
            print("    clock %.2f GHz, VALU busy %.1f %% of SIMD cycles" % (cyc/v['dur']/1e6, 100*v['SQ_ACTIVE_INST_VALU']*4/(cyc*1024)))
PY
