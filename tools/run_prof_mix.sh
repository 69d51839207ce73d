#!/bin/bash
# wait / issue counters of rowmix_kernel at S = 200
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/profmix
mkdir -p $O
ARGS="bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples ${1:-200} --tile-rows ${2:-5000000} --table-rows 5000000 --steps 3 --warmup 2"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $O/a -o pmc -f csv -- python3 $ARGS > $O/a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT -d $O/b -o pmc -f csv -- python3 $ARGS > $O/b.log 2>&1 || true
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum -d $O/c -o pmc -f csv -- python3 $ARGS > $O/c.log 2>&1 || true
python - <<'PY'
import csv, collections
for d_ in "abc":
    try: f=open("gpurun_out/profmix/%s/pmc_counter_collection.csv" % d_)
    except OSError: print(d_, "failed"); continue
    rows=collections.OrderedDict()
    for r in csv.DictReader(f):
        if 'row' not in r['Kernel_Name']: continue
        d=int(r['Dispatch_Id']); rows.setdefault(d,{'k':r['Kernel_Name'][:30]})[r['Counter_Name']]=float(r['Counter_Value'])
        rows[d]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    for d,v in list(rows.items())[-1:]:
        print(v['k'], "dur %.3f ms" % v['dur'], " ".join("%s=%.4g" % (k,x) for k,x in v.items() if k not in ('k','dur')))
PY
