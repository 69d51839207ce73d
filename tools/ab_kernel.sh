#!/bin/bash
# A/B of the row kernel between the working tree and a git revision, by hardware counters (instruction count and
# shader cycles per launch are far less noisy than wall time, and the placement search of dnaf_reserve_output keeps
# both runs in the fast mode).  Run from the repository root in the build container:
#     tools/ab_kernel.sh [REV]        (default REV = HEAD)
set -e
REV=${1:-HEAD}
python dna-factory_amd/build.py > /dev/null
cp dna-factory_amd/csrc/libdnaf_hip.so tools/ab_new.so
DIRTY=0; git diff --quiet -- dna-factory_amd/csrc || DIRTY=1
[ $DIRTY = 1 ] && git stash -q -- dna-factory_amd/csrc
git checkout -q "$REV" -- dna-factory_amd/csrc
python dna-factory_amd/build.py > /dev/null
cp dna-factory_amd/csrc/libdnaf_hip.so tools/ab_base.so
git checkout -q HEAD -- dna-factory_amd/csrc
[ $DIRTY = 1 ] && git stash pop -q
python dna-factory_amd/build.py > /dev/null
/usr/local/graft/bin/gpurun --timeout 900 -- 'cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT; for v in base new base new; do cp $R/tools/ab_$v.so $R/dna-factory_amd/csrc/libdnaf_hip.so; rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/ab_$v.$RANDOM -o v -- python3 $R/tools/bgzf_experiments/kernel_jitter.py 10 > /dev/null 2>&1; done' > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
for d0 in sorted(glob.glob("gpurun_out/ab_base.*") + glob.glob("gpurun_out/ab_new.*")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(d0 + "/v_counter_collection.csv")):
        if "rowgen" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d0.split("/")[-1].split(".")[0], {k: round(sum(x[-4:]) / 4 / 1e6, 1) for k, x in d.items()})
PY
rm -rf gpurun_out/ab_base.* gpurun_out/ab_new.* tools/ab_base.so tools/ab_new.so
