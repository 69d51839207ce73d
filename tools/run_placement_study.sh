#!/bin/bash
# Runs on the GPU box (gpurun): instruction rates, the placement study un-profiled, then three --pmc passes of its
# lottery part (per-dispatch TCC counters to join with the printed times).  Outputs under gpurun_out/place/.
set -e -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/place
mkdir -p $O
./tools/valu_rates > $O/valu_rates.log 2>&1
echo "valu_rates done"
timeout -k 10 300 ./tools/placement_study all --k 8 --reps 3 > $O/study_all.log 2>&1
echo "study all done"
timeout -k 10 120 ./tools/placement_study textarena --reps 3 > $O/study_textarena.log 2>&1
echo "textarena done"
P=1
for C in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_BUSY" \
         "TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_EA0_WRREQ_LEVEL" \
         "TCC_HIT TCC_MISS TCC_WRITEBACK TCC_EA0_RDREQ"; do
  timeout -k 10 300 rocprofv3 --pmc $C -d $O/pmc$P -o pmc -f csv json -- ./tools/placement_study lottery --k 8 --reps 2 > $O/pmc$P.log 2>&1
  echo "pmc pass $P done"
  P=$((P+1))
done
ls -la $O $O/pmc1 | head -40
