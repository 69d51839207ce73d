#!/bin/bash
# Every measurement script of the repo in one place; runs on a GPU box from the repo root (gpurun).  Output goes to
# gpurun_out/<what>/; what is to be judged is copied to profiles/ by hand (profiles/README.md says from which command).
#   tools/measure.sh profiles [tag]       round artefacts of the row kernel: bench line, kernel-trace stats, PMC traffic, spread
#   tools/measure.sh counters PATTERN [bench args]   SQ counters of the kernels whose name contains PATTERN
#   tools/measure.sh shapes               the bench at other sample counts (100 .. 50,000)
#   tools/measure.sh shapes-kt            kernel-trace stats at BASELINE config 1's and 2's sample counts
#   tools/measure.sh gtz [tag]            the on-GPU BGZF encoder: kernel-trace stats and SQ counters of the bench's BGZF leg
#   tools/measure.sh mt                   numpy-compatible MT19937 mode: bench line, kernel stats, counters
#   tools/measure.sh ab [rounds] [args]   A/B of build_ab/libdnaf_*.so on the row kernel (bench.py), interleaved
#   tools/measure.sh gtz-ab               A/B of build_ab/libdnaf_gtz_*.so on the encoder (per MAF class and bench leg)
#   tools/measure.sh tile19               A/B of build_ab/libdnaf_*.so on the bench table's 20th tile (X / Y rows)
#   tools/measure.sh concat [tag]         config 3 through the CLI with one and with eight contexts; BGZF structure of both files
#   tools/measure.sh tiles                CLI wall time by tile size and number of contexts
# Variants are built with tools/gtz_variant.sh NAME [-Dflags] (or any hipcc line writing build_ab/libdnaf_NAME.so).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
what=${1:-help}; shift || true

summarise() { # csv pattern: one line per dispatch of the matching kernels with every counter, clock and VALU share
python - "$1" "$2" <<'PY'
import csv, collections, sys
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']: continue
    d = int(r['Dispatch_Id'])
    rows.setdefault(d, {'k': r['Kernel_Name'][:56]})[r['Counter_Name']] = float(r['Counter_Value'])
    rows[d]['dur'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for d, v in rows.items():
    if v['dur'] < 0.05: continue
    print(d, v['k'], "dur %.3f ms" % v['dur'], " ".join("%s=%.4g" % (k, x) for k, x in v.items() if k not in ('k', 'dur')))
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_ACTIVE_INST_VALU' in v:
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        print("    clock %.2f GHz, VALU busy %.1f %% of SIMD cycles" % (cyc / v['dur'] / 1e6, 100 * v['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024)))
PY
}
line() { python -c "
import json,sys
d=json.load(open('$1')); r=d['roofline']
print('$2 kernel %.3f ms  %.0f GB/s  frac %.3f  %.2f M rows/s  (%s)' % (r['kernel_ms_mean'], r['achieved'], r['frac'], d['value']/1e6, r['kernel']))"; }

case $what in
profiles)
  O=gpurun_out/${1:-r03}; mkdir -p $O
  python bench.py > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
  line $O/bench.json "bench:"
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 2 > $O/profiled_bench.json 2> $O/kt.err
  cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o pmc -f csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --bgzf-tiles 0 > $O/pmc_w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o pmc -f csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --bgzf-tiles 0 > $O/pmc_f.log 2>&1
  python tools/pmc_traffic.py $O/pmc_w/pmc_counter_collection.csv $O/pmc_f/pmc_counter_collection.csv --out $O/pmc_traffic.json
  for i in 1 2 3 4 5 6; do python bench.py --cpu-seconds 0 --bgzf-tiles 0 > $O/spread$i.json 2>/dev/null && line $O/spread$i.json "fresh process $i:"; done ;;
counters)
  pat=$1; shift; O=gpurun_out/counters; mkdir -p $O
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVES -d $O/a -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --steps 3 --warmup 2 "$@" > $O/a.log 2>&1
  summarise $O/a/pmc_counter_collection.csv "$pat"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA -d $O/b -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --steps 3 --warmup 2 "$@" > $O/b.log 2>&1 || true
  summarise $O/b/pmc_counter_collection.csv "$pat" ;;
shapes)
  O=gpurun_out/shapes; mkdir -p $O
  for st in "100 5000000 5000000" "200 5000000 5000000" "400 5000000 5000000" "1000 2500000 5000000" "2000 2500000 5000000" "4000 1250000 5000000" "8000 600000 3000000" "50000 100000 1000000"; do
    set -- $st
    python bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $1 --tile-rows $2 --table-rows $3 --steps 10 --warmup 4 > $O/S$1.json 2> $O/S$1.err && line $O/S$1.json "S=$1 tile $2 rows:" || tail -3 $O/S$1.err
  done ;;
shapes-kt)
  O=gpurun_out/shapes_kt; mkdir -p $O
  for S in 200 2000; do
    T=$([ $S = 200 ] && echo 5000000 || echo 2500000)
    rocprofv3 --kernel-trace --stats -d $O/S$S -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 0 --samples $S --tile-rows $T --table-rows 5000000 --steps 10 --warmup 4 > $O/S$S.json 2> $O/S$S.err
    head -3 $O/S$S/kt_kernel_stats.csv
  done ;;
gtz)
  O=gpurun_out/gtz_${1:-r03}; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 3 --steps 2 --warmup 2 > $O/bench.json 2> $O/kt.err
  python -c "import json;print(json.load(open('$O/bench.json'))['bgzf'])"; head -12 $O/kt/kt_kernel_stats.csv
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pmc -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 1 --steps 2 --warmup 1 > $O/pmc.log 2>&1
  summarise $O/pmc/pmc_counter_collection.csv gtz_encode
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES -d $O/pmc2 -o pmc -f csv -- python3 bench.py --cpu-seconds 0 --bgzf-tiles 1 --steps 2 --warmup 1 > $O/pmc2.log 2>&1 || true
  summarise $O/pmc2/pmc_counter_collection.csv gtz_encode ;;
mt)
  O=gpurun_out/mt; mkdir -p $O
  python bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err && line $O/bench.json "mt19937:"
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python3 bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 6 --warmup 2 > $O/kt.log 2>&1
  head -4 $O/kt/kt_kernel_stats.csv
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc -o pmc -f csv -- python3 bench.py --rng mt19937 --cpu-seconds 0 --bgzf-tiles 0 --steps 2 --warmup 1 > $O/pmc.log 2>&1
  summarise $O/pmc/pmc_counter_collection.csv mt_ ;;
ab)
  R=${1:-2}; shift || true; O=gpurun_out/ab; mkdir -p $O
  for r in $(seq 1 $R); do for lib in build_ab/libdnaf_*.so; do n=$(basename $lib .so)
    DNAF_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 20 "$@" > $O/$n.$r.json 2> $O/$n.$r.err && line $O/$n.$r.json "$n round $r:" || { echo "$n failed"; tail -3 $O/$n.$r.err; }
  done; done ;;
gtz-ab)
  O=gpurun_out/gtzab; mkdir -p $O
  for lib in build_ab/libdnaf_gtz_*.so; do n=$(basename $lib .so)
    DNAF_LIB=$PWD/$lib python tools/bgzf_experiments/gtz_bymaf.py > $O/$n.bymaf 2>&1 || { echo "$n failed"; tail -3 $O/$n.bymaf; continue; }
    DNAF_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --steps 2 --warmup 2 --bgzf-tiles 3 > $O/$n.json 2> $O/$n.err
    echo "== $n: $(tail -1 $O/$n.bymaf)"
    python -c "
import json
d=json.load(open('$O/$n.json'))['bgzf']
print('   bench: %.2f ms per tile, %.4f B/genotype, ratio %.2f' % (d['kernel_ms_per_tile'], d['bytes_per_genotype'], d['ratio']))
print('   by class ms:', ' '.join(l.split()[-3] for l in open('$O/$n.bymaf') if l.startswith('p=')))"
  done ;;
tile19)
  for r in 1 2 3 4; do for lib in build_ab/libdnaf_*.so; do
    DNAF_LIB=$PWD/$lib python bench.py --cpu-seconds 0 --bgzf-tiles 0 --steps 1 --warmup 19 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$(basename $lib .so) round $r: tile 19 kernel %.3f ms' % d['roofline']['kernel_ms_mean'])"
  done; done ;;
concat)
  tag=${1:-r03}; O=/tmp/dnaf_concat; rm -rf $O; mkdir -p $O gpurun_out; df -h /tmp | tail -1
  for dev in 0 0,0,0,0,0,0,0,0; do
    n=$(echo $dev | tr -cd , | wc -c); n=$((n+1))
    ( time python pop_factory.py -s 10000 -c 10000 -x 5000000 -f 0.01 --seed 7 --devices $dev --outdir $O/d$n ) > gpurun_out/concat_${tag}_$n.log 2>&1
    rc=$?; echo "== $n context(s): cli rc=$rc"; grep -E "Finished|real|Elapsed|Time to write" gpurun_out/concat_${tag}_$n.log
    [ $rc -eq 0 ] || { grep -v Output gpurun_out/concat_${tag}_$n.log | tail -5; exit $rc; }
    python - <<PY
import struct, os
p = "$O/d$n/population.vcf.gz"; size = os.path.getsize(p); f = open(p, "rb"); pos = n = isize = 0
while pos < size:
    f.seek(pos); h = f.read(18)
    assert h[:4] == b"\x1f\x8b\x08\x04" and h[12:14] == b"BC", pos
    b = struct.unpack("<H", h[16:18])[0] + 1
    f.seek(pos + b - 4); isize += struct.unpack("<I", f.read(4))[0]; pos += b; n += 1
print("   %d bytes, %d BGZF members, %d uncompressed bytes" % (size, n, isize))
PY
  done; rm -rf $O ;;
tiles)
  O=/tmp/dnaf_sweep; rm -rf $O; mkdir -p $O gpurun_out
  for dev in 0 0,0,0,0; do for tr in 0 26000 100000; do
    n=$(echo $dev | tr -cd , | wc -c); n=$((n+1))
    python pop_factory.py -s 10000 -c 10000 -x 5000000 -f 0.01 --seed 7 --devices $dev --tile-rows $tr --outdir $O/x > gpurun_out/tiles.log 2>&1 || { tail -3 gpurun_out/tiles.log; exit 1; }
    echo "contexts $n tile-rows $tr: $(grep -E 'Finished VCF' gpurun_out/tiles.log)"; rm -rf $O/x
  done; done; rm -rf $O ;;
*) sed -n 2,16p "$0" ;;
esac
