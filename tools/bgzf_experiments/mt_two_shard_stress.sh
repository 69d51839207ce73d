#!/bin/bash
# the MT19937 two-shard CLI path N times against the one-device output (bodies must be identical every time)
N=${1:-20}
python pop_factory.py -s 30 -c 22 -x 700 -f 0.01 --seed 86399 --rng mt19937 --outdir /tmp/mt_one > /dev/null 2>&1
want=$(python -c "import gzip,hashlib;print(hashlib.md5(gzip.open('/tmp/mt_one/population.vcf.gz','rb').read().split(b'#CHROM',1)[1]).hexdigest())")
bad=0
for i in $(seq 1 $N); do
  python pop_factory.py -s 30 -c 22 -x 700 -f 0.01 --seed 86399 --rng mt19937 --tile-rows 90 --devices 0,0 --outdir /tmp/mt_two > /dev/null 2>&1
  got=$(python -c "import gzip,hashlib;print(hashlib.md5(gzip.open('/tmp/mt_two/population.vcf.gz','rb').read().split(b'#CHROM',1)[1]).hexdigest())")
  if [ "$got" != "$want" ]; then bad=$((bad+1)); echo "run $i differs"; fi
done
echo "$N runs, $bad differ"
rm -rf /tmp/mt_one /tmp/mt_two
