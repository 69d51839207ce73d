#!/bin/bash
# config 3 through the CLI on one GPU: one context, then eight contexts (tiles dealt round-robin, one ordered writer):
# wall times, and the two files' BGZF structure side by side (members, sum of the members' uncompressed sizes)
cd "$GRAFT_REPO_ROOT"
O=/tmp/dnaf_concat; rm -rf $O; mkdir -p $O gpurun_out
df -h /tmp | tail -1
set -o pipefail
for dev in 0 0,0,0,0,0,0,0,0; do
  n=$(echo $dev | tr -cd , | wc -c); n=$((n+1))
  ( time python pop_factory.py -s 10000 -c 10000 -x 5000000 -f 0.01 --seed 7 --devices $dev --outdir $O/d$n ) > gpurun_out/r3_concat_$1_$n.log 2>&1
  rc=$?; echo "== $n context(s): cli rc=$rc"
  grep -E "shard files|Finished|real|Elapsed|Time to write" gpurun_out/r3_concat_$1_$n.log
  [ $rc -eq 0 ] || { grep -v Output gpurun_out/r3_concat_$1_$n.log | tail -5; exit $rc; }
  python - <<PY
import struct, os, sys
p = "$O/d$n/population.vcf.gz"
size = os.path.getsize(p)
f = open(p, "rb"); pos = 0; n = 0; isize = 0
while pos < size:
    f.seek(pos); h = f.read(18)
    assert h[:4] == b"\x1f\x8b\x08\x04" and h[12:14] == b"BC", pos
    b = struct.unpack("<H", h[16:18])[0] + 1
    f.seek(pos + b - 4); isize += struct.unpack("<I", f.read(4))[0]
    pos += b; n += 1
print("   %s: %d bytes, %d BGZF members, %d uncompressed bytes" % (os.path.basename(os.path.dirname(p)), size, n, isize))
PY
done
rm -rf $O
