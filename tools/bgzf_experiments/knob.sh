#!/bin/bash
# usage: knob.sh NAME VALUE  -> rebuild with constexpr NAME = VALUE and report bits/cell by MAF
set -e
cd /root/repo
cp dna-factory_amd/csrc/gtz_kernels.hip.h /tmp/gtz_kernels.bak
python - "$1" "$2" <<'PY'
import re,sys
p="/root/repo/dna-factory_amd/csrc/gtz_kernels.hip.h"
s=open(p).read()
name,val=sys.argv[1],sys.argv[2]
s2=re.sub(r"(constexpr uint32_t %s = )[^;]+;"%name, r"\g<1>%s;"%val, s, count=1)
assert s2!=s or ("= %s;"%val) in s
open(p,"w").write(s2)
PY
python dna-factory_amd/build.py 2>&1 | grep -E "error" -A3 | head -5
