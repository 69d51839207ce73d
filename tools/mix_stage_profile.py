"""Where a wave of rowmix_kernel spends its life: a build with -DDNAF_MIX_PROFILE (build_ab/libdnaf_mixprof.so) stamps
s_memtime at the kernel's stage boundaries; this prints the mean cycles per wave and stage at S samples.
usage (GPU box): DNAF_LIB=$PWD/build_ab/libdnaf_mixprof.so python tools/mix_stage_profile.py [S]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from dna_factory_amd import capi
from dna_factory_amd.genome import synthetic_snp_table
from dna_factory_amd.rowgen import RowGenerator

S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 5000000
rs = np.random.RandomState(20261004)
tb = synthetic_snp_table(n, 0.01, rs)
st = bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(20261004))
names = ["cell table + barrier", "records -> row table", "fixed columns -> LDS", "the rows (trips)", "stores drained"]
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(st)
    g.set_snps(tb)
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(3):
        g.ctx.generate(0, n, 3)
    g.ctx.sync()
    capi.lib.dnaf_debug_mix_profile(out)
    g.ctx.generate(0, n, 3)
    g.ctx.sync()
    capi.lib.dnaf_debug_mix_profile(out)
    waves = float(out[8])
    print("S=%d: %d waves; s_memtime ticks (100 MHz) per wave and stage:" % (S, waves))
    for k, name in enumerate(names):
        print("  %-26s %9.1f ticks = %6.2f us" % (name, out[k] / waves, out[k] / waves / 100.0))
