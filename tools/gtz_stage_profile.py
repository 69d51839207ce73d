"""Where gtz_encode's time goes: a build with -DDNAF_GTZ_PROFILE (build_ab/libdnaf_gtzprof.so) stamps s_memtime at the
kernel's stage boundaries (thread 0 of every workgroup); this prints the share of each stage on the bench tile.
usage (GPU box): DNAF_LIB=$PWD/build_ab/libdnaf_gtzprof.so python tools/gtz_stage_profile.py"""
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

S, n = 20000, 250000
rs = np.random.RandomState(20261004)
tb = synthetic_snp_table(n, 0.01, rs)
st = bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(20261004))
names = ["rows located, layout", "cells + code tables to LDS", "key tables (4 quarters)", "thread 0: search + tokens + count + CRC",
         "wait for the slowest thread of pass 1", "scan", "thread 0: emission", "wait for the slowest emitter", "copy to the slot"]
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(st)
    g.set_snps(tb)
    g.ctx.generate(0, n, 3)
    g.ctx.bgzf_compress()
    out = (ctypes.c_ulonglong * 16)()
    capi.lib.dnaf_debug_gtz_profile(out)  # reset (the histogram pass is not stamped, the first encode is)
    g.ctx.generate(0, n, 3)
    nbytes = g.ctx.bgzf_compress(reuse_tables=True)
    ms = g.ctx.last_bgzf_ms()
    capi.lib.dnaf_debug_gtz_profile(out)
    tot = float(sum(out[:9]))
    print("encoder %.2f ms per tile, %d bytes; stage shares of the workgroups' time:" % (ms, nbytes))
    for k, name in enumerate(names):
        print("  %-44s %5.1f %%" % (name, 100.0 * out[k] / tot))
