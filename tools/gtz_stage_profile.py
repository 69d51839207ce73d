"""Where gtz_encode's time goes: a build with -DDNAF_GTZ_PROFILE (build_ab/libdnaf_gtzprof.so) stamps s_memtime at the
kernel's stage boundaries (thread 0 of every workgroup); this prints the share of each stage, on the bench tile or (with
MAF values as arguments) on rows of one minor-allele frequency each.
usage (GPU box): DNAF_LIB=$PWD/build_ab/libdnaf_gtzprof.so python tools/gtz_stage_profile.py [maf ...]"""
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
from dna_factory_amd.tables import SampleTable, SnpTable

S = 20000
NAMES = os.environ.get("GTZ_STAGE_NAMES", "rows located, layout|cells + code tables to LDS|key tables|thread 0: search + tokens + count + CRC|"
                       "wait for the slowest thread of pass 1|scan|thread 0: emission|wait for the slowest emitter|copy to the slot").split("|")


def run(tb, st, n, label):
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
        tot = float(sum(out[:len(NAMES)]))
        print("%s: encoder %.2f ms per 250k rows, %.4f B/genotype; stage shares of the workgroups' time:" % (label, ms * 250000 / n, nbytes / (n * S)))
        for k, name in enumerate(NAMES):
            print("  %-44s %5.1f %%  (%.2f ms of the tile)" % (name, 100.0 * out[k] / tot, ms * 250000 / n * out[k] / tot))


mafs = [float(x) for x in sys.argv[1:]]
if not mafs:
    n = 250000
    rs = np.random.RandomState(20261004)
    tb = synthetic_snp_table(n, 0.01, rs)
    run(tb, bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(20261004)), n, "bench tile")
for p in mafs:
    n = 12000
    cum = np.ones((n, 4))
    cum[:, 0] = 1 - p
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"1", dtype="S2"), np.arange(n) * 100 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"T", dtype="S1"), np.full(n, 2, np.uint8), cum)
    run(tb, SampleTable((np.random.RandomState(1).rand(S) <= 0.5).astype(np.uint8)), n, "MAF %.3f" % p)
