"""dnaf_read_packed at the bench shape (S = 20,000, 250,000 rows, padded pitch): seconds for the dense [n][W] matrix on the host.
usage (GPU box): python tools/read_packed_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dna_factory_amd import capi
from dna_factory_amd.genome import synthetic_snp_table
from dna_factory_amd.rowgen import RowGenerator
S, n = 20000, 250000
tb = synthetic_snp_table(n, 0.01, np.random.RandomState(1))
st = bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(1))
with RowGenerator(device=0, seed=3) as g:
    g.set_samples(st); g.set_snps(tb)
    g.ctx.generate(0, n, capi.OUT_PACKED)
    g.ctx.sync()
    for rep in range(2):
        t0 = time.time(); m = g.ctx.read_packed(n, g.words_per_row); dt = time.time() - t0
        print("read_packed %s: %.3f s for %.2f GB (%s)" % ("2-D strided copy" if os.environ.get("DNAF_READ_PACKED_2D") else "compact + plain copy", dt, m.nbytes / 1e9, "first call" if rep == 0 else "second"), flush=True)
    print("checksum", int(m.sum(dtype=np.uint64)))
