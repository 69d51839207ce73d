import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable, SnpTable
S, n = 20000, 1500
flags = (np.random.RandomState(1).rand(S) <= 0.5).astype(np.uint8)
for p in (0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.495):
    cum = np.ones((n, 4)); cum[:, 0] = 1 - p
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"1", dtype="S2"), np.arange(n) * 100 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"T", dtype="S1"), np.full(n, 2, np.uint8), cum)
    with RowGenerator(device=0, seed=7) as g:
        g.set_samples(SampleTable(flags)); g.set_snps(tb)
        g.ctx.generate(0, n, capi.OUT_TEXT | capi.OUT_PACKED)
        a = g.ctx.bgzf_compress(text_domain=True)
        b = g.ctx.bgzf_compress()
        print("p=%.3f  bits/cell: text-domain %.3f  genotype-domain %.3f" % (p, a * 8 / (n * S), b * 8 / (n * S)), flush=True)
