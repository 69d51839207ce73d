"""Which buffer's placement sets the row kernel's speed mode?  One context; the packed buffer alone, then the text
buffer alone, is re-allocated (by reserving a little more each time) and the same tile is timed after every step."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable
os.environ["DNAF_NO_PLACEMENT_PROBE"] = "1"
S, n = 20000, 250000
rs = np.random.RandomState(20261004)
tb = genome.synthetic_snp_table(n, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
def timed(g):
    ms = []
    for i in range(7):
        g.ctx.generate(0, n, 3); g.ctx.sync(); ms.append(g.ctx.last_generate_ms())
    return float(np.median(ms[3:]))
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(SampleTable(flags)); g.set_snps(tb)
    T, P = g.ctx.text_bytes(0, n), n * g.words_per_row
    g.ctx.reserve_output(T, P)
    print("start            %.3f ms" % timed(g), flush=True)
    for step in range(1, 9):
        P += 1 << 20
        g.ctx.reserve_output(T, P)
        print("packed re-alloc  %.3f ms" % timed(g), flush=True)
    for step in range(1, 9):
        T += 64 << 20
        g.ctx.reserve_output(T, P)
        print("text re-alloc    %.3f ms" % timed(g), flush=True)
