"""Is the row kernel's speed mode uniform over the output allocation?  Per context: ms per 50k-row slice at five
places of the 250k-row tile's buffers (each slice writes its own part of the text and packed regions)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable
S, n = 20000, 250000
rs = np.random.RandomState(20261004)
tb = genome.synthetic_snp_table(n, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
for rep in range(6):
    with RowGenerator(device=0, seed=12345) as g:
        g.set_samples(SampleTable(flags)); g.set_snps(tb)
        g.ctx.reserve_output(g.ctx.text_bytes(0, n), n * g.words_per_row)
        whole = []
        for i in range(8):
            g.ctx.generate(0, n, 3); g.ctx.sync(); whole.append(g.ctx.last_generate_ms())
        out = []
        for b in range(0, n, 50000):
            ms = []
            for i in range(8):
                g.ctx.generate(b, 50000, 3); g.ctx.sync(); ms.append(g.ctx.last_generate_ms())
            out.append(np.median(ms[2:]))
        print("context %d: whole tile %.3f ms; 50k-row slices (all written at the start of the buffers): %s" % (rep, np.median(whole[2:]), " ".join("%.3f" % v for v in out)), flush=True)
