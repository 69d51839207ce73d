"""Per-tile kernel time over the 20 tiles of the bench table in one context (placement chosen by reserve_output)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable
S, n, tiles = 20000, 250000, 20
rs = np.random.RandomState(20261004)
tb = genome.synthetic_snp_table(n * tiles, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(SampleTable(flags)); g.set_snps(tb)
    g.ctx.reserve_output(max(g.ctx.text_bytes(t * n, n) for t in range(tiles)), n * g.words_per_row)
    for rep in range(2):
        out = []
        for t in range(tiles):
            ms = []
            for i in range(4):
                g.ctx.generate(t * n, n, 3); g.ctx.sync(); ms.append(g.ctx.last_generate_ms())
            out.append(min(ms[1:]))
        print("pass %d: " % rep + " ".join("%.2f" % v for v in out) + "  mean %.3f" % np.mean(out), flush=True)
