"""Does the row kernel's speed mode change when the output buffers are re-allocated inside one process?"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable
S, n = 20000, 250000
WHAT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rs = np.random.RandomState(20261004)
tb = genome.synthetic_snp_table(n, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
for rep in range(8):
    with RowGenerator(device=0, seed=12345) as g:
        g.set_samples(SampleTable(flags)); g.set_snps(tb)
        ms = []
        for i in range(15):
            g.ctx.generate(0, n, WHAT)
            g.ctx.sync()
            ms.append(g.ctx.last_generate_ms())
        d = g.ctx.device_buffers() if hasattr(g.ctx, "device_buffers") else None
        print("context %d: median %.3f ms  %s" % (rep, np.median(ms[3:]), d), flush=True)
