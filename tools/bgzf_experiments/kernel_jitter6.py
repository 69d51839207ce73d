"""Row kernel time vs segment length (DNAF_SEG_ITERS = trips per wave), eight fresh contexts each, no placement probe."""
import os, sys
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
os.environ["DNAF_NO_PLACEMENT_PROBE"] = "1"
for seg in (4, 8, 16, 32):
    os.environ["DNAF_SEG_ITERS"] = str(seg)
    res = []
    for rep in range(8):
        with RowGenerator(device=0, seed=12345) as g:
            g.set_samples(SampleTable(flags)); g.set_snps(tb)
            ms = []
            for i in range(8):
                g.ctx.generate(0, n, 3); g.ctx.sync(); ms.append(g.ctx.last_generate_ms())
            res.append(np.median(ms[3:]))
    print("seg_iters=%2d: %s  mean %.3f" % (seg, " ".join("%.2f" % v for v in res), np.mean(res)), flush=True)
