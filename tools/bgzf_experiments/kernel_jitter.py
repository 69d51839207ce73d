"""Per-launch durations of the row kernel on the bench tile (S = 20k, 250k rows): is the spread inside a process or
between processes?"""
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
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(SampleTable(flags)); g.set_snps(tb)
    g.ctx.reserve_output(g.ctx.text_bytes(0, n), n * g.words_per_row)  # includes the placement search
    ms = []
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        g.ctx.generate(0, n, capi.OUT_TEXT | capi.OUT_PACKED)
        g.ctx.sync()
        ms.append(g.ctx.last_generate_ms())
    ms = np.array(ms[3:])
    print("n=%d  min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f ms" % (len(ms), ms.min(), np.percentile(ms, 10), np.median(ms), np.percentile(ms, 90), ms.max()), flush=True)
    print(" first 20:", " ".join("%.2f" % v for v in ms[:20]))
