"""Placement probe on / off: whole-tile kernel time of 10 fresh contexts each (bench shape)."""
import os, sys, time
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
for probe in (False, True):
    if probe: os.environ.pop("DNAF_NO_PLACEMENT_PROBE", None)
    else: os.environ["DNAF_NO_PLACEMENT_PROBE"] = "1"
    res = []
    for rep in range(10):
        with RowGenerator(device=0, seed=12345) as g:
            g.set_samples(SampleTable(flags)); g.set_snps(tb)
            t0 = time.time()
            g.ctx.reserve_output(g.ctx.text_bytes(0, n), n * g.words_per_row)
            dt = time.time() - t0
            ms = []
            for i in range(10):
                g.ctx.generate(0, n, 3); g.ctx.sync(); ms.append(g.ctx.last_generate_ms())
            res.append((np.median(ms[2:]), dt))
    print("probe=%s: tile ms %s | reserve s %s" % (probe, " ".join("%.2f" % a for a, _ in res), " ".join("%.2f" % b for _, b in res)), flush=True)
