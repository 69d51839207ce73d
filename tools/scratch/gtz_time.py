import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable
S = int(sys.argv[1]); n = int(sys.argv[2])
rs = np.random.RandomState(5)
tb = genome.synthetic_snp_table(n, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
with RowGenerator(device=0, seed=7) as g:
    g.set_samples(SampleTable(flags)); g.set_snps(tb)
    g.ctx.generate(0, n, capi.OUT_TEXT | capi.OUT_PACKED)
    os.environ.pop("DNAF_GTZ_STOP", None)
    g.ctx.bgzf_compress()
    for stop in (3, 4, 0):
        os.environ["DNAF_GTZ_STOP"] = str(stop)
        g.ctx.bgzf_compress(reuse_tables=True)
        print("stop_after=%d  total %.2f ms" % (stop, g.ctx.last_bgzf_ms()), flush=True)
