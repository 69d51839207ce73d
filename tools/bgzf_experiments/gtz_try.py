import sys, time, gzip, zlib
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd import genome, capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable

S = int(sys.argv[1]); n = int(sys.argv[2]); check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rs = np.random.RandomState(5)
tb = genome.synthetic_snp_table(n, min_maf=0.01, rs=rs)
flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
st = SampleTable(flags)
with RowGenerator(device=0, seed=7) as g:
    g.set_samples(st); g.set_snps(tb)
    for mode in (True, False):
        g.ctx.generate(0, n, capi.OUT_TEXT | capi.OUT_PACKED)
        nb = g.ctx.bgzf_compress(text_domain=mode)
        ms = g.ctx.last_bgzf_ms()
        nb2 = g.ctx.bgzf_compress(text_domain=mode, reuse_tables=True)
        ms2 = g.ctx.last_bgzf_ms()
        tbytes = g.ctx.text_bytes(0, n)
        print("text_domain=%s: %d -> %d bytes  ratio %.2f  %.4f B/genotype  %.2f ms (reuse %.2f ms)" % (mode, tbytes, nb, tbytes / nb, nb / (n * S), ms, ms2), flush=True)
        if check:
            raw = g.ctx.read_bgzf(nb).tobytes()
            text = g.ctx.read_text(tbytes).tobytes()
            t0 = time.time()
            out = gzip.decompress(raw)
            print("  inflate ok:", out == text, "%.1f s" % (time.time() - t0), flush=True)
            if out != text:
                k = next(i for i in range(min(len(out), len(text))) if out[i] != text[i])
                print("  first diff at", k, "member", k // 61440, out[k-20:k+20], text[k-20:k+20])
