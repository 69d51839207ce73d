"""GPU: bits per cell and encoder time by minor-allele frequency (S = 20,000, rows of one MAF each), for the genotype-domain
encoder (and, with `text` as argument, the text-domain one).  Time is scaled to the bench tile (250,000 rows) and
weighted with the share of each class in the bench table (-f 0.01) to estimate the bench figure.
usage (GPU box): [DNAF_LIB=...] python tools/bgzf_experiments/gtz_bymaf.py [text]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import __graft_entry__ as ge

ge.build()
from dna_factory_amd import capi
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable, SnpTable

S, n = 20000, 12000
flags = (np.random.RandomState(1).rand(S) <= 0.5).astype(np.uint8)
# (MAF, share of the rows of the bench table in that class of the encoder's eight)
CLASSES = ((0.01, 0.230), (0.03, 0.173), (0.06, 0.136), (0.10, 0.094), (0.16, 0.095), (0.25, 0.104), (0.35, 0.087), (0.45, 0.081))
tot_ms = tot_bits = 0.0
for p, share in CLASSES:
    cum = np.ones((n, 4))
    cum[:, 0] = 1 - p
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"1", dtype="S2"), np.arange(n) * 100 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"T", dtype="S1"), np.full(n, 2, np.uint8), cum)
    with RowGenerator(device=0, seed=7) as g:
        g.set_samples(SampleTable(flags))
        g.set_snps(tb)
        g.ctx.generate(0, n, capi.OUT_TEXT | capi.OUT_PACKED)
        line = "p=%.3f" % p
        if "text" in sys.argv[1:]:
            a = g.ctx.bgzf_compress(text_domain=True)
            line += "  text-domain %.3f bits/cell %.2f ms/250k rows" % (a * 8 / (n * S), g.ctx.last_bgzf_ms() * 250000 / n)
        g.ctx.bgzf_compress()
        b = g.ctx.bgzf_compress(reuse_tables=True)
        ms = g.ctx.last_bgzf_ms() * 250000 / n
        line += "  genotype-domain %.3f bits/cell %.2f ms/250k rows" % (b * 8 / (n * S), ms)
        tot_ms += share * ms
        tot_bits += share * b * 8 / (n * S)
        print(line, flush=True)
print("weighted with the bench table's class shares: %.2f ms per 250k-row tile, %.4f B/genotype" % (tot_ms, tot_bits / 8))
