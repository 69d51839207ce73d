"""Is the row kernel's run-to-run spread the clock or the memory?  One process, one pair of buffers: every round times
the full kernel, the packed-only kernel (arithmetic only, 1.3 GB of stores) and the text-only kernel on the same tile.
If all three drift together it is the clock; if only the text ones do, the memory side.
usage: drift_watch.py [rounds] [pause_seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from dna_factory_amd import capi
from dna_factory_amd.genome import synthetic_snp_table
from dna_factory_amd.rowgen import RowGenerator

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
S, n = 20000, 250000
rs = np.random.RandomState(20261004)
tb = synthetic_snp_table(n, 0.01, rs)
st = bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(20261004))
with RowGenerator(device=0, seed=12345) as g:
    g.set_samples(st)
    g.set_snps(tb)
    g.ctx.reserve_output(g.ctx.text_bytes(0, n), n * g.words_per_row)
    t0 = time.time()
    for r in range(rounds):
        out = []
        for what in (3, 2, 1):
            g.ctx.generate(0, n, what)
            g.ctx.sync()
            g.ctx.timing_begin()
            for _ in range(8):
                g.ctx.generate(0, n, what)
            ms, nrec, _ = g.ctx.timing_end()
            out.append(ms / nrec)
        print("t=%6.1f s  text+packed %.3f ms   packed only %.3f ms   text only %.3f ms   ratio both/packed %.3f" % (
            time.time() - t0, out[0], out[1], out[2], out[0] / out[1]), flush=True)
        if pause:
            time.sleep(pause)
