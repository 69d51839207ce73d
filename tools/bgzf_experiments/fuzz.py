"""GPU fuzz of the BGZF encoders: random sample counts, table sizes, prefix widths, forced sets and row ranges; every
output must inflate to the text.  usage: fuzz.py SECONDS [seed]"""
import gzip
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np

import __graft_entry__ as ge

ge.build()
from dna_factory_amd import genome
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable

budget = float(sys.argv[1])
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    kind = rs.randint(0, 4)
    S = int([rs.randint(2, 40), rs.randint(900, 5000), rs.randint(5000, 70000), rs.choice([15360, 15359, 15361, 30720, 61440 // 4 * 3])][kind])
    max_text = 40e6
    n = int(max(3, min(4000, max_text / (4 * S + 40))))
    n = int(rs.randint(max(3, n // 4), n + 1))
    tb = genome.synthetic_snp_table(n, min_maf=float(rs.choice([0.005, 0.05, 0.3])), rs=rs)
    if rs.rand() < 0.5:  # wider and narrower fixed columns: member boundaries fall at other places of the rows
        tb.snp_id = tb.snp_id * int(rs.choice([1, 1000, 10**9])) + int(rs.randint(0, 5))
    flags = (rs.rand(S) <= rs.choice([0.0, 0.5, 1.0])).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    forced_rows = rs.choice(n, size=min(n, 3), replace=False)
    carriers = np.sort(rs.choice(np.arange(S // 2, S), size=max(1, (S - S // 2) // 3), replace=False)).astype(np.uint32)
    st = SampleTable(flags, [int(tb.snp_id[j]) for j in forced_rows], np.arange(len(forced_rows) + 1) * len(carriers), np.tile(carriers, len(forced_rows)))
    with RowGenerator(device=0, seed=int(rs.randint(0, 2**31))) as g:
        g.set_samples(st)
        g.set_snps(tb)
        for _ in range(2):
            b = int(rs.randint(0, n))
            m = int(rs.randint(1, n - b + 1))
            if rs.rand() < 0.3:
                b, m = 0, n
            text, _, _ = g.generate(b, m, text=True, packed=False)
            blob, n_text = g.generate_bgzf(b, m, reuse_tables=bool(rs.rand() < 0.3))
            ok = n_text == len(text) and gzip.decompress(blob.tobytes()) == text.tobytes()
            cases += 1
            if not ok:
                print("MISMATCH S=%d n=%d range=(%d,%d) kind=%d" % (S, n, b, m, kind), flush=True)
                sys.exit(1)
    if cases % 20 == 0:
        print("%d cases ok, %.0f s" % (cases, time.time() - t0), flush=True)
print("fuzz done: %d cases, all inflate to the text" % cases)
