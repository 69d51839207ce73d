import sys, time; sys.path.insert(0,'.')
import numpy as np, bench
t0=time.time()
from dna_factory_amd.genome import synthetic_snp_table
from dna_factory_amd.rowgen import RowGenerator
S=20000; n=400000
tb=synthetic_snp_table(n, 0.01, np.random.RandomState(1)); st=bench.synthetic_samples(S, n, tb.snp_id, np.random.RandomState(1))
print("tables %.2f" % (time.time()-t0)); t=time.time()
g=RowGenerator(device=0, seed=5, rng="mt19937"); g.__enter__(); print("ctx %.2f" % (time.time()-t)); t=time.time()
g.set_samples(st); g.set_snps(tb); print("set %.2f" % (time.time()-t)); t=time.time()
q=g.row_quantum
g.ctx.generate(0, 4096*q, 3); g.ctx.sync(); print("first generate %.2f" % (time.time()-t)); t=time.time()
g.ctx.generate(4096*q, 4096*q, 3); g.ctx.sync(); print("second generate %.2f" % (time.time()-t)); t=time.time()
