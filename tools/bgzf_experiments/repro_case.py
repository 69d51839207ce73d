import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import test_bgzf as tb_mod
from dna_factory_amd.rowgen import RowGenerator
S, n = int(sys.argv[1]), int(sys.argv[2])
tb, st = tb_mod._sorted_case(S, n, seed=S + n)
with RowGenerator(device=0, seed=11) as g:
    g.set_samples(st); g.set_snps(tb)
    print("generate", flush=True)
    blob, n_text = g.generate_bgzf()
    print("ok", len(blob), n_text, flush=True)
