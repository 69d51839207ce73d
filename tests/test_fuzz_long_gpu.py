"""A longer random sweep of the row kernels against the oracle: every kernel form (short-row, line-owning, segment; Philox
and MT19937), shapes drawn around their boundaries.  DNAF_FUZZ_CASES sets the number of cases (default 40; a soak run on a
GPU box: DNAF_FUZZ_CASES=600 python -m pytest tests/test_fuzz_long_gpu.py -m gpu -q)."""
import os

import numpy as np
import pytest

from helpers import random_case
from test_rowgen_gpu import RowGenerator, check_case, oracle_run  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu

POOLS = [range(1, 70), range(60, 260), range(250, 1030), range(1000, 2040), range(2030, 2100), range(2100, 4200), range(8180, 8200),
         range(16380, 16390), range(19990, 20010)]


def test_random_shapes_all_kernel_forms(RowGenerator, oracle):
    n_cases = int(os.environ.get("DNAF_FUZZ_CASES", "40"))
    rs = np.random.RandomState(int(os.environ.get("DNAF_FUZZ_SEED", "99")))
    for case in range(n_cases):
        S = int(rs.choice(list(POOLS[case % len(POOLS)])))
        n_rows = int(rs.randint(1, 8 if S > 8000 else 40 if S > 2000 else 260))
        snps, samples = random_case(rs, n_rows, S, p_sex_chrom=[0.0, 0.15, 0.4, 0.9][case % 4], n_del=[0, 2, 6][case % 3])
        check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=int(rs.randint(0, 10**6)))


def test_random_shapes_mt19937(RowGenerator, oracle):
    n_cases = max(4, int(os.environ.get("DNAF_FUZZ_CASES", "40")) // 4)
    rs = np.random.RandomState(int(os.environ.get("DNAF_FUZZ_SEED", "99")) + 1)
    for case in range(n_cases):
        S = int(rs.choice(list(POOLS[case % 5])))
        n_rows = int(rs.randint(1, 120))
        snps, samples = random_case(rs, n_rows, S, p_sex_chrom=[0.0, 0.3, 0.8][case % 3], n_del=[0, 3][case % 2])
        seed = int(rs.randint(0, 2**31))
        rec = oracle.snp_records(snps)
        sex, ctrl, off, ids = oracle.sample_arrays(samples)
        want_text, want_off, want_packed = oracle.generate(rec, sex, ctrl, off, ids, rng_mode=oracle.RNG_MT19937, seed=seed, rng_row0=0)
        with RowGenerator(device=0, seed=seed, rng="mt19937") as g:
            g.set_samples(samples)
            g.set_snps(snps)
            text, off2, packed = g.generate()
        assert np.array_equal(off2, want_off) and text.tobytes() == want_text and np.array_equal(packed, want_packed)
