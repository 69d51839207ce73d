"""Genome constants and the synthetic SNP-table generator (structure-of-arrays, vectorised).

The tables are data: chromosome list, share of SNPs per chromosome and chromosome lengths as dna-factory uses
them (common/snp.py:8-60), and the MAF distribution of `data/snp_freq_cdf.csv` (pop_factory.py:148-158).
`synthetic_snp_table` draws from the same distributions as SnpFactory.random_snp_tuples
(pop_factory.py:172-193) but builds numpy columns directly instead of a list of objects; it does not replay
the reference's RNG call sequence (that is the seed-level front end, a later row of the scope table).
"""
import os

import numpy as np

from .tables import SnpTable

CHROMOSOME_LIST = [str(c) for c in range(1, 23)] + ["X", "Y"]
CHROMOSOME_PROB = [
    0.07426087261566, 0.07930487311426, 0.06669253502772, 0.068216704579376, 0.060859452377757, 0.061620602417568,
    0.056436996345677, 0.052745283940636, 0.041811456817423, 0.047572674763057, 0.046903788666524, 0.045558978461098,
    0.033875108161329, 0.030837930905743, 0.028329099437382, 0.030535626281104, 0.026508783521902, 0.026711126377244,
    0.022471493713103, 0.021115686613365, 0.013429462318399, 0.013635819040166, 0.048111412615406, 0.002454231888101,
]
CHROMOSOME_MAX_POSITION = {
    "1": 248946339, "2": 242765766, "3": 198235509, "4": 190181952, "5": 181477687, "6": 170744571, "7": 159335932,
    "8": 145571444, "9": 138258771, "10": 133787363, "11": 135076614, "12": 133265032, "13": 114352979,
    "14": 107270972, "15": 101981181, "16": 90228323, "17": 83247315, "18": 80262386, "19": 58607512,
    "20": 64333614, "21": 46699955, "22": 50806829, "X": 156040000, "Y": 57217333,
}
NUCLEOTIDES = ["A", "T", "C", "G"]  # order of pop_factory.py:183

_CDF_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "snp_freq_cdf.csv")


def load_maf_pdf(path=_CDF_FILE):
    """(sorted MAF values, probability of each) from the CDF table; pdf = successive differences
    (pop_factory.py:143-146)."""
    m = np.loadtxt(path, skiprows=1, delimiter=",")
    maf, cdf = m[:, 0], m[:, 1]
    pdf = np.diff(np.concatenate([[0.0], cdf]))
    return maf, pdf


def synthetic_snp_table(n, min_maf=0.005, rs=None):
    """n biallelic SNPs sorted by (chromosome string, position) -- the reference's lexicographic order
    (pop_factory.py:245); ids 1..n are assigned before the sort (pop_factory.py:186)."""
    rs = rs if rs is not None else np.random.RandomState(0)
    maf_vals, pdf = load_maf_pdf()
    keep = maf_vals >= min_maf  # first index with min_maf <= m, pop_factory.py:161-165
    p = pdf[keep] / pdf[keep].sum()
    chrom_idx = rs.choice(len(CHROMOSOME_LIST), size=n, p=np.asarray(CHROMOSOME_PROB) / np.sum(CHROMOSOME_PROB))
    maf = rs.choice(maf_vals[keep], size=n, p=p)
    max_pos = np.array([CHROMOSOME_MAX_POSITION[c] for c in CHROMOSOME_LIST], dtype=np.float64)
    position = (rs.random_sample(n) * max_pos[chrom_idx]).astype(np.int64)
    ref_idx = rs.randint(0, 4, size=n)
    alt_idx = (ref_idx + rs.randint(1, 4, size=n)) % 4  # uniform over the other three
    nt = np.array([c.encode() for c in NUCLEOTIDES], dtype="S1")
    names = np.array([c.encode() for c in CHROMOSOME_LIST], dtype="S2")
    # rank of each chromosome name in string order, then sort by (rank, position); stable like list.sort
    rank = np.argsort(np.argsort(names, kind="stable"), kind="stable")
    order = np.lexsort((position, rank[chrom_idx]))
    cum = np.ones((n, 4), dtype=np.float64)
    cum[:, 0] = 1 - maf  # (ref, 1 - maf), (alt, 1.0): pop_factory.py:187-191
    ids = np.arange(1, n + 1, dtype=np.int64)
    return SnpTable(ids[order], names[chrom_idx][order], position[order], nt[ref_idx][order], nt[alt_idx][order],
                    np.full(n, 2, dtype=np.uint8), cum[order])
