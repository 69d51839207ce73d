"""Shared helpers for the parity tests: golden fixture loading and random table builders."""
import glob
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(p)[: -len(".json.gz")] for p in glob.glob(os.path.join(GOLD, "*.json.gz")))


def load_golden(name):
    with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def golden_inputs(doc):
    """-> (snps [(id, chrom, pos, [(letter, cum)])], samples [(sex, is_control, carried ids or None)])"""
    snps = [(s[0], s[1], s[2], [(t[0], t[1]) for t in s[3]]) for s in doc["snps"]]
    samples = [(s[1], s[2], s[3]) for s in doc["samples"]]
    return snps, samples


CHROMS = [str(c) for c in range(1, 23)] + ["X", "Y", "MT"]


def random_case(rs, n_rows, S, multi=True, n_del=6, p_sex_chrom=0.3):
    """Random SNP list + sample table exercising every cell rule (pop_factory.py:479-499)."""
    letters = ["A", "C", "G", "T"]
    snps = []
    for j in range(n_rows):
        if rs.rand() < p_sex_chrom:
            chrom = ["X", "Y", "MT"][rs.randint(0, 3)]
        else:
            chrom = str(rs.randint(1, 23))
        perm = list(rs.permutation(letters))
        maf = 0.005 * rs.randint(1, 100)
        k = rs.randint(0, 10) if multi else 0
        if k == 8:
            tuples = [(perm[0], 0.70), (perm[1], 0.90), (perm[2], 1.0)]
        elif k == 9:
            tuples = [(perm[0], 0.55), (perm[1], 0.80), (perm[2], 0.93), (perm[3], 1.0)]
        else:
            tuples = [(perm[0], 1 - maf), (perm[1], 1.0)]
        snps.append((j + 1, chrom, int(rs.randint(0, 250000000)), tuples))
    ids = [s[0] for s in snps]
    del_ids = list(rs.choice(ids, size=min(n_del, len(ids)), replace=False)) if ids and n_del else []
    n_ctrl = S // 2
    samples = []
    for i in range(S):
        ctrl = i < n_ctrl
        sex = 1 if rs.rand() <= 0.5 else 2
        carried = None
        if not ctrl:
            k = rs.randint(0, len(del_ids) + 1) if del_ids else 0
            carried = [int(x) for x in rs.choice(del_ids, size=k, replace=False)] if k else []
        samples.append((sex, ctrl, carried))
    return snps, samples
