"""The reference's own unit tests for the code around the path, restated against this implementation
(test/unit/common_util_test.py, snp_factory_test.py, pop_factory_test.py of dna-factory)."""
import os
import random

import numpy
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import cli, genome, population

    return cli, genome, population


def test_split_list(mods):
    # test/unit/common_util_test.py:7-12: split_list(range(100), 3) -> 33, 33, 34
    cli, _, _ = mods
    sizes = [hi - lo for lo, hi in cli.split_ranges(100, 3)]
    assert sizes == [33, 33, 34]
    assert cli.split_ranges(100, 3)[0][0] == 0 and cli.split_ranges(100, 3)[-1][1] == 100


def test_random_snp_tuples(mods):
    # test/unit/snp_factory_test.py:14-37 on 100k synthetic SNPs
    _, genome, pop = mods
    numpy.random.seed(11)
    random.seed(11)
    n, min_maf = 100000, 0.01
    tb = pop.generate_snp_table(n, min_maf=min_maf)
    assert len(tb) == n
    assert np.all(tb.n_alleles == 2)                       # biallelic
    maf = 1 - tb.cum[:, 0]
    assert maf.min() >= min_maf - 1e-12                    # min_maf respected
    assert np.all(tb.ref != tb.alt)                        # ref != alt
    maf_vals, pdf = genome.load_maf_pdf()
    keep = maf_vals >= min_maf
    p = pdf[keep] / pdf[keep].sum()
    assert abs(np.mean(np.isclose(maf, maf_vals[keep][-1])) - p[-1]) < 0.01      # share of the largest MAF bin
    assert abs(np.mean(tb.chrom == b"1") - genome.CHROMOSOME_PROB[0]) < 0.01     # share of chromosome 1


def _snp_table(pop, n):
    from dna_factory_amd.tables import SnpTable

    return SnpTable.from_snps([(i + 1, "1", 50000, [("G", 0.70), ("A", 0.90), ("T", 1.0)]) for i in range(n)])


def test_deleterious_group_from_yml(mods, tmp_path):
    # test/unit/pop_factory_test.py:36-44: 2 instances, 3 deleterious each, weights 0.5 -> 2 mutations selected
    _, _, pop = mods
    cfg = tmp_path / "g.yml"
    cfg.write_text("groupA:\n  mutation_weights: [0.5, 0.5, 0.5]\n  num_instances: 2\n  population_weight: 5\n"
                   "  min_minor_allele_freq: 0.01\n")
    numpy.random.seed(1)
    random.seed(1)
    groups = list(pop.pick_deleterious(_snp_table(pop, 4), str(cfg)).values())
    assert len(groups) == 2
    assert groups[0].population_weight == 5
    assert len(groups[0].deleterious) == 3
    assert len(groups[0].select_mutations()) == 2
    assert [g.name for g in groups] == ["groupA-0", "groupA-1"]


def test_deleterious_groups_are_drawn_by_population_weight(mods, tmp_path):
    # test/unit/pop_factory_test.py:53-75: groups 3,4 weigh 10x groups 1,2
    _, _, pop = mods
    cfg = tmp_path / "w.yml"
    cfg.write_text("".join("group%d:\n  mutation_weights: [0.5, 0.5]\n  population_weight: %d\n" % (k, 1 if k < 3 else 10)
                           for k in range(1, 5)))
    numpy.random.seed(2)
    random.seed(2)
    groups = pop.pick_deleterious(_snp_table(pop, 5), str(cfg))
    pop.build_samples(groups, 0, 100, 0.5, str(tmp_path))
    counts = {}
    for line in open(tmp_path / "pop_deleterious.txt"):
        name = line.split("\t")[1]
        counts[name] = counts.get(name, 0) + 1
    assert counts.get("group3-0", 0) > 2
    assert counts.get("group1-0", 0) < 99
    assert sum(counts.values()) == 100
