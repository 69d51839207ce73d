"""BASELINE.json configurations at their full sample counts, through the C ABI.

config 2  1k cases + 1k controls x 1M SNPs, default deleterious.yml: every byte of text and packed matrix against
          the CPU oracle (threaded verifier).
config 5  polygenic stress, 4 deleterious group instances, 25k + 25k samples: per-SNP case / control alt-allele
          frequencies against the configuration.
"""
import os
import random

import numpy
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import population
    from dna_factory_amd.rowgen import RowGenerator

    return population, RowGenerator


def _host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def test_config2_full_grid_bit_exact(mods, oracle, tmp_path):
    pop, RowGenerator = mods
    numpy.random.seed(20261004)
    random.seed(20261004)
    M, n_case, n_ctrl, seed = 1_000_000, 1000, 1000, 12345
    tb = pop.generate_snp_table(M, min_maf=0.005)
    groups = pop.pick_deleterious(tb, os.path.join(REPO, "deleterious.yml"))
    st = pop.build_samples(groups, n_ctrl, n_case, 0.5, str(tmp_path))
    assert len(st.forced_ids) > 20  # the default config forces up to 66 SNPs
    sex, ctrl, doff, dids = oracle.sample_arrays_from_table(st.flags, st.forced_ids, st.carrier_off, st.carrier_idx)
    threads = _host_threads()
    tile = 100_000
    with RowGenerator(device=0, seed=seed) as g:
        g.set_samples(st)
        g.set_snps(tb)
        for b in range(0, M, tile):
            text, off, packed = g.generate(b, tile)
            bad = oracle.verify_threads(oracle.table_records(tb, b, b + tile), sex, ctrl, doff, dids, seed, threads,
                                        text, off, packed, rng_row0=b)
            assert bad is None, "row %d differs from the oracle" % (b + bad)


def test_config5_case_control_allele_frequency_skew(mods, tmp_path):
    pop, RowGenerator = mods
    cfg = tmp_path / "four_groups.yml"
    cfg.write_text(
        "one_of_three:\n  mutation_weights: [1, 1, 1]\n  num_instances: 2\n  population_weight: 1\n  max_minor_allele_freq: 0.40\n"
        "all_of_two:\n  mutation_weights: [0.5, 0.5]\n  num_instances: 1\n  population_weight: 2\n  min_minor_allele_freq: 0.02\n"
        "single:\n  mutation_weights: [1]\n  num_instances: 1\n  population_weight: 4\n")
    numpy.random.seed(5)
    random.seed(5)
    M, n_case, n_ctrl = 60_000, 25_000, 25_000
    S = n_case + n_ctrl
    tb = pop.generate_snp_table(M, min_maf=0.01)
    groups = pop.pick_deleterious(tb, str(cfg))
    assert len(groups) == 4
    st = pop.build_samples(groups, n_ctrl, n_case, 0.5, str(tmp_path))
    with RowGenerator(device=0, seed=77) as g:
        g.set_samples(st)
        g.set_snps(tb)
        _, _, packed = g.generate(text=False, packed=True)
    bits = np.unpackbits(packed.view(np.uint8), axis=1, bitorder="little")[:, : 2 * S].reshape(M, S, 2)
    male = (st.flags & 1).astype(bool)
    cls = tb.chrom_class
    maf = 1 - tb.cum[:, 0]
    # expected share of cases carrying each deleterious SNP, from the configuration
    total_w = 1 + 1 + 2 + 4
    expect_q = {}
    for name, grp in groups.items():
        share = grp.population_weight / total_w
        frac = {"one_of_three": 1 / 3, "all_of_two": 1.0, "single": 1.0}[name.rsplit("-", 1)[0]]
        for sid in grp.deleterious:
            expect_q[sid] = expect_q.get(sid, 0.0) + share * frac
    row_of = {int(s): j for j, s in enumerate(tb.snp_id)}
    checked = 0
    for sid, q in expect_q.items():
        j = row_of[sid]
        if cls[j] == 2:
            continue  # Y rows: female cells are '.', skip
        # alleles actually present per sample on this row: 2 on diploid cells, 1 on haploid
        ploidy = np.where((cls[j] == 3) | ((cls[j] == 1) & male), 1, 2)
        alt = bits[j, :, 0] + bits[j, :, 1]
        ctrl_af = alt[:n_ctrl].sum() / ploidy[:n_ctrl].sum()
        case_af = alt[n_ctrl:].sum() / ploidy[n_ctrl:].sum()
        want = maf[j] + q * (1 - maf[j])
        sd_q = np.sqrt(q * (1 - q) / n_case)
        assert abs(ctrl_af - maf[j]) < 5 * np.sqrt(maf[j] * (1 - maf[j]) / (2 * n_ctrl)) + 1e-9
        assert abs(case_af - want) < 5 * (sd_q + np.sqrt(maf[j] / n_case)) + 1e-9
        assert case_af > ctrl_af + 0.5 * q  # the skew the configuration asks for is there
        checked += 1
    assert checked >= 7
    # no skew anywhere else: case and control frequencies agree on a sample of ordinary autosomal rows
    plain = [j for j in range(0, M, 997) if int(tb.snp_id[j]) not in expect_q and cls[j] == 0]
    for j in plain:
        alt = bits[j, :, 0] + bits[j, :, 1]
        sd = np.sqrt(2 * maf[j] * (1 - maf[j]) / (2 * n_ctrl))
        assert abs(alt[:n_ctrl].mean() / 2 - alt[n_ctrl:].mean() / 2) < 6 * sd + 1e-9


def test_config5_full_size_on_gpu_counts(mods, tmp_path):
    """BASELINE config 5 at its stated size -- 4 deleterious group instances, 25k + 25k samples x 1M SNPs -- with the
    per-row case / control allele counts taken on the GPU from the packed matrix (dnaf_allele_counts)."""
    pop, RowGenerator = mods
    cfg = tmp_path / "four_groups.yml"
    cfg.write_text(
        "one_of_three:\n  mutation_weights: [1, 1, 1]\n  num_instances: 2\n  population_weight: 1\n  max_minor_allele_freq: 0.40\n"
        "all_of_two:\n  mutation_weights: [0.5, 0.5]\n  num_instances: 1\n  population_weight: 2\n  min_minor_allele_freq: 0.02\n"
        "single:\n  mutation_weights: [1]\n  num_instances: 1\n  population_weight: 4\n")
    numpy.random.seed(55)
    random.seed(55)
    M, n_case, n_ctrl = 1_000_000, 25_000, 25_000
    tb = pop.generate_snp_table(M, min_maf=0.01)
    groups = pop.pick_deleterious(tb, str(cfg))
    st = pop.build_samples(groups, n_ctrl, n_case, 0.5, str(tmp_path))
    counts = np.zeros((M, 2), dtype=np.int64)
    tile = 100_000
    with RowGenerator(device=0, seed=501) as g:
        g.set_samples(st)
        g.set_snps(tb)
        for b in range(0, M, tile):
            g.ctx.generate(b, tile, 2)  # packed only
            counts[b: b + tile] = g.ctx.allele_counts(tile)
            if b == 0:  # the GPU counts agree with counting the packed words on the host
                packed = g.ctx.read_packed(tile, g.words_per_row)[:2000]
                bits = np.unpackbits(packed.view(np.uint8), axis=1, bitorder="little")[:, : 2 * (n_case + n_ctrl)]
                per_sample = bits.reshape(2000, -1, 2).sum(axis=2)
                assert np.array_equal(per_sample[:, :n_ctrl].sum(axis=1), counts[:2000, 0])
                assert np.array_equal(per_sample[:, n_ctrl:].sum(axis=1), counts[:2000, 1])
    male = (st.flags & 1).astype(bool)
    n_male_ctrl, n_male_case = int(male[:n_ctrl].sum()), int(male[n_ctrl:].sum())
    cls = tb.chrom_class
    maf = 1 - tb.cum[:, 0]
    # alleles present per row among controls / cases: 2 per diploid cell, 1 per haploid cell, 0 for '.'
    def denom(n, n_male):
        return np.select([cls == 0, cls == 1, cls == 2], [2 * n, 2 * n - n_male, n_male], default=n).astype(float)

    ctrl_af = counts[:, 0] / denom(n_ctrl, n_male_ctrl)
    case_af = counts[:, 1] / denom(n_case, n_male_case)
    forced_ids = set(int(s) for grp in groups.values() for s in grp.deleterious)
    is_forced = np.isin(tb.snp_id, list(forced_ids))
    assert is_forced.sum() == len(forced_ids) == 3 * 2 + 2 + 1
    plain = ~is_forced
    # ordinary rows: controls and cases both sit on the row's MAF; z-scores behave like a standard normal
    se = np.sqrt(maf * (1 - maf) / denom(n_ctrl, n_male_ctrl))
    z_ctrl = (ctrl_af - maf)[plain] / se[plain]
    z_case = ((case_af - maf) / np.sqrt(maf * (1 - maf) / denom(n_case, n_male_case)))[plain]
    for z in (z_ctrl, z_case):
        assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and np.abs(z).max() < 6.5
    # deleterious rows: cases are skewed by the share of cases the configuration gives the SNP
    total_w = 1 + 1 + 2 + 4
    for name, grp in groups.items():
        share = grp.population_weight / total_w
        frac = {"one_of_three": 1 / 3, "all_of_two": 1.0, "single": 1.0}[name.rsplit("-", 1)[0]]
        for sid in grp.deleterious:
            j = int(np.flatnonzero(tb.snp_id == sid)[0])
            if cls[j] == 2:
                continue
            q = share * frac
            want = maf[j] + q * (1 - maf[j])
            assert abs(ctrl_af[j] - maf[j]) < 5 * se[j]
            assert abs(case_af[j] - want) < 5 * (np.sqrt(q * (1 - q) / n_case) + np.sqrt(maf[j] / n_case))
