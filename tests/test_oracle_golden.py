"""The CPU oracle against the real reference worker (pop_factory.py:471-513).

Fixtures under tests/golden were produced by oracle/make_golden.py from the
imported reference; when /root/reference is present (build container only)
the reference is additionally driven live on fresh random cases.
"""
import os
import sys

import numpy as np
import pytest

from helpers import golden_inputs, golden_names, load_golden, random_case


def _oracle_lines(orc, snps, samples, mode, seed, mt_state=None):
    rec = orc.snp_records(snps)
    sex, ctrl, off, ids = orc.sample_arrays(samples)
    text, row_off, packed = orc.generate(rec, sex, ctrl, off, ids, rng_mode=mode, seed=seed, mt_state=mt_state)
    lines = [text[int(row_off[j]): int(row_off[j + 1])].decode() for j in range(len(rec))]
    return lines, packed


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_fixture(oracle, name):
    doc = load_golden(name)
    snps, samples = golden_inputs(doc)
    if doc.get("lines_mt") is not None:
        st = (np.array(doc["mt_state_key"], dtype=np.uint32), doc["mt_state_pos"]) if "mt_state_key" in doc else None
        lines, _ = _oracle_lines(oracle, snps, samples, oracle.RNG_MT19937, doc["seed"], st)
        assert lines == doc["lines_mt"]
    if doc.get("lines_philox") is not None:
        lines, _ = _oracle_lines(oracle, snps, samples, oracle.RNG_PHILOX, doc["seed"])
        assert lines == doc["lines_philox"]


def test_pick_allele_index_reference_known_answers(oracle):
    # test/unit/pop_factory_test.py:24-28 -- thresholds 0.70/0.90/1.0: roll 0.95 -> 2, roll 0.4 -> 0
    rec = oracle.snp_records([(100, "1", 50000, [("G", 0.70), ("A", 0.90), ("T", 1.0)])])
    p = rec.ctypes.data
    assert oracle.lib().oracle_pick_allele_index(p, 0.95) == 2
    assert oracle.lib().oracle_pick_allele_index(p, 0.4) == 0
    # `>=`: a roll equal to a threshold picks that (lower) index, pop_factory.py:92-95
    assert oracle.lib().oracle_pick_allele_index(p, 0.70) == 0
    assert oracle.lib().oracle_pick_allele_index(p, np.nextafter(0.70, 1.0)) == 1
    assert oracle.lib().oracle_pick_allele_index(p, 1.0) == 2


def test_mt19937_matches_numpy_legacy(oracle):
    for seed in (0, 1, 12345, 235959, 2**32 - 1):
        np.random.seed(seed)
        want = np.random.rand(1500)
        assert np.array_equal(oracle.mt_rand(seed, 1500), want)
    np.random.seed(77)
    np.random.rand(1000)
    assert np.array_equal(oracle.mt_rand(77, 10, skip=1000), np.random.rand(10))


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in oracle.philox4x32_10(ctr, key)) == want


def test_philox_numpy_draws_match_c(oracle):
    # the numpy draw function used to patch the reference == the C oracle's draws
    seed, row, S = 0x1234567890ABCDEF, (1 << 33) + 5, 37
    u = oracle.philox_row_uniforms(seed, row, 2 * S)
    for n in (0, 1, 5, 2 * S - 1):
        blk = oracle.philox4x32_10((n >> 2, 0, row & 0xFFFFFFFF, row >> 32), (seed & 0xFFFFFFFF, seed >> 32))
        assert u[n] == float(blk[n & 3]) / 4294967296.0


def test_packed_matches_text(oracle):
    rs = np.random.RandomState(5)
    snps, samples = random_case(rs, 40, 53, multi=False)
    lines, packed = _oracle_lines(oracle, snps, samples, oracle.RNG_PHILOX, 9)
    for j, line in enumerate(lines):
        cells = line.rstrip("\n").split("\t")[9:]
        for i, c in enumerate(cells):
            bits = (int(packed[j, i // 16]) >> (2 * (i % 16))) & 3
            if c == ".":
                want = 0
            elif "/" in c:
                a, b = c.split("/")
                want = (a != "0") | ((b != "0") << 1)
            else:
                want = int(c != "0")
            assert bits == want


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference only exists in the build container")
def test_oracle_matches_live_reference(oracle):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import make_golden as mg

    pf_mod = mg.import_reference()
    rs = np.random.RandomState(2026)
    for trial in range(6):
        n_rows, S = int(rs.randint(1, 40)), int(rs.randint(0, 90))
        snps, samples = random_case(rs, n_rows, S)
        ref_snps = [mg.make_snp(pf_mod, *s) for s in snps]
        spec = [(100001 + i, sx, c, ({k: 1.0 for k in car} if car is not None else None))
                for i, (sx, c, car) in enumerate(samples)]
        fam = mg.make_samples(pf_mod, spec)
        seed = 1000 + trial
        want_mt = mg.run_worker_mt(pf_mod, fam, ref_snps, seed)
        want_ph = mg.run_worker_philox(pf_mod, fam, ref_snps, seed)
        assert _oracle_lines(oracle, snps, samples, oracle.RNG_MT19937, seed)[0] == want_mt
        assert _oracle_lines(oracle, snps, samples, oracle.RNG_PHILOX, seed)[0] == want_ph
