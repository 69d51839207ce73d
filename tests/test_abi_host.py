"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/dnaf.h declares, fails loudly
without a GPU, and its host helper formats the fixed columns like pop_factory.py:503-507."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge

    ge.build()
    import dna_factory_amd.capi as capi

    return capi


def test_header_symbols_are_exported(capi):
    hdr = open(os.path.join(REPO, "include", "dnaf.h")).read()
    declared = set(re.findall(r"\b(dnaf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dnaf_ctx"}
    assert declared == set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(capi.lib, name)
    assert capi.lib.dnaf_abi_version() == capi.ABI_VERSION


def test_no_silent_cpu_fallback(capi):
    try:
        n = capi.device_count()
    except capi.DnafError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.DnafError):
        capi.Context(0)
    from dna_factory_amd.rowgen import RowGenerator

    with pytest.raises(capi.DnafError):
        RowGenerator(device=0)


def test_format_prefixes_matches_reference_format(capi):
    rs = np.random.RandomState(3)
    n = 500
    chroms = np.array([str(c) for c in range(1, 23)] + ["X", "Y", "MT"])[rs.randint(0, 25, n)]
    pos = rs.randint(0, 250000000, n)
    pos[:3] = [0, 9, 10]
    ids = rs.randint(1, 10**9, n)
    ids[:2] = [1, 2**40]
    ref = np.array(list("ACGT"))[rs.randint(0, 4, n)]
    alt = np.array(["A", "C,T", "G,A,T", "T"])[rs.randint(0, 4, n)]
    blob, off = capi.format_prefixes(chroms.astype("S"), pos, ids, ref.astype("S"), alt.astype("S"))
    b = blob.tobytes()
    for j in range(n):
        want = "%s\t%i\trs%s\t%s\t%s\t40\tPASS\t.\tGT\t" % (chroms[j], pos[j], ids[j], ref[j], alt[j])
        assert b[int(off[j]): int(off[j + 1])].decode() == want


def test_tables_from_reference_shaped_objects(capi):
    from dna_factory_amd.tables import SampleTable, SnpTable

    class Snp:  # shaped like SNPTuples, pop_factory.py:74-85
        def __init__(self, i, c, p, t):
            self.id, self.chromosome, self.position, self.tuples = i, c, p, t

    tb = SnpTable.from_snps([Snp(5, "X", 10, [("A", 0.9), ("T", 1.0)]), Snp(6, "7", 11, [("G", 0.5), ("A", 0.7), ("C", 1.0)]),
                             Snp(7, "MT", 12, [("G", 1.0)])])
    assert list(tb.chrom_class) == [capi.CLASS_X, capi.CLASS_AUTOSOME, capi.CLASS_MT]
    assert list(tb.alt) == [b"T", b"A,C", b"G"] and list(tb.ref) == [b"A", b"G", b"G"]
    assert list(tb.n_alleles) == [2, 3, 1]

    class Sam:  # shaped like SampleInfo, pop_factory.py:47-60
        def __init__(self, pid, sex, ctrl, d):
            self.person_id, self.sex, self.is_control, self.deleterious_snps = pid, sex, ctrl, d

    # str keys never match an int id (`snp.id not in sample.deleterious_snps`, pop_factory.py:485)
    st = SampleTable.from_samples([Sam(1, 1, True, None), Sam(2, 2, False, {5: 1.0, "6": 1.0}), Sam(3, 1, False, {5: 0.5, 7: 0.5})])
    assert list(st.flags) == [capi.SAMPLE_MALE | capi.SAMPLE_CONTROL, 0, capi.SAMPLE_MALE]
    assert list(st.forced_ids) == [5, 7]
    assert list(st.carrier_off) == [0, 2, 3] and list(st.carrier_idx) == [1, 2, 2]
    with pytest.raises(TypeError):
        SampleTable.from_samples([Sam(2, 2, False, None)])
