"""numpy-compatible MT19937 mode: jump-ahead machinery (CPU) and the GPU path against the UNMODIFIED reference
(fixtures `lines_mt`: pop_factory.py:471-513 after numpy.random.seed(k), no patching) and the oracle's MT mode."""
import ctypes
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from helpers import golden_inputs, golden_names, load_golden, random_case


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import capi

    return capi


def _regen(k):
    k = k.copy()
    for i in range(624):
        y = (int(k[i]) & 0x80000000) | (int(k[(i + 1) % 624]) & 0x7FFFFFFF)
        k[i] = int(k[(i + 397) % 624]) ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
    return k


def test_characteristic_polynomial(capi):
    w = np.zeros(312, np.uint64)
    d = ctypes.c_int(0)
    assert capi.lib.dnaf_mt_characteristic_polynomial(capi.ptr(w), ctypes.byref(d)) == 0
    assert d.value == 19937
    assert sum(bin(int(x)).count("1") for x in w) == 135  # the known weight of MT19937's characteristic polynomial


def test_host_jump_matches_stepping_and_numpy(capi):
    np.random.seed(4242)
    key = np.random.get_state()[1].astype(np.uint32)
    blocks = [key]
    for _ in range(4):
        blocks.append(_regen(blocks[-1]))
    seq = np.concatenate(blocks)
    out = np.zeros(624, np.uint32)
    for J in (1, 2, 226, 227, 623, 624, 625, 1871):
        assert capi.lib.dnaf_mt_jump_host(capi.ptr(key), J, capi.ptr(out)) == 0
        assert np.array_equal(out, seq[J: J + 624])
    # a long jump against numpy itself: consuming 624 * k words leaves numpy's key = the window k blocks later
    J = 624 * 7001
    np.random.rand(J // 2)
    st = np.random.get_state()
    assert st[2] == 624
    assert capi.lib.dnaf_mt_jump_host(capi.ptr(key), J, capi.ptr(out)) == 0
    assert np.array_equal(out, st[1].astype(np.uint32))


@pytest.fixture(scope="module")
def RowGenerator(capi):
    from dna_factory_amd.rowgen import RowGenerator

    return RowGenerator


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in golden_names() if not n.endswith("_philox")])
def test_gpu_mt19937_reproduces_the_unmodified_reference(RowGenerator, name):
    doc = load_golden(name)
    if doc.get("lines_mt") is None:
        pytest.skip("fixture has no MT lines")
    snps, samples = golden_inputs(doc)
    st = (doc["mt_state_key"], doc["mt_state_pos"]) if "mt_state_key" in doc else None
    with RowGenerator(device=0, seed=doc["seed"], rng="mt19937", mt_state=st) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        assert g.lines() == doc["lines_mt"]


def _oracle_mt(orc, snps, samples, seed, rng_row0=0, mt_state=None):
    rec = orc.snp_records(snps)
    sex, ctrl, off, ids = orc.sample_arrays(samples)
    return orc.generate(rec, sex, ctrl, off, ids, rng_mode=orc.RNG_MT19937, seed=seed, rng_row0=rng_row0, mt_state=mt_state)


@pytest.mark.gpu
def test_gpu_mt19937_rolls_that_tie_with_a_threshold_inside_a_row(RowGenerator, oracle):
    """The row kernel's 64-sample step looks only at the high 27 bits of a roll and leaves a roll whose high part EQUALS
    the threshold's to its general step.  Here thresholds are the very doubles numpy draws for samples in the middle of a
    row (a roll equal to the threshold picks the lower index), their neighbours one ulp down and up, and doubles
    that share the roll's high 27 bits but differ below."""
    S, seed, n_rows = 300, 4242, 8
    numpy_rs = np.random.RandomState(seed)
    d = [numpy_rs.rand(2 * S) for _ in range(n_rows)]
    def near(x, k):  # same high 27 bits of the 53-bit numerator, different low 26
        r = int(x * 2.0**53)
        return ((r & ~((1 << 26) - 1)) | ((r + k) & ((1 << 26) - 1))) / 2.0**53
    cums = [float(d[0][40]), float(d[1][41]), float(np.nextafter(d[2][90], 0.0)), float(np.nextafter(d[3][17], 1.0)),
            near(d[4][66], 1), near(d[5][67], -1), float(d[6][127]), float(d[7][128])]
    letters = "ACGT"
    snps = [(r + 1, str(1 + r % 22), 1000 + r, [(letters[r % 4], cums[r]), (letters[(r + 1) % 4], 1.0)]) for r in range(n_rows)]
    samples = [(1 + (i % 2), True, None) for i in range(S)]
    want_text, want_off, want_packed = _oracle_mt(oracle, snps, samples, seed)
    with RowGenerator(device=0, seed=seed, rng="mt19937") as g:
        g.set_samples(samples)
        g.set_snps(snps)
        text, off, packed = g.generate()
    assert np.array_equal(off, want_off)
    assert text.tobytes() == want_text
    assert np.array_equal(packed, want_packed)
    # the ties are really there: draw 40 of row 0 equals its threshold, so that allele is index 0 ("0"), and one ulp less flips it
    lines = want_text.decode().split("\n")
    assert lines[0].split("\t")[9 + 20].split("/")[0] == "0" and lines[2].split("\t")[9 + 45].split("/")[0] == "1"


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,S", [(5, 1), (40, 3), (200, 39), (45, 156), (30, 257), (90, 1000), (8, 20000)])
def test_gpu_mt19937_matches_oracle(RowGenerator, oracle, n_rows, S):
    rs = np.random.RandomState(n_rows * 7 + S)
    snps, samples = random_case(rs, n_rows, S)
    seed = int(rs.randint(0, 2**31))
    want_text, want_off, want_packed = _oracle_mt(oracle, snps, samples, seed)
    with RowGenerator(device=0, seed=seed, rng="mt19937") as g:
        g.set_samples(samples)
        g.set_snps(snps)
        text, off, packed = g.generate()
        t_only, _, _ = g.generate(text=True, packed=False)
    assert np.array_equal(off, want_off)
    assert text.tobytes() == want_text
    assert np.array_equal(packed, want_packed)
    assert t_only.tobytes() == want_text


@pytest.mark.gpu
def test_gpu_mt19937_tiles_and_explicit_state(RowGenerator, oracle):
    """Sequential tiles (states advance by one jump per tile), random access (rebuild), and a mid-block entry state."""
    rs = np.random.RandomState(31)
    S = 52  # 156 / gcd(52, 156) = 3 rows per wave
    snps, samples = random_case(rs, 300, S)
    np.random.seed(99)
    np.random.rand(1234)  # the parent has drawn before the worker starts: entry position is mid-block
    st = np.random.get_state()
    state = (st[1].astype(np.uint32), int(st[2]))
    assert 0 < state[1] < 624
    want_text, want_off, want_packed = _oracle_mt(oracle, snps, samples, 0, mt_state=state)
    with RowGenerator(device=0, rng="mt19937", mt_state=state) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        assert g.row_quantum == 3
        pieces = []
        for b in range(0, 300, 60):  # sequential tiles of 20 waves each
            t, _, p = g.generate(b, 60)
            pieces.append(t.tobytes())
            assert np.array_equal(p, want_packed[b: b + 60])
        assert b"".join(pieces) == want_text
        for b, n in ((240, 60), (0, 3), (150, 33), (297, 3)):  # random access
            t, _, _ = g.generate(b, n)
            assert t.tobytes() == want_text[int(want_off[b]): int(want_off[b + n])]
        from dna_factory_amd import capi

        with pytest.raises(capi.DnafError) as e:
            g.generate(1, 3)  # not on a wave boundary
        assert e.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_gpu_mt19937_shard_offset(RowGenerator, oracle):
    rs = np.random.RandomState(32)
    snps, samples = random_case(rs, 156, 100)  # 39 rows per wave
    whole, off, _ = _oracle_mt(oracle, snps, samples, 2024)
    with RowGenerator(device=0, seed=2024, rng="mt19937") as g:
        g.set_samples(samples)
        g.set_snps(snps[78:], rng_row0=78)
        t, _, _ = g.generate()
    assert t.tobytes() == whole[int(off[78]):]


@pytest.mark.gpu
def test_gpu_mt19937_new_sample_table_invalidates_cached_states(RowGenerator, oracle):
    """Wave start states depend on the number of samples; a context reused with another sample table must rebuild them."""
    rs = np.random.RandomState(41)
    with RowGenerator(device=0, seed=321, rng="mt19937") as g:
        for S in (52, 100, 52):
            snps, samples = random_case(rs, 78, S)
            want, _, _ = _oracle_mt(oracle, snps, samples, 321)
            g.set_samples(samples)
            g.set_snps(snps)
            t, _, _ = g.generate()
            assert t.tobytes() == want


@pytest.mark.gpu
def test_dropin_write_vcf_snps_with_reference_shaped_objects(RowGenerator, oracle):
    """The INTEGRATION.md stub: objects shaped like the reference's SampleInfo / SNPTuples, a writable sink, and the
    parent's numpy stream handed over at the point where the reference forks its worker."""
    import io

    class SampleInfo:  # pop_factory.py:47-71
        def __init__(self, fid, iid, sex, ctrl, dele):
            self.family_id, self.person_id, self.sex, self.is_control, self.deleterious_snps = fid, iid, sex, ctrl, dele

        def is_male(self):
            return self.sex == 1

    class SNPTuples:  # pop_factory.py:74-85
        def __init__(self, sid, chrom, pos, tuples):
            self.id, self.chromosome, self.position, self.tuples = sid, chrom, pos, tuples

    rs = np.random.RandomState(77)
    snps, samples = random_case(rs, 90, 60)
    fam = [SampleInfo(i + 1, 100001 + i, sx, c, None if c else {k: 1.0 for k in (car or [])}) for i, (sx, c, car) in enumerate(samples)]
    ref_snps = [SNPTuples(*s) for s in snps]
    np.random.seed(235959)
    np.random.rand(len(fam))  # generate_fam_file drew the sexes before the fork (pop_factory.py:352)
    state = np.random.get_state()
    want, _, _ = _oracle_mt(oracle, snps, samples, 0, mt_state=(state[1].astype(np.uint32), int(state[2])))
    sink = io.BytesIO()
    with RowGenerator(device=0, rng="mt19937", mt_state=state[1:3]) as g:
        assert g.write_vcf_snps(fam, ref_snps, sink) == 90
    assert sink.getvalue() == want


@pytest.mark.gpu
def test_mt_mode_first_use_from_several_threads(tmp_path):
    """Multi-GPU runs create one context per host thread; the jump-ahead machinery (characteristic polynomial, found
    once per process) must come up safely when several threads need it first at the same moment.  Fresh processes,
    four threads each, all results equal to the single-threaded answer."""
    import subprocess
    import sys

    code = r'''
import sys, threading, time
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
ge.build()
from dna_factory_amd.rowgen import RowGenerator
from dna_factory_amd.tables import SampleTable, SnpTable
S, n = 52, 600
flags = (np.arange(S) %% 2).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
cum = np.ones((n, 4)); cum[:, 0] = 1 - 0.005 * (1 + np.arange(n) %% 90)
tb = SnpTable(np.arange(1, n + 1), np.full(n, b"5", dtype="S2"), np.arange(n) + 1, np.full(n, b"A", dtype="S1"),
              np.full(n, b"C", dtype="S1"), np.full(n, 2, np.uint8), cum)
out = [None] * 4
gate = threading.Barrier(4)
def work(k):
    with RowGenerator(device=0, seed=4321, rng="mt19937") as g:
        g.set_samples(SampleTable(flags)); g.set_snps(tb)
        q = g.row_quantum
        gate.wait()  # the first generate of the process builds the jump machinery: the four arrive a few ms apart
        time.sleep(0.004 * k)
        parts = [g.generate(b, min(150 // q * q, n - b))[0].tobytes() for b in range(0, n, 150 // q * q)]
        out[k] = b"".join(parts)
ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
with RowGenerator(device=0, seed=4321, rng="mt19937") as g:
    g.set_samples(SampleTable(flags)); g.set_snps(tb)
    ref = g.generate()[0].tobytes()
assert all(o == ref for o in out), [o == ref for o in out]
print("ok")
''' % REPO
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
