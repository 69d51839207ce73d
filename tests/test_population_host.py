"""Host-side feeders against the reference's own feeders (fixture full_flow_120x400, produced by
oracle/make_golden.py from SnpFactory / pick_deleterious_snps / generate_fam_file with fixed seeds)."""
import gzip
import os
import random

import numpy
import numpy as np
import pytest

from helpers import load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pop():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import population

    return population


def test_feeders_replay_the_reference_at_fixed_seeds(pop, tmp_path):
    doc = load_golden("full_flow_120x400")
    numpy.random.seed(doc["numpy_seed"])
    random.seed(doc["python_random_seed"])
    tb = pop.generate_snp_table(400, min_maf=0.01)
    groups = pop.pick_deleterious(tb, os.path.join(REPO, "deleterious.yml"))
    pop.write_deleterious_json(groups, tmp_path / "deleterious.json")
    st = pop.build_samples(groups, 60, 60, 0.5, str(tmp_path))
    # SNP list, in sorted order
    for j, (sid, chrom, position, tuples) in enumerate(doc["snps"]):
        assert (int(tb.snp_id[j]), tb.chrom[j].decode(), int(tb.position[j])) == (sid, chrom, position)
        assert tb.ref[j].decode() == tuples[0][0] and tb.alt[j].decode() == tuples[1][0]
        assert tb.cum[j, 0] == tuples[0][1] and tb.cum[j, 1] == tuples[1][1]
    assert open(tmp_path / "deleterious.json").read() == doc["deleterious_json"]
    assert open(tmp_path / "population.fam").read() == doc["population_fam"]
    assert open(tmp_path / "pop_deleterious.txt").read() == doc["pop_deleterious_txt"]
    hdr = pop.vcf_header(st.person_id).split("\n")
    hdr[1] = "##filedate=FROZEN"
    assert "\n".join(hdr) == doc["vcf_header"]
    # the numpy stream is exactly where the reference's worker would start drawing
    state = numpy.random.get_state()
    assert list(state[1]) == doc["mt_state_key"] and state[2] == doc["mt_state_pos"]
    # the sample table the worker sees
    for i, (iid, sex, ctrl, carried) in enumerate(doc["samples"]):
        assert int(st.person_id[i]) == iid
        assert bool(st.flags[i] & 1) == (sex == 1) and bool(st.flags[i] & 2) == ctrl
    per_sample = {}
    for k, sid in enumerate(st.forced_ids):
        for i in st.carrier_idx[int(st.carrier_off[k]): int(st.carrier_off[k + 1])]:
            per_sample.setdefault(int(i), set()).add(int(sid))
    for i, (_, _, ctrl, carried) in enumerate(doc["samples"]):
        assert per_sample.get(i, set()) == set(carried or [])


def test_snps_file_roundtrip_and_layout(pop, tmp_path):
    numpy.random.seed(5)
    random.seed(6)
    tb = pop.generate_snp_table(300, min_maf=0.005)
    path = tmp_path / "snps.json.gz"
    pop.write_snps_file(tb, path)
    lines = gzip.open(path, "rt").read().splitlines()
    assert len(lines) == 300
    # layout of str(SNPTuples), pop_factory.py:118-124
    assert lines[0] == '{"id": %d, "chromosome": "%s", "position": %d, "tuples": {"%s": %r, "%s": 1.0}}' % (
        tb.snp_id[0], tb.chrom[0].decode(), tb.position[0], tb.ref[0].decode(), float(tb.cum[0, 0]), tb.alt[0].decode())
    back = pop.load_snps_file(path)
    for col in ("snp_id", "chrom", "position", "ref", "alt", "n_alleles"):
        assert np.array_equal(getattr(back, col), getattr(tb, col))
    assert np.array_equal(back.cum[:, :2], tb.cum[:, :2])


def test_sorted_by_chromosome_string_then_position(pop):
    numpy.random.seed(9)
    random.seed(9)
    tb = pop.generate_snp_table(5000)
    keys = [(c.decode(), int(p)) for c, p in zip(tb.chrom, tb.position)]
    assert keys == sorted(keys)  # lexicographic on the chromosome string: 1,10,11,...,19,2,20,...
    assert sorted(tb.snp_id) == list(range(1, 5001))
    assert tb.cum[:, 0].min() >= 0.5 and np.all(tb.ref != tb.alt)


def test_deleterious_json_key_quirk(pop, tmp_path):
    g = pop.DeleteriousGroup("g-0", 2)
    g.deleterious = {17: 0.5, 4: 0.5}
    line = g.to_json()
    assert line == '{"deleterious": {"17": 0.5, "4": 0.5}, "name": "g-0", "population_weight": 2}'
    assert list(pop.DeleteriousGroup.from_json(line).deleterious) == [17, 4]
    # reference behaviour (pop_factory.py:613-619): string keys, which never match an int SNP id
    assert list(pop.DeleteriousGroup.from_json(line, int_keys=False).deleterious) == ["17", "4"]


def test_bad_yaml_bounds_raise_like_the_reference(pop, tmp_path):
    numpy.random.seed(1)
    random.seed(1)
    tb = pop.generate_snp_table(50)
    cfg = tmp_path / "bad.yml"
    cfg.write_text("g:\n  mutation_weights: [1]\n  population_weight: 1\n  max_minor_allele_freq: 0.7\n")
    with pytest.raises(Exception, match="max_minor_allele_freq must be between 0 and 0.5"):
        pop.pick_deleterious(tb, str(cfg))
    cfg.write_text("g:\n  mutation_weights: [1]\n  population_weight: 1\n  max_minor_allele_freq: 0.0001\n")
    with pytest.raises(Exception, match="No SNPs in list to choose from"):
        pop.pick_deleterious(tb, str(cfg))


def test_fast_snps_json_helpers_match_python_json(pop, tmp_path):
    """The library's C formatter / parser for snps.json.gz against json.dumps / json.loads, incl. multi-allelic
    rows, a single-tuple row and multi-character alleles; other layouts fall back to the JSON parser."""
    from dna_factory_amd import capi
    from dna_factory_amd.tables import SnpTable

    tb = SnpTable.from_snps([(5, "X", 10, [("A", 0.1), ("T", 0.75), ("G", 1.0)]), (6, "MT", 0, [("C", 1.0)]),
                             (7, "2", 99, [("G", 0.25), ("A", 0.5), ("C", 0.875), ("T", 1.0)]),
                             (8, "1", 3, [("AT", 1 - 0.005 * 7), ("G", 1.0)]), (2**40, "22", 248946339, [("T", 0.505), ("C", 1.0)])])
    blob = capi.format_snps_json(tb.snp_id, tb.chrom, tb.position, tb.ref, tb.alt, tb.n_alleles, tb.cum)
    assert blob.tobytes() == pop._snps_json_lines_python(tb, 0, len(tb))
    path = tmp_path / "s.json.gz"
    pop.write_snps_file(tb, path, threads=2)
    back = pop.load_snps_file(path)
    for col in ("snp_id", "chrom", "position", "ref", "alt", "n_alleles"):
        assert np.array_equal(getattr(back, col), getattr(tb, col)), col
    assert np.array_equal(back.cum, tb.cum)
    # a quote in an allele needs JSON escaping: the C formatter declines, the writer falls back, the reader too
    odd = SnpTable.from_snps([(1, "1", 5, [('A"', 0.5), ("C", 1.0)])])
    assert capi.format_snps_json(odd.snp_id, odd.chrom, odd.position, odd.ref, odd.alt, odd.n_alleles, odd.cum) is None
    pop.write_snps_file(odd, path)
    assert pop.load_snps_file(path).ref[0] == b'A"'
    with gzip.open(path, "wt") as f:  # compact separators: not the fast path's layout
        f.write('{"id":1,"chromosome":"1","position":5,"tuples":{"A":0.5,"C":1.0}}\n')
    assert capi.parse_snps_json(gzip.open(path, "rb").read()) is None
    assert pop.load_snps_file(path).cum[0, 0] == 0.5


def test_random_choice_replay_leaves_python_random_in_step(pop):
    """generate_snp_table draws its per-SNP alternative allele without a Python loop; the values and the state of
    `random` afterwards are those of the loop it replaces (random.choice per SNP, pop_factory.py:190)."""
    for count in (1, 2, 511, 512, 513, 5000):
        random.seed(1234 + count)
        want = [random.choice((0, 1, 2)) for _ in range(count)]
        after = random.getstate()
        random.seed(1234 + count)
        got = pop._replay_random_choice3(count)
        assert [int(x) for x in got] == want
        assert random.getstate() == after


def test_table_slices_carry_the_formatted_columns():
    """SnpTable.slice of a table whose fixed columns were formatted once (the multi-device writer formats the job's table
    once and hands every device views of its tiles): the same bytes and offsets as formatting the slice itself."""
    from dna_factory_amd import genome

    tb = genome.synthetic_snp_table(5000, 0.01, np.random.RandomState(8))
    blob, off = tb.prefixes()
    for lo, hi in ((0, 5000), (0, 1), (17, 2048), (4999, 5000), (1234, 1234)):
        t = tb.slice(lo, hi)
        fresh = genome.synthetic_snp_table(5000, 0.01, np.random.RandomState(8)).slice(lo, hi)
        b1, o1 = t.prefixes()
        b2, o2 = fresh.prefixes()
        assert bytes(b1) == bytes(b2) and list(o1) == list(o2)
        assert (t.chrom_class == fresh.chrom_class).all()


def test_numpy_legacy_choice_replay_is_numpy():
    """dnaf_numpy_legacy_choice (the reference's deleterious picks, pop_factory.py:576, without shuffling the whole id list):
    the same indices as numpy.random.choice(n, k, replace=False) and the same generator state afterwards, from any
    position in the stream, across block boundaries of the generator and powers of two of n."""
    import numpy

    from dna_factory_amd import capi

    rs = np.random.RandomState(5)
    cases = [(1, 1), (2, 1), (2, 2), (3, 2), (10, 3), (7, 7), (623, 2), (624, 5), (625, 3), (1000, 10), (65535, 4), (65536, 5), (65537, 4),
             (1000003, 10), (5000, 32), (5000, 33), (40000, 200)]  # 32 entries: the last the C replay takes; beyond, numpy's own shuffle
    cases += [(int(rs.randint(1, 300000)), int(rs.randint(1, 12))) for _ in range(20)]
    for n, k in cases:
        k = min(k, n)
        for pre in (0, 1, 623, 624, 625, int(rs.randint(0, 3000))):
            seed = int(rs.randint(0, 2**31))
            numpy.random.seed(seed)
            numpy.random.bytes(4 * pre)
            want = numpy.random.choice(n, k, replace=False)
            want_state = numpy.random.get_state()
            follow = numpy.random.rand(3)
            numpy.random.seed(seed)
            numpy.random.bytes(4 * pre)
            got = capi.numpy_legacy_choice(n, k)
            st = numpy.random.get_state()
            assert np.array_equal(got, want), (n, k, pre)
            assert st[2] == want_state[2] and np.array_equal(st[1], want_state[1]), (n, k, pre)
            assert np.array_equal(numpy.random.rand(3), follow)
    with pytest.raises(ValueError):
        capi.numpy_legacy_choice(3, 4)  # numpy's own error for a sample larger than the population


def test_round_robin_writer_keeps_order_and_buffers(tmp_path, monkeypatch):
    """cli._write_round_robin without a GPU: tiles dealt to three "devices" whose generators finish in random order and hand
    over views of FOUR staging buffers used in turn (as the library does); the file must hold header + every tile's members
    in table order, whatever the timing -- a device that fetched a tile before the writer had written the tile four
    before it would overwrite bytes that are still to be written."""
    import gzip
    import random
    import time

    from dna_factory_amd import bgzf, cli, genome, rowgen
    from dna_factory_amd.tables import SampleTable

    tb = genome.synthetic_snp_table(3000, 0.01, np.random.RandomState(2))
    st = SampleTable(np.zeros(8, dtype=np.uint8))

    def tile_text(lo, hi):
        return b"".join(b"row %d of the table\n" % j for j in range(lo, hi)) * 40

    class FakeCtx:
        def __init__(self, rnd):
            self.rnd, self.bufs, self.k, self.lo, self.hi = rnd, [bytearray(1 << 20) for _ in range(4)], 0, 0, 0

        def generate(self, b, m, what):
            time.sleep(self.rnd.random() * 0.001)

        def bgzf_compress(self, reuse_tables=False):
            pass

        def sync(self):
            pass

        def export_bed(self):
            return b"bed[%d,%d)" % (self.lo, self.hi)

        def fetch_bgzf(self):
            text = tile_text(self.lo, self.hi)
            blob = b"".join(bgzf.compress_block(text[i: i + bgzf.BLOCK_INPUT], 1) for i in range(0, len(text), bgzf.BLOCK_INPUT))
            buf = self.bufs[self.k % 4]  # the fourth-next fetch reuses this buffer
            self.k += 1
            buf[: len(blob)] = blob
            return memoryview(buf)[: len(blob)]

    class FakeGen:
        def __init__(self, device=0, seed=0, rng="philox", mt_state=None):
            self.ctx = FakeCtx(random.Random(device))

        def __enter__(self):
            return self

        def __exit__(self, *a):
            for b in self.ctx.bufs:  # a context that is closed too early takes its buffers with it
                b[:] = b"\xff" * len(b)

        def set_samples(self, st):
            pass

        def set_snps(self, tb, rng_row0=0):
            self.ctx.lo, self.ctx.hi = rng_row0, rng_row0 + len(tb)

    monkeypatch.setattr(rowgen, "RowGenerator", FakeGen)
    real_pwrite = bgzf._pwrite_all

    def slow_pwrite(fd, mv, offset):  # a writer slower than the devices: they run ahead as far as the protocol lets them
        time.sleep(0.004)
        real_pwrite(fd, mv, offset)

    monkeypatch.setattr(bgzf, "_pwrite_all", slow_pwrite)
    path, bed = str(tmp_path / "out.vcf.gz"), str(tmp_path / "out.bed")
    done = []
    cli._write_round_robin(path, "##header\n", st, tb, 1, "philox", [0, 1, 2], 6, 100, 2, False, None, bed, done.append)
    want = b"##header\n" + b"".join(tile_text(lo, min(3000, lo + 100)) for lo in range(0, 3000, 100))
    assert gzip.open(path, "rb").read() == want
    assert bgzf.block_sizes(path)[-1] == 28
    assert open(bed, "rb").read() == cli.BED_MAGIC + b"".join(b"bed[%d,%d)" % (lo, lo + 100) for lo in range(0, 3000, 100))
    assert sum(done) == 3000
