"""Parity of the HIP path (through the C ABI) with the CPU oracle and the committed reference fixtures.
Bit-exact: text bytes, row offsets and the packed 2-bit matrix."""
import numpy as np
import pytest

from helpers import golden_inputs, golden_names, load_golden, random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def RowGenerator():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd.rowgen import RowGenerator

    return RowGenerator


def oracle_run(orc, snps, samples, seed, rng_row0=0):
    rec = orc.snp_records(snps)
    sex, ctrl, off, ids = orc.sample_arrays(samples)
    return orc.generate(rec, sex, ctrl, off, ids, rng_mode=orc.RNG_PHILOX, seed=seed, rng_row0=rng_row0)


def check_case(RowGenerator, orc, snps, samples, seed, rng_row0=0):
    with RowGenerator(device=0, seed=seed) as g:
        g.set_samples(samples)
        g.set_snps(snps, rng_row0=rng_row0)
        text, off, packed = g.generate()
        t_only, _, _ = g.generate(text=True, packed=False)
        _, _, p_only = g.generate(text=False, packed=True)
    want_text, want_off, want_packed = oracle_run(orc, snps, samples, seed, rng_row0)
    assert np.array_equal(off, want_off)
    assert text.tobytes() == want_text
    assert np.array_equal(packed, want_packed)
    assert t_only.tobytes() == want_text
    assert np.array_equal(p_only, want_packed)


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.endswith("_mt")])
def test_hip_matches_reference_fixture(RowGenerator, name):
    doc = load_golden(name)
    if doc.get("lines_philox") is None:
        pytest.skip("fixture has no Philox lines")
    snps, samples = golden_inputs(doc)
    with RowGenerator(device=0, seed=doc["seed"]) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        assert g.lines() == doc["lines_philox"]


@pytest.mark.parametrize("name", ["headline_10x20000", "config5_shape_4x50000"])
@pytest.mark.parametrize("form", ["segments", "line_owning"])
def test_long_row_fixtures_in_both_kernel_forms(RowGenerator, monkeypatch, name, form):
    """The reference's own lines at 20,000 and 50,000 samples (BASELINE configs 3 and 5) from both long-row kernel forms: a
    wave per row segment (what a range of a few rows gets by default) and the line-owning form (what a full tile gets)."""
    if form == "segments":
        monkeypatch.setenv("DNAF_FORCE_SEGMENTS", "1")
    else:
        monkeypatch.setenv("DNAF_MIN_RUN_WAVES", "0")
    doc = load_golden(name)
    snps, samples = golden_inputs(doc)
    with RowGenerator(device=0, seed=doc["seed"]) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        assert g.lines() == doc["lines_philox"]


@pytest.mark.parametrize("n_rows,S", [(1, 1), (7, 3), (33, 64), (50, 257), (40, 1000), (12, 4099), (3, 20000)])
def test_hip_matches_oracle_random(RowGenerator, oracle, n_rows, S):
    rs = np.random.RandomState(1000 + n_rows + S)
    snps, samples = random_case(rs, n_rows, S)
    check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)))


def test_seed_and_row_index_use_all_64_bits(RowGenerator, oracle):
    rs = np.random.RandomState(8)
    snps, samples = random_case(rs, 9, 70)
    check_case(RowGenerator, oracle, snps, samples, seed=0xFEDCBA9876543210, rng_row0=(1 << 32) - 4)


def test_shards_reproduce_the_whole_job(RowGenerator, oracle):
    """SNP-range shards (the multi-GPU partition) give the bytes of the unsharded job."""
    rs = np.random.RandomState(11)
    snps, samples = random_case(rs, 90, 150)
    whole_text, whole_off, whole_packed = oracle_run(oracle, snps, samples, 77)
    parts_t, parts_p = [], []
    for b, e in ((0, 31), (31, 32), (32, 90)):
        with RowGenerator(device=0, seed=77) as g:
            g.set_samples(samples)
            g.set_snps(snps[b:e], rng_row0=b)
            t, _, p = g.generate()
            parts_t.append(t.tobytes())
            parts_p.append(p)
    assert b"".join(parts_t) == whole_text
    assert np.array_equal(np.concatenate(parts_p), whole_packed)


def test_row_ranges_of_one_table(RowGenerator, oracle):
    rs = np.random.RandomState(12)
    snps, samples = random_case(rs, 60, 97)
    want_text, want_off, want_packed = oracle_run(oracle, snps, samples, 5)
    with RowGenerator(device=0, seed=5) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        for b, n in ((0, 0), (0, 1), (10, 25), (59, 1), (60, 0)):
            t, off, p = g.generate(b, n)
            assert t.tobytes() == want_text[int(want_off[b]): int(want_off[b + n])]
            assert np.array_equal(off, want_off[b: b + n + 1] - want_off[b])
            assert np.array_equal(p, want_packed[b: b + n])


def test_write_vcf_snps_dropin(RowGenerator, oracle, tmp_path):
    import io

    rs = np.random.RandomState(13)
    snps, samples = random_case(rs, 45, 83)
    want_text, _, _ = oracle_run(oracle, snps, samples, 99)
    with RowGenerator(device=0, seed=99) as g:
        sink = io.BytesIO()
        assert g.write_vcf_snps(samples, snps, sink, tile_rows=7) == 45
        assert sink.getvalue() == want_text
        tsink = io.StringIO()
        g.write_vcf_snps(samples, snps, tsink)
        assert tsink.getvalue() == want_text.decode()


def test_errors_are_reported_not_fatal(RowGenerator):
    from dna_factory_amd import capi

    with RowGenerator(device=0, seed=1) as g:
        with pytest.raises(capi.DnafError):
            g.set_snps([(1, "1", 5, [("A", 0.9), ("T", 1.0)])])  # before set_samples
        g.set_samples([(1, True, None)])
        with pytest.raises(capi.DnafError) as e:
            g.set_snps([(1, "1", 5, [("A", 0.5), ("T", 0.9)])])  # a roll above 0.9 has no allele
        assert e.value.code == capi.ERR_THRESHOLD
        with pytest.raises(capi.DnafError):
            g.set_snps([(1, "1", 5, [("A", 0.1), ("C", 0.2), ("G", 0.3), ("T", 0.4), ("N", 1.0)])])
        g.set_snps([(1, "1", 5, [("A", 0.5), ("T", 1.0)])])
        with pytest.raises(capi.DnafError):
            g.ctx.generate(0, 2)
        with pytest.raises(capi.DnafError):
            g.ctx.set_rng(capi.RNG_MT19937, 2**32)  # numpy.random.seed accepts 0 .. 2^32-1 only
        with pytest.raises(capi.DnafError):
            g.ctx.set_rng(7, 1)


def test_full_size_properties(RowGenerator):
    """BASELINE config shape S=20k: size-independent properties on a tile too big for the scalar oracle:
    text is consistent with the packed matrix, cell grammar holds, and a forced SNP is 1/1 for its carriers."""
    S, n = 20000, 2000
    rs = np.random.RandomState(21)
    from dna_factory_amd.tables import SampleTable, SnpTable

    flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    carriers = np.sort(rs.choice(np.arange(S // 2, S), 500, replace=False)).astype(np.uint32)
    st = SampleTable(flags, [17], [0, len(carriers)], carriers)
    maf = 0.005 * rs.randint(2, 100, n)
    cum = np.ones((n, 4))
    cum[:, 0] = 1 - maf
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"1", dtype="S2"), np.arange(n) * 10 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"G", dtype="S1"), np.full(n, 2, np.uint8), cum)
    with RowGenerator(device=0, seed=4) as g:
        g.set_samples(st)
        g.set_snps(tb)
        text, off, packed = g.generate()
    assert int(off[-1]) == len(text)
    row_len = np.diff(off.astype(np.int64))
    _, poff = tb.prefixes()
    assert np.array_equal(row_len, np.diff(poff.astype(np.int64)) + 4 * S)
    # decode cells of every row from the fixed-width layout
    alt_freq = np.zeros(n)
    for j in sorted(set(range(0, n, 97)) | {16}):
        cells = text[int(off[j]) + int(poff[j + 1] - poff[j]): int(off[j + 1])].reshape(S, 4)
        assert np.all(cells[:, 1] == ord("/")) and np.all(cells[:-1, 3] == 9) and cells[-1, 3] == 10
        a, b = cells[:, 0] - 48, cells[:, 2] - 48
        assert set(np.unique(a)) <= {0, 1} and set(np.unique(b)) <= {0, 1}
        bits = (packed[j][:, None] >> (2 * np.arange(16))[None, :]).reshape(-1)[:S] & 3
        assert np.array_equal(bits, a | (b << 1))
        if j == 16:
            assert np.all(a[carriers] == 1) and np.all(b[carriers] == 1)
        ctrl = slice(0, S // 2)
        alt_freq[j] = (a[ctrl].sum() + b[ctrl].sum()) / S
        assert abs(alt_freq[j] - maf[j]) < 6 * np.sqrt(maf[j] * (1 - maf[j]) / S) + 1e-9


def test_tile_larger_than_4_gib(RowGenerator, oracle):
    """One launch whose text exceeds 2^32 bytes: the rows at the far end (64-bit offsets) still match the oracle."""
    from dna_factory_amd.tables import SampleTable, SnpTable

    S, n = 20000, 56000  # 56,000 rows x ~80 KB = 4.5 GB
    rs = np.random.RandomState(77)
    flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    st = SampleTable(flags)
    maf = 0.005 * rs.randint(2, 100, n)
    cum = np.ones((n, 4))
    cum[:, 0] = 1 - maf
    chrom = np.where(np.arange(n) % 50 == 7, b"X", b"3").astype("S2")
    tb = SnpTable(np.arange(1, n + 1), chrom, np.arange(n) * 10 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"G", dtype="S1"), np.full(n, 2, np.uint8), cum)
    sex, ctrl, doff, dids = oracle.sample_arrays_from_table(st.flags, st.forced_ids, st.carrier_off, st.carrier_idx)
    with RowGenerator(device=0, seed=9) as g:
        g.set_samples(st)
        g.set_snps(tb)
        g.ctx.generate(0, n, 1 | 2)
        off = g.ctx.row_offsets(0, n)
        assert int(off[n]) > 2**32
        packed_tail = None
        for b in (0, n // 2, n - 48):
            text = g.ctx.read_text_range(int(off[b]), int(off[b + 48] - off[b]))
            want, _, want_packed = oracle.generate(oracle.table_records(tb, b, b + 48), sex, ctrl, doff, dids, seed=9, rng_row0=b)
            assert text.tobytes() == want
        packed = g.ctx.read_packed(n, g.words_per_row)
        assert np.array_equal(packed[n - 48:], want_packed)


@pytest.mark.parametrize("S,n_rows", [(1_000_003, 6), (4, 300_000)])
def test_extreme_shapes(RowGenerator, oracle, S, n_rows):
    """Very wide rows (a million samples: many segments per row, 4 MB lines) and very many narrow rows."""
    from dna_factory_amd.tables import SampleTable, SnpTable

    rs = np.random.RandomState(S % 1000)
    flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    st = SampleTable(flags)
    cum = np.ones((n_rows, 4))
    cum[:, 0] = 1 - 0.005 * rs.randint(1, 100, n_rows)
    chrom = np.array([b"7", b"X", b"Y", b"MT"])[np.arange(n_rows) % 4].astype("S2")
    tb = SnpTable(np.arange(1, n_rows + 1), chrom, np.arange(n_rows) + 1, np.full(n_rows, b"C", dtype="S1"),
                  np.full(n_rows, b"T", dtype="S1"), np.full(n_rows, 2, np.uint8), cum)
    sex, ctrl, doff, dids = oracle.sample_arrays_from_table(st.flags, st.forced_ids, st.carrier_off, st.carrier_idx)
    with RowGenerator(device=0, seed=3) as g:
        g.set_samples(st)
        g.set_snps(tb)
        text, off, packed = g.generate()
    bad = oracle.verify_threads(oracle.table_records(tb), sex, ctrl, doff, dids, 3, 8, text, off, packed)
    assert bad is None, "row %s differs" % bad


@pytest.mark.parametrize("n_rows,S", [(5, 1), (9, 5), (20, 16), (33, 17), (12, 4099)])
def test_bed_export_matches_the_text(RowGenerator, oracle, n_rows, S):
    """dnaf_export_bed: PLINK 1 .bed rows from the packed matrix equal what the ORACLE's VCF text says cell by cell (A1 =
    ALT: 0/0 -> 3, one non-reference allele -> 2, two -> 0; haploid calls homozygous; '.' -> 1), including
    multi-allelic rows (any non-reference allele counts) and every chromosome class; padding bits are zero.  (plink
    itself is not in the image: parity with its own VCF import is unpinned; these are its documented rules.)"""
    rs = np.random.RandomState(77 + n_rows + S)
    snps, samples = random_case(rs, n_rows, S)
    seed = int(rs.randint(0, 2**31))
    want_text, want_off, _ = oracle_run(oracle, snps, samples, seed)
    lines = [want_text[int(want_off[j]): int(want_off[j + 1])].decode("ascii") for j in range(n_rows)]
    with RowGenerator(device=0, seed=seed) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        g.ctx.generate(0, n_rows, 3)
        bed = g.ctx.export_bed().reshape(n_rows, (S + 3) // 4)
    for j, line in enumerate(lines):
        codes = (bed[j][:, None] >> (2 * np.arange(4))[None, :]).reshape(-1) & 3
        assert not codes[S:].any()
        for i, cell in enumerate(line.rstrip("\n").split("\t")[9:]):
            if cell == ".":
                want = 1
            elif "/" in cell:
                a, b = cell.split("/")
                want = {0: 3, 1: 2, 2: 0}[(a != "0") + (b != "0")]
            else:
                want = 3 if cell == "0" else 0
            assert codes[i] == want, (j, i, cell)


def test_reserved_buffers_and_a_large_tile_stay_exact(RowGenerator, oracle):
    """dnaf_reserve_output sizes both buffers up front (dense word count in, padded device pitch inside); a 1.1 GB tile
    generated into them is the oracle's at both ends, and the reservation is not re-allocated by the launch."""
    from dna_factory_amd.tables import SampleTable, SnpTable

    S, n = 20000, 14000
    rs = np.random.RandomState(31)
    flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    st = SampleTable(flags)
    cum = np.ones((n, 4))
    cum[:, 0] = 1 - 0.005 * rs.randint(2, 100, n)
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"4", dtype="S2"), np.arange(n) * 7 + 1, np.full(n, b"G", dtype="S1"),
                  np.full(n, b"A", dtype="S1"), np.full(n, 2, np.uint8), cum)
    sex, ctrl, doff, dids = oracle.sample_arrays_from_table(st.flags, st.forced_ids, st.carrier_off, st.carrier_idx)
    with RowGenerator(device=0, seed=44) as g:
        g.set_samples(st)
        g.set_snps(tb)
        assert g.ctx.text_bytes(0, n) > 1 << 30
        assert g.ctx.packed_pitch_words() == 1280 and g.words_per_row == 1250  # whole 128-byte lines per row
        g.ctx.reserve_output(g.ctx.text_bytes(0, n), n * g.words_per_row)
        g.ctx.generate(0, 1, 3)
        before = g.ctx.device_buffers()
        g.ctx.generate(0, n, 3)
        after = g.ctx.device_buffers()
        assert (before[0], before[2]) == (after[0], after[2]) and after[3] == n * 1280
        off = g.ctx.row_offsets(0, n)
        for b in (0, n - 40):
            text = g.ctx.read_text_range(int(off[b]), int(off[b + 40] - off[b]))
            want, _, want_packed = oracle.generate(oracle.table_records(tb, b, b + 40), sex, ctrl, doff, dids, seed=44, rng_row0=b)
            assert text.tobytes() == want
        assert np.array_equal(g.ctx.read_packed(n, g.words_per_row)[n - 40:], want_packed)


@pytest.mark.parametrize("rpw", [1, 2, 3, 8, 64])
def test_line_owning_kernel_fuzz(RowGenerator, oracle, monkeypatch, rpw):
    """The line-owning form of the kernel (rowrun_kernels.hip.h: a wave stores whole 128-byte lines, i.e. its rows
    minus the first bytes of its first row plus the first bytes of the next wave's first row), forced on for small
    tables and for 1 .. 64 rows per wave: every chromosome class, forced cells, 1-4 alleles, sample counts around the
    staging ring's boundaries (1024 / 2048 groups) and the padded / dense pitch switch (W = 128 words at S = 2033)."""
    monkeypatch.setenv("DNAF_MIN_RUN_WAVES", "0")
    monkeypatch.setenv("DNAF_ROWS_PER_WAVE", str(rpw))
    rs = np.random.RandomState(777 + rpw)
    pools = [range(1021, 1100), range(2029, 2052), range(4090, 4104), range(8186, 8200), range(12280, 12300), range(20000, 20004)]
    for case in range(18):
        S = int(rs.choice(list(pools[case % len(pools)])))
        n_rows = int(rs.randint(1, 12 if S > 4000 else 70))
        snps, samples = random_case(rs, n_rows, S, p_sex_chrom=0.4)
        check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=int(rs.randint(0, 1000)))


@pytest.mark.parametrize("rpw", [1, 3, 64])
def test_line_owning_kernel_on_dense_rows(RowGenerator, oracle, monkeypatch, rpw):
    """rowrun_kernel's dense (unpadded-pitch) path: packed rows shorter than 128 words (1,021 .. 2,032 samples) normally go
    to the short-row kernel, so this path -- whole rows per wave, head / skip line ownership over a dense packed matrix --
    only runs with DNAF_NO_ROWMIX (the library's switch for A/B runs); it ships, so it is tested."""
    monkeypatch.setenv("DNAF_NO_ROWMIX", "1")
    monkeypatch.setenv("DNAF_MIN_RUN_WAVES", "0")
    monkeypatch.setenv("DNAF_ROWS_PER_WAVE", str(rpw))
    rs = np.random.RandomState(31 + rpw)
    for case in range(10):
        S = int(rs.randint(1021, 2033))
        snps, samples = random_case(rs, int(rs.randint(1, 70)), S, p_sex_chrom=0.4)
        with RowGenerator(device=0, seed=1) as probe:  # the switch took: this shape would otherwise run rowmix_kernel
            probe.set_samples(samples)
            probe.set_snps(snps)
            probe.ctx.generate(0, len(snps), 3)
            assert probe.ctx.last_kernel() == "rowrun_kernel"
        check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=int(rs.randint(0, 1000)))


@pytest.mark.parametrize("rpw", [1, 5, 64])
def test_short_row_kernel_fuzz(RowGenerator, oracle, monkeypatch, rpw):
    """The short-row form (rowmix_kernels.hip.h: the lanes of a trip span the rows of a wave's run) at 1 .. 64 rows per
    wave: sample counts from 1 to 2,032 (one group per row, rows shorter and longer than a trip, last groups of 1-4
    samples, pad bytes in the packed rows), stretches of ordinary rows broken by X / Y / MT, multi-allelic and forced
    rows."""
    monkeypatch.setenv("DNAF_ROWS_PER_WAVE", str(rpw))
    if rpw != 1:  # X rows in the lane-spanning trips at every sample count (by default only up to 640 samples)
        monkeypatch.setenv("DNAF_MIX_X_MAX_GROUPS", "508")
    rs = np.random.RandomState(4000 + rpw)
    pools = [range(1, 9), range(9, 70), range(190, 270), range(500, 530), range(1000, 1030), range(2017, 2033)]
    for case in range(24):
        S = int(rs.choice(list(pools[case % len(pools)])))
        n_rows = int(rs.randint(1, 200 if S < 300 else 90))
        snps, samples = random_case(rs, n_rows, S, p_sex_chrom=0.15 if case % 2 else 0.4, n_del=3 if case % 3 else 0)
        check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=int(rs.randint(0, 1000)))


@pytest.mark.parametrize("S", [1, 3, 6, 50, 199, 200, 201, 202, 1001, 2032])
def test_runs_of_x_y_mt_rows_in_the_short_row_form(RowGenerator, oracle, monkeypatch, S):
    """A sorted table holds its X, Y and MT rows in blocks: whole runs of them go through the lane-spanning trips of
    rowmix_kernel with the cells' places and widths from the per-group table (all-male, all-female and mixed sample tables;
    sample counts with every size of last group; autosome rows, multi-allelic and forced rows break the runs).
    The row-at-a-time path (DNAF_NO_MIX_X=1) writes the same bytes."""
    rs = np.random.RandomState(9000 + S)
    n_rows = 260 if S < 300 else 40
    snps, samples = random_case(rs, n_rows, S, p_sex_chrom=0.0, n_del=2)
    snps = [(i, ("X", "X", "Y", "MT", "X", "Y")[(k // 9) % 6] if k % 37 != 5 else ch, pos, tu) for k, (i, ch, pos, tu) in enumerate(snps)]
    for sexes in ("mixed", "male", "female"):
        sm = [(sex if sexes == "mixed" else (1 if sexes == "male" else 2), ctrl, car) for (sex, ctrl, car) in samples]
        outs = []
        monkeypatch.setenv("DNAF_MIX_X_MAX_GROUPS", "508")  # (the library's own limit is lower: long rows gain nothing)
        for no_x in ("0", "1"):
            monkeypatch.setenv("DNAF_NO_MIX_X", no_x)
            with RowGenerator(device=0, seed=77 + S) as g:
                g.set_samples(sm)
                g.set_snps(snps)
                t, off, p = g.generate()
                assert g.ctx.last_kernel() == "rowmix_kernel"
                outs.append((t.tobytes(), off.tobytes(), p.tobytes()))
        assert outs[0] == outs[1], sexes
        want_text, _, want_packed = oracle_run(oracle, snps, sm, 77 + S)
        assert outs[0][0] == want_text and outs[0][2] == want_packed.tobytes(), sexes


@pytest.mark.parametrize("share_long", [0.1, 1.0])
def test_short_rows_with_long_fixed_columns(RowGenerator, oracle, share_long):
    """Rows whose fixed columns are longer than 64 bytes (15-letter alleles -- the oracle's records hold no longer ones -- and
    13-digit ids): the library then keeps no fixed-stride copy
    of them (rowmix_kernel falls back to the run's piece of the prefix blob in LDS), and when a run's fixed columns exceed
    that cache too (every row long) they are copied from global memory ahead of the trips."""
    rs = np.random.RandomState(int(1000 * share_long) + 77)
    snps, samples = random_case(rs, 260, 120, p_sex_chrom=0.1, n_del=2)
    long_a, long_b = "ACGTACGTACGTACG", "TTGCATTGCATTGCA"
    snps = [(i + 10**12, ch, pos, [(long_a if k == 0 else long_b if k == 1 else a, c) for k, (a, c) in enumerate(tu)])
            if rs.rand() < share_long else (i, ch, pos, tu) for (i, ch, pos, tu) in snps]
    samples = [(sex, ctrl, None if car is None else [c + 10**12 if any(c + 10**12 == sn[0] for sn in snps) else c for c in car])
               for (sex, ctrl, car) in samples]
    check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)))


def test_short_rows_with_fixed_columns_of_49_to_64_bytes(RowGenerator, oracle):
    """13-digit ids with ordinary alleles: the fixed-stride copy of the fixed columns then has a stride of 64 bytes and a
    wave's run is capped at 48 rows so that they fit their LDS cache."""
    rs = np.random.RandomState(4964)
    snps, samples = random_case(rs, 300, 90, p_sex_chrom=0.1, n_del=0)
    snps = [(i + 10**12, ch, 200000000 + pos % 49000000, tu) for (i, ch, pos, tu) in snps]
    from dna_factory_amd.tables import SnpTable

    po = np.asarray(SnpTable.from_snps(snps).prefixes()[1])
    assert 48 < int((po[1:] - po[:-1]).max()) <= 64
    check_case(RowGenerator, oracle, snps, samples, seed=31)


@pytest.mark.parametrize("S", [2029, 2032, 2033, 2036])
def test_shapes_around_the_dense_pitch_limit(RowGenerator, oracle, S):
    """2,032 samples = 127 packed words per row is the last dense pitch (short-row form); from 2,033 on the rows are padded
    to 128-byte lines (line-owning form)."""
    rs = np.random.RandomState(S)
    snps, samples = random_case(rs, 150, S, p_sex_chrom=0.2)
    check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=5)


def test_short_row_and_segment_forms_write_the_same_bytes(RowGenerator, oracle, monkeypatch):
    rs = np.random.RandomState(6060)
    snps, samples = random_case(rs, 700, 200, p_sex_chrom=0.2)
    outs = []
    for no_mix in ("1", "0"):
        monkeypatch.setenv("DNAF_NO_ROWMIX", no_mix)
        monkeypatch.setenv("DNAF_ROWS_PER_WAVE", "41")
        with RowGenerator(device=0, seed=21) as g:
            g.set_samples(samples)
            g.set_snps(snps)
            t, off, p = g.generate()
            outs.append((t.tobytes(), off.tobytes(), p.tobytes()))
    assert outs[0] == outs[1]
    want_text, _, want_packed = oracle_run(oracle, snps, samples, 21)
    assert outs[0][0] == want_text and outs[0][2] == want_packed.tobytes()


def test_both_kernel_forms_write_the_same_bytes(RowGenerator, oracle, monkeypatch):
    """DNAF_FORCE_SEGMENTS=1 (a wave per row segment) and the default line-owning form on one mid-size table."""
    rs = np.random.RandomState(5150)
    snps, samples = random_case(rs, 300, 3000, p_sex_chrom=0.5)
    outs = []
    for force in ("1", "0"):
        monkeypatch.setenv("DNAF_FORCE_SEGMENTS", force)
        monkeypatch.setenv("DNAF_MIN_RUN_WAVES", "0")
        with RowGenerator(device=0, seed=12) as g:
            g.set_samples(samples)
            g.set_snps(snps)
            t, off, p = g.generate()
            outs.append((t.tobytes(), off.tobytes(), p.tobytes()))
    assert outs[0] == outs[1]
    want_text, _, want_packed = oracle_run(oracle, snps, samples, 12)
    assert outs[0][0] == want_text and outs[0][2] == want_packed.tobytes()


def test_hip_matches_oracle_fuzz(RowGenerator, oracle):
    """Eighty random cases around the kernel's internal boundaries: one trip (<= 256 samples), the whole-trip loop vs
    its tail, one segment = 2048 groups = 8192 samples, two and three segments; every chromosome class, forced cells,
    1-4 alleles (helpers.random_case).  Text, offsets and packed matrix are the oracle's."""
    rs = np.random.RandomState(4242)
    pools = [range(1, 70), range(250, 262), range(1000, 1100), range(8186, 8200), range(16378, 16392), range(20000, 20004)]
    for case in range(80):
        S = int(rs.choice(list(pools[case % len(pools)])))
        n_rows = int(rs.randint(1, 10 if S > 4000 else 40))
        snps, samples = random_case(rs, n_rows, S)
        check_case(RowGenerator, oracle, snps, samples, seed=int(rs.randint(0, 2**31)), rng_row0=int(rs.randint(0, 1000)))
