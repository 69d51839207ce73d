"""BGZF: the host writer, the encoder's Huffman / header / CRC tables (CPU), and the GPU encoder itself (inflate
its output with zlib: any conforming reader must return the input and accept CRC-32 / ISIZE)."""
import gzip
import io
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import random_case


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import capi

    return capi


def test_host_writer_roundtrip(capi, tmp_path):
    from dna_factory_amd import bgzf

    data = (b"0/0\t0/1\t1/0\t0/0\t" * 30000) + os.urandom(150000) + b"tail\n"
    path = tmp_path / "t.vcf.gz"
    with bgzf.BgzfWriter(str(path), compresslevel=5, threads=3) as w:
        w.write(data[:7])
        w.write(np.frombuffer(data[7:200000], dtype=np.uint8))
        w.write(data[200000:].decode("latin1").encode("latin1"))
    assert gzip.open(path, "rb").read() == data
    sizes = bgzf.block_sizes(str(path))
    assert sizes[-1] == 28 and max(sizes) <= 65536 and len(sizes) >= len(data) // 65280 + 1


def _tables(capi, lit_hist, match_hist, n_eob):
    import ctypes

    lit = np.asarray(lit_hist, dtype=np.uint64)
    mat = np.asarray(match_hist, dtype=np.uint64)
    size = ctypes.c_uint64(0)
    capi.lib.dnaf_bgzf_tables_from_histogram(capi.ptr(lit), capi.ptr(mat), n_eob, None, 0, ctypes.byref(size))
    blob = np.zeros(size.value // 4, dtype=np.uint32)
    rc = capi.lib.dnaf_bgzf_tables_from_histogram(capi.ptr(lit), capi.ptr(mat), n_eob, capi.ptr(blob), size.value, ctypes.byref(size))
    assert rc == 0
    t = {"lit": blob[0:256], "match": blob[256:512], "eob": int(blob[512]), "hdr_nbits": int(blob[513]),
         "hdr": blob[514:562], "crc_table": blob[562:562 + 1024].reshape(4, 256),
         "shift_chunk": blob[1586:2610], "shift_byte": blob[2610:2866]}
    assert len(blob) == 2866
    return t


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put_entry(self, e):
        self.v |= (int(e) >> 5) << self.n
        self.n += int(e) & 31

    def put_words(self, words, nbits):
        for i, w in enumerate(words):
            take = min(32, nbits - 32 * i)
            if take <= 0:
                break
            self.v |= (int(w) & ((1 << take) - 1)) << self.n
            self.n += take

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def test_encoder_tables_make_a_valid_deflate_stream(capi):
    rs = np.random.RandomState(4)
    lit = np.zeros(256, dtype=np.uint64)
    for ch, n in ((ord("0"), 5000), (ord("1"), 900), (ord("/"), 300), (ord("\t"), 300), (ord("\n"), 3)):
        lit[ch] = n
    mat = np.zeros(256, dtype=np.uint64)
    mat[[0, 1, 5, 30, 100, 249]] = [400, 300, 200, 100, 50, 900]
    t = _tables(capi, lit, mat, 10)
    # a token stream using every kind of entry: literals (frequent, rare and never-seen bytes) and all match lengths
    data = bytearray(b"0/0\t")
    bits = Bits()
    bits.put_words(t["hdr"], t["hdr_nbits"])
    for b in data:
        bits.put_entry(t["lit"][b])
    for L in list(range(3, 259)) + [3, 258, 17]:
        for _ in range(L):
            data.append(data[-4])
        bits.put_entry(t["match"][L - 3])
        for b in (rs.randint(0, 256), ord("1"), 0, 255):
            data.append(int(b))
            bits.put_entry(t["lit"][int(b)])
    bits.put_entry(t["eob"])
    assert zlib.decompressobj(-15).decompress(bits.bytes()) == bytes(data)
    # every literal has a code of at most 15 bits; frequent symbols are short
    assert all(1 <= (int(e) & 31) <= 15 for e in t["lit"]) and (int(t["lit"][ord("0")]) & 31) <= 3


def test_encoder_crc_constants(capi):
    t = _tables(capi, np.ones(256, np.uint64), np.ones(256, np.uint64), 1)
    tab = t["crc_table"]

    def crc_bytes(b):  # table-driven, init ~0, final ~0
        c = 0xFFFFFFFF
        for x in b:
            c = int(tab[0][(c ^ x) & 0xFF]) ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    def mulmod(a, b):
        m, r = 1 << 31, 0
        for _ in range(32):
            if a & m:
                r ^= b
            m >>= 1
            b = (b >> 1) ^ 0xEDB88320 if b & 1 else b >> 1
        return r

    a, b = os.urandom(64 * 3), os.urandom(64 * 2 + 17)
    assert crc_bytes(a) == zlib.crc32(a)
    # crc(A || B) = crc(A) * x^(8|B|) + crc(B), with |B| = 64*2 + 17 split into the chunk and byte tables
    sh = mulmod(int(t["shift_chunk"][2]), int(t["shift_byte"][17]))
    assert mulmod(sh, zlib.crc32(a)) ^ zlib.crc32(b) == zlib.crc32(a + b)
    # slicing tables: one dword step equals four byte steps
    c0 = 0x12345678
    v = c0 ^ int.from_bytes(b"\x01\x02\x03\x04", "little")
    c4 = int(tab[3][v & 0xFF]) ^ int(tab[2][(v >> 8) & 0xFF]) ^ int(tab[1][(v >> 16) & 0xFF]) ^ int(tab[0][v >> 24])
    c = c0
    for x in b"\x01\x02\x03\x04":
        c = int(tab[0][(c ^ x) & 0xFF]) ^ (c >> 8)
    assert c == c4


def _members(blob):
    out, pos = [], 0
    while pos < len(blob):
        magic, xlen, si1, si2, slen, bsize = struct.unpack_from("<4sxxxxxxHBBHH", blob, pos)
        assert magic == b"\x1f\x8b\x08\x04" and (xlen, si1, si2, slen) == (6, 0x42, 0x43, 2)
        out.append(blob[pos: pos + bsize + 1])
        pos += bsize + 1
    assert pos == len(blob)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,S", [(1, 1), (3, 0), (40, 333), (900, 2000), (64, 20000), (16, 61440 // 4)])
def test_gpu_bgzf_inflates_to_the_text(capi, oracle, n_rows, S):
    """What the members must inflate to is the ORACLE's text (the CPU restatement of the reference worker), not the
    text the same GPU run produced."""
    from dna_factory_amd.rowgen import RowGenerator

    rs = np.random.RandomState(n_rows + S)
    snps, samples = random_case(rs, n_rows, S)
    with RowGenerator(device=0, seed=3) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        text, _, _ = g.generate(text=True, packed=False)
        blob, n_text = g.generate_bgzf()
        blob2, _ = g.generate_bgzf(reuse_tables=True)
    sex, ctrl, off, ids = oracle.sample_arrays(samples)
    want_text, _, _ = oracle.generate(oracle.snp_records(snps), sex, ctrl, off, ids, rng_mode=oracle.RNG_PHILOX, seed=3)
    assert text.tobytes() == want_text
    raw = blob.tobytes()
    assert n_text == len(text)
    assert gzip.decompress(raw) == text.tobytes()  # checks every member's CRC-32 and ISIZE too
    assert gzip.decompress(blob2.tobytes()) == text.tobytes()
    members = _members(raw)
    assert len(members) == max(1, -(-len(text) // 61440)) and max(len(m) for m in members) <= 65536
    if S >= 2000:
        assert len(raw) * 4 < len(text)  # even on this worst-case mix (MAF uniform to 0.5, 30 % sex chromosomes)
    # a BGZF reader sees a valid file once the EOF block is appended
    from dna_factory_amd import bgzf

    with gzip.open(io.BytesIO(raw + bgzf.EOF_BLOCK), "rb") as f:
        assert f.read() == text.tobytes()


@pytest.mark.gpu
def test_gpu_bgzf_stored_fallback_on_incompressible_text(capi):
    """Rows with S = 0 are just their fixed columns: random prefix bytes make an incompressible stream, which must
    come out as stored blocks (still valid members)."""
    from dna_factory_amd import capi as c
    from dna_factory_amd.rowgen import RowGenerator

    rs = np.random.RandomState(1)
    n = 1500
    plen = rs.randint(100, 200, n)
    off = np.concatenate([[0], np.cumsum(plen)]).astype(np.uint64)
    blobp = rs.randint(0, 256, int(off[-1])).astype(np.uint8)
    cum = np.ones((n, 4))
    with RowGenerator(device=0, seed=3) as g:
        g.ctx.set_samples(np.zeros(0, np.uint8))
        g.ctx.set_forced(np.zeros(0, np.int64), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
        g.ctx.set_snps(np.arange(n), np.zeros(n, np.uint8), np.full(n, 2, np.uint8), cum, blobp, off)
        g.ctx.generate(0, n, c.OUT_TEXT)
        text = g.ctx.read_text(g.ctx.text_bytes(0, n)).tobytes()
        raw = g.ctx.read_bgzf(g.ctx.bgzf_compress()).tobytes()
    assert gzip.decompress(raw) == text
    members = _members(raw)
    full = [m for m in members if len(m) == 61440 + 5 + 26]
    assert len(full) >= 2 and all(m[18] == 1 for m in full)  # BFINAL=1, BTYPE=00


def _gtz_tables(capi, hist):
    import ctypes

    h = np.ascontiguousarray(hist, dtype=np.uint32)
    assert h.shape == (8, 320)
    size = ctypes.c_uint64(0)
    capi.lib.dnaf_gtz_tables_from_histogram(capi.ptr(h), None, 0, ctypes.byref(size))
    blob = np.zeros(size.value // 4, dtype=np.uint32)
    assert capi.lib.dnaf_gtz_tables_from_histogram(capi.ptr(h), capi.ptr(blob), size.value, ctypes.byref(size)) == 0
    # lit, len, dist, eob + hdr_nbits, hdr, gap table (264), len4 (72), unit_ok
    per = 256 + 256 + 32 + 2 + 64 + 264 + 72 + 1
    classes = []
    for c in range(8):
        b = blob[c * per: (c + 1) * per]
        classes.append({"lit": b[0:256], "len": b[256:512], "dist": b[512:544], "eob": int(b[544]), "hdr_nbits": int(b[545]),
                        "hdr": b[546:610], "gap": b[610:874], "len4": b[874:946], "unit_ok": int(b[946])})
    rest = blob[8 * per:]
    t = {"cls": classes, "crc_table": rest[0:256], "crc_base": rest[256:321], "unit_a": rest[321:385], "unit_b": rest[385:449],
         "shift_chunk": rest[449:1473], "shift_byte": rest[1473:1729], "crc_nib": rest[1729:2241].reshape(32, 16)}
    assert len(rest) == 2241
    return t


def _dist_symbol(d):
    t = d - 1
    if t < 4:
        return t, 0, 0
    hb = t.bit_length() - 1
    return 2 * hb + ((t >> (hb - 1)) & 1), hb - 1, t & ((1 << (hb - 1)) - 1)


def test_gtz_tables_make_valid_deflate_streams(capi):
    """Per-class code tables + run-length coded block header of the genotype-domain encoder: a token stream with
    literals, every match length and distances from every distance symbol inflates to the data it describes."""
    rs = np.random.RandomState(9)
    hist = np.zeros((8, 320), dtype=np.uint32)
    for c in range(8):
        hist[c, ord("0")] = 40000 >> c
        hist[c, ord("1")] = 300 << c
        hist[c, ord("/")] = hist[c, ord("\t")] = 500 + 100 * c
        hist[c, 256] = 10
        hist[c, 257:286] = rs.randint(0, 2000, 29) if c != 3 else 0  # class 3: no match ever seen in the sample
        hist[c, 286:316] = rs.randint(0, 500, 30) if c != 3 else 0
    t = _gtz_tables(capi, hist)
    for c in range(8):
        k = t["cls"][c]
        data = bytearray(rs.randint(0, 256, 40000).astype(np.uint8).tobytes())  # history for far matches
        bits = Bits()
        bits.put_words(k["hdr"], k["hdr_nbits"])
        for b in data:
            bits.put_entry(k["lit"][b])
        for L in list(range(3, 259)) + [20, 24, 256]:
            d = int(rs.choice([1, 2, 3, 4, 5, 8, 12, 100, 1000, 4096, 4097, 20000, 32768]))
            sym, eb, ev = _dist_symbol(d)
            for _ in range(L):
                data.append(data[-d])
            bits.put_entry(k["len"][L - 3])
            bits.put_entry(k["dist"][sym])
            bits.v |= ev << bits.n
            bits.n += eb
            data.append(ord("1"))
            bits.put_entry(k["lit"][ord("1")])
        # the walk's units (gtz_tokens.hip.h): a gap of L repeating bytes is one table entry -- a match at distance 4 for L >= 3,
        # the '/' literal for L = 1 (the tab behind an allele b is taken from the literal table), nothing for L = 0 -- and a
        # far match of f cells takes its length code from len4[f]
        assert k["unit_ok"] == 1
        for half, sep in ((0, ord("/")),):
            for L in range(0, 257):
                if L == 2:
                    continue
                entry = int(k["gap"][L])
                assert (entry & 31) + max(int(k["lit"][ord("0")]) & 31, int(k["lit"][ord("1")]) & 31) <= 32
                bits.put_entry(entry)
                if L == 1:
                    data.append(sep)
                else:
                    for _ in range(L):
                        data.append(data[-4])
                assert (L == 0) == (entry == 0)
                data.append(ord("0"))
                bits.put_entry(k["lit"][ord("0")])
        for f in range(1, 65):
            assert int(k["len4"][f]) == int(k["len"][4 * f - 3])
        bits.put_entry(k["eob"])
        assert zlib.decompressobj(-15).decompress(bits.bytes()) == bytes(data), "class %d" % c
        assert k["hdr_nbits"] < 700  # run-length coded: far below the 1338 bits of a flat 4-bit listing


def test_gtz_crc_of_a_chunk_from_its_cells(capi):
    """CRC-32 is affine: the CRC of n cells is the CRC of n x "0/0\\t" XOR one unit per '1' allele, indexed by the
    cell's distance from the end of the chunk."""
    t = _gtz_tables(capi, np.ones((8, 320), dtype=np.uint32))
    rs = np.random.RandomState(2)
    for n in (1, 2, 17, 63, 64):
        codes = rs.randint(0, 4, n)
        text = b"".join(b"%d/%d\t" % (c & 1, c >> 1) for c in codes)
        crc = int(t["crc_base"][n])
        for e in range(n):
            c = codes[n - 1 - e]
            if c & 1:
                crc ^= int(t["unit_a"][e])
            if c & 2:
                crc ^= int(t["unit_b"][e])
        assert crc == zlib.crc32(text)
        # the same from the table by pairs of cells, as the kernel does it: cells shifted so that the last one is cell 63
        v = 0
        for j, c in enumerate(codes):
            v |= int(c) << (2 * (64 - n + j))
        crc2 = int(t["crc_base"][n])
        for k in range(32):
            crc2 ^= int(t["crc_nib"][31 - k][(v >> (4 * k)) & 15])
        assert crc2 == zlib.crc32(text)
    assert int(t["crc_base"][0]) == 0


# ---- genotype-domain members (gtz_kernels.hip.h): made from the packed matrix, checked like the others ----
def _sorted_case(S, n, seed, min_maf=0.01):
    from dna_factory_amd import genome
    from dna_factory_amd.tables import SampleTable

    rs = np.random.RandomState(seed)
    tb = genome.synthetic_snp_table(n, min_maf=min_maf, rs=rs)
    flags = (rs.rand(S) <= 0.5).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
    return tb, SampleTable(flags)


@pytest.mark.gpu
@pytest.mark.parametrize("S,n", [(2, 9000), (3, 7000), (17, 3000), (700, 400), (950, 300), (2000, 300), (15360, 40),
                                 (20000, 120), (20001, 60), (100003, 12)])
def test_gtz_members_inflate_to_the_text(capi, oracle, S, n):
    """Rows sorted as the CLI sorts them (autosomes, then MT / X / Y): most members are made from the genotype
    matrix, the ones that touch other chromosome classes from the text; every one must inflate to the text -- the
    ORACLE's text for these rows, not this run's own."""
    from dna_factory_amd.rowgen import RowGenerator

    tb, st = _sorted_case(S, n, seed=S + n)
    with RowGenerator(device=0, seed=11) as g:
        g.set_samples(st)
        g.set_snps(tb)
        blob, n_text = g.generate_bgzf()
        blob_again, _ = g.generate_bgzf(reuse_tables=True)
        blob_text, _ = g.generate_bgzf(text_domain=True)
    sex, ctrl, doff, dids = oracle.sample_arrays_from_table(st.flags, st.forced_ids, st.carrier_off, st.carrier_idx)
    want, _, _ = oracle.generate(oracle.table_records(tb), sex, ctrl, doff, dids, seed=11)
    assert n_text == len(want)
    for b in (blob, blob_again, blob_text):
        assert gzip.decompress(b.tobytes()) == want
    members = _members(blob.tobytes())
    assert len(members) == -(-len(want) // 61440) and max(len(m) for m in members) <= 65536
    if S >= 2000:
        assert len(blob) < 0.85 * len(blob_text)  # LZ77 matches over cells beat the distance-4-only coder


@pytest.mark.gpu
def test_gtz_row_ranges_forced_cells_and_multiallelic_rows(capi):
    """Ranges that do not start at row 0 (text_base != 0), carriers forced to 1/1, and rows with three alleles in
    between (those members fall back to the text-domain encoder)."""
    from dna_factory_amd.rowgen import RowGenerator
    from dna_factory_amd.tables import SampleTable

    S, n = 5000, 400
    tb, st = _sorted_case(S, n, seed=3)
    autosome = np.flatnonzero(tb.chrom_class == 0)
    forced = [int(tb.snp_id[autosome[5]]), int(tb.snp_id[autosome[40]])]
    carriers = np.arange(S // 2, S, 3, dtype=np.uint32)
    st = SampleTable(st.flags, forced, [0, len(carriers), 2 * len(carriers)], np.concatenate([carriers, carriers]))
    tb.alt = tb.alt.astype("S3")
    for j in autosome[100:110]:  # three alleles: cells may hold '2'
        tb.n_alleles[j] = 3
        tb.cum[j] = [0.6, 0.8, 1.0, 1.0]
        tb.alt[j] = b"C,G" if tb.ref[j] not in (b"C", b"G") else b"A,T"
    with RowGenerator(device=0, seed=5) as g:
        g.set_samples(st)
        g.set_snps(tb)
        for b, m in ((0, n), (7, 150), (95, 30), (n - 20, 20), (33, 1)):
            text, _, _ = g.generate(b, m, text=True, packed=False)
            blob, n_text = g.generate_bgzf(b, m)
            assert n_text == len(text)
            assert gzip.decompress(blob.tobytes()) == text.tobytes()
        # the forced rows really are 1/1 for the carriers in what was compressed
        text, off, _ = g.generate(text=True, packed=False)
        row = int(np.flatnonzero(tb.snp_id == forced[0])[0])
        line = text[int(off[row]): int(off[row + 1])].tobytes().rstrip(b"\n").split(b"\t")[9:]
        assert all(line[i] == b"1/1" for i in carriers[:50])


@pytest.mark.gpu
def test_gtz_extreme_rows(capi):
    """All-reference rows (one long distance-4 run), rows at MAF 0.5 (dense LZ77 matches) and an alternating
    pattern forced through the deleterious set: sizes stay within BGZF's 64 KiB and everything inflates."""
    from dna_factory_amd.rowgen import RowGenerator
    from dna_factory_amd.tables import SampleTable, SnpTable

    S, n = 20000, 24
    cum = np.ones((n, 4))
    cum[:, 0] = np.where(np.arange(n) % 3 == 0, 1.0, np.where(np.arange(n) % 3 == 1, 0.5, 0.995))
    tb = SnpTable(np.arange(1, n + 1), np.full(n, b"2", dtype="S2"), np.arange(n) * 100 + 1, np.full(n, b"A", dtype="S1"),
                  np.full(n, b"T", dtype="S1"), np.full(n, 2, np.uint8), cum)
    flags = np.zeros(S, np.uint8)  # all cases, all female
    carriers = np.arange(0, S, 2, dtype=np.uint32)  # every other sample is 1/1 on row 0: "1/1\t0/0\t" repeated
    st = SampleTable(flags, [1], [0, len(carriers)], carriers)
    with RowGenerator(device=0, seed=1) as g:
        g.set_samples(st)
        g.set_snps(tb)
        text, _, _ = g.generate(text=True, packed=False)
        blob, _ = g.generate_bgzf()
    assert gzip.decompress(blob.tobytes()) == text.tobytes()
    assert text[:200].tobytes().count(b"1/1\t0/0\t") > 10


@pytest.mark.gpu
def test_bgzf_random_shapes_and_ranges(capi):
    """Forty random combinations of sample count, table size, fixed-column width, sex mix, forced set and row range
    (tools/bgzf_experiments/fuzz.py ran 10,000 of them): whatever the mix of encoders, the output inflates to the text."""
    from dna_factory_amd import genome
    from dna_factory_amd.rowgen import RowGenerator
    from dna_factory_amd.tables import SampleTable

    rs = np.random.RandomState(2026)
    for case in range(40):
        S = int([rs.randint(2, 40), rs.randint(900, 5000), rs.randint(5000, 40000), rs.choice([15359, 15360, 15361, 30720])][case % 4])
        n = int(max(3, min(1500, 12e6 / (4 * S + 40))))
        tb = genome.synthetic_snp_table(n, min_maf=float(rs.choice([0.005, 0.05, 0.3])), rs=rs)
        if case % 3 == 0:
            tb.snp_id = tb.snp_id * int(rs.choice([1000, 10**9])) + 3
        flags = (rs.rand(S) <= rs.choice([0.0, 0.5, 1.0])).astype(np.uint8) | (np.arange(S) < S // 2).astype(np.uint8) * 2
        carriers = np.sort(rs.choice(np.arange(S // 2, S), size=max(1, (S - S // 2) // 3), replace=False)).astype(np.uint32)
        st = SampleTable(flags, [int(tb.snp_id[n // 2])], [0, len(carriers)], carriers)
        with RowGenerator(device=0, seed=int(rs.randint(0, 2**31))) as g:
            g.set_samples(st)
            g.set_snps(tb)
            b = int(rs.randint(0, n))
            m = int(rs.randint(1, n - b + 1))
            text, _, _ = g.generate(b, m, text=True, packed=False)
            blob, n_text = g.generate_bgzf(b, m)
            assert n_text == len(text) and gzip.decompress(blob.tobytes()) == text.tobytes(), (case, S, n, b, m)


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bit_count_self_check_fires_on_a_wrong_count(tmp_path):
    """The chunks' bit offsets in a member come from gtz_count_bits (mask arithmetic), the bits from gtz_walk; the histogram
    kernel runs both on every sampled chunk and dnaf_bgzf_compress refuses to go on when they disagree.  A build whose count
    is wrong in one rare case (-DDNAF_GTZ_BREAK_COUNT: chunks of 64 cells whose last allele is an event) must be refused --
    and is, instead of writing members that do not inflate."""
    import shutil
    import subprocess
    import sys

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    csrc = os.path.join(REPO, "dna-factory_amd", "csrc")
    lib = str(tmp_path / "libdnaf_broken_count.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDNAF_GTZ_BREAK_COUNT", "-o", lib,
                           "dnaf_api.hip"], cwd=csrc)
    script = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_bgzf as t\n"
        "from dna_factory_amd import capi\n"
        "from dna_factory_amd.rowgen import RowGenerator\n"
        "tb, st = t._sorted_case(20000, 64, seed=1)\n"
        "with RowGenerator(device=0, seed=11) as g:\n"
        "    g.set_samples(st); g.set_snps(tb)\n"
        "    try:\n"
        "        g.generate_bgzf()\n"
        "    except capi.DnafError as e:\n"
        "        print('REFUSED', e)\n"
        "    else:\n"
        "        print('WROTE MEMBERS')\n" % (REPO, os.path.join(REPO, "tests")))
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, DNAF_LIB=lib), capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout and "bit count disagrees" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
