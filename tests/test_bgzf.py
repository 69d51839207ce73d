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
def test_gpu_bgzf_inflates_to_the_text(capi, n_rows, S):
    from dna_factory_amd.rowgen import RowGenerator

    rs = np.random.RandomState(n_rows + S)
    snps, samples = random_case(rs, n_rows, S)
    with RowGenerator(device=0, seed=3) as g:
        g.set_samples(samples)
        g.set_snps(snps)
        text, _, _ = g.generate(text=True, packed=False)
        blob, n_text = g.generate_bgzf()
        blob2, _ = g.generate_bgzf(reuse_tables=True)
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
