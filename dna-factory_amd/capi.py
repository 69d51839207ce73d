"""ctypes binding of the C ABI in include/dnaf.h (libdnaf_hip.so).

This is the only way into the generator: there is no CPU fallback.  If the
HIP library is missing or fails to load, importing this module raises.
"""
import ctypes
import os

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("DNAF_LIB", os.path.join(_CSRC, "libdnaf_hip.so"))

ABI_VERSION = 2

OK = 0
ERR_BAD_ARG, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_STATE, ERR_THRESHOLD = -1, -2, -3, -4, -5, -6
ERR_NAMES = {-1: "DNAF_ERR_BAD_ARG", -2: "DNAF_ERR_HIP", -3: "DNAF_ERR_OOM", -4: "DNAF_ERR_UNSUPPORTED",
             -5: "DNAF_ERR_STATE", -6: "DNAF_ERR_THRESHOLD"}

SAMPLE_MALE, SAMPLE_CONTROL = 1, 2
CLASS_AUTOSOME, CLASS_X, CLASS_Y, CLASS_MT = 0, 1, 2, 3
MAX_ALLELES = 4
RNG_PHILOX, RNG_MT19937 = 0, 1
OUT_TEXT, OUT_PACKED = 1, 2
BGZF_REUSE_TABLES = 1
BGZF_TEXT_DOMAIN = 2

# every symbol include/dnaf.h declares
SYMBOLS = [
    "dnaf_abi_version", "dnaf_device_count", "dnaf_create", "dnaf_destroy", "dnaf_last_error", "dnaf_set_samples",
    "dnaf_set_forced", "dnaf_set_rng", "dnaf_set_snps", "dnaf_text_bytes", "dnaf_row_offsets", "dnaf_generate",
    "dnaf_reserve_output",
    "dnaf_sync", "dnaf_last_generate_ms", "dnaf_timing_begin", "dnaf_timing_end", "dnaf_read_text", "dnaf_read_packed", "dnaf_device_buffers",
    "dnaf_format_prefixes", "dnaf_bgzf_compress", "dnaf_read_bgzf", "dnaf_last_bgzf_ms",
    "dnaf_bgzf_tables_from_histogram", "dnaf_gtz_tables_from_histogram", "dnaf_set_mt_state", "dnaf_set_rng_row0", "dnaf_set_rng_divisor", "dnaf_last_kernel", "dnaf_mt_jump_host", "dnaf_mt_characteristic_polynomial",
    "dnaf_fetch_bgzf", "dnaf_fetch_text", "dnaf_read_text_range", "dnaf_format_snps_json", "dnaf_parse_snps_json",
    "dnaf_allele_counts", "dnaf_export_bed", "dnaf_read_bed", "dnaf_packed_pitch_words", "dnaf_numpy_legacy_choice",
]


class DnafError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, str(code)), message))
        self.code = code


class SnpTableStruct(ctypes.Structure):
    """struct dnaf_snp_table"""
    _fields_ = [
        ("n_rows", ctypes.c_uint64),
        ("snp_id", ctypes.c_void_p),
        ("chrom_class", ctypes.c_void_p),
        ("n_alleles", ctypes.c_void_p),
        ("cum_freq", ctypes.c_void_p),
        ("prefix_bytes", ctypes.c_void_p),
        ("prefix_off", ctypes.c_void_p),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP library %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    if "DNAF_LIB" not in os.environ:
        # the .so is git-ignored but travels to the GPU box: refuse one that was built from other sources
        import importlib.util

        spec = importlib.util.spec_from_file_location("_dnaf_build", os.path.join(os.path.dirname(_CSRC), "build.py"))
        bld = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bld)
        if bld.is_stale():
            raise ImportError("HIP library %s is stale (csrc/ or include/dnaf.h changed since it was built): rebuild "
                              "with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    u64, u32, vp, ci = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int
    sig = {
        "dnaf_abi_version": (ci, []),
        "dnaf_device_count": (ci, [vp]),
        "dnaf_create": (ci, [ci, vp]),
        "dnaf_destroy": (ci, [vp]),
        "dnaf_last_error": (ctypes.c_char_p, [vp]),
        "dnaf_set_samples": (ci, [vp, vp, u64]),
        "dnaf_set_forced": (ci, [vp, vp, u64, vp, vp]),
        "dnaf_set_rng": (ci, [vp, u32, u64, u64]),
        "dnaf_set_snps": (ci, [vp, vp]),
        "dnaf_text_bytes": (ci, [vp, u64, u64, vp]),
        "dnaf_row_offsets": (ci, [vp, u64, u64, vp]),
        "dnaf_generate": (ci, [vp, u64, u64, u32]),
        "dnaf_reserve_output": (ci, [vp, u64, u64]),
        "dnaf_sync": (ci, [vp]),
        "dnaf_last_generate_ms": (ci, [vp, vp]),
        "dnaf_timing_begin": (ci, [vp]),
        "dnaf_timing_end": (ci, [vp, vp, vp, vp]),
        "dnaf_read_text": (ci, [vp, vp, u64, vp]),
        "dnaf_read_packed": (ci, [vp, vp, u64, vp]),
        "dnaf_device_buffers": (ci, [vp, vp, vp, vp, vp]),
        "dnaf_packed_pitch_words": (ci, [vp, vp]),
        "dnaf_numpy_legacy_choice": (ci, [vp, vp, u64, u32, vp]),
        "dnaf_format_prefixes": (ci, [u64, vp, u32, vp, vp, vp, u32, vp, u32, vp, u64, vp]),
        "dnaf_bgzf_compress": (ci, [vp, u32, vp]),
        "dnaf_read_bgzf": (ci, [vp, vp, u64, vp]),
        "dnaf_last_bgzf_ms": (ci, [vp, vp]),
        "dnaf_bgzf_tables_from_histogram": (ci, [vp, vp, u64, vp, u64, vp]),
        "dnaf_gtz_tables_from_histogram": (ci, [vp, vp, u64, vp]),
        "dnaf_set_mt_state": (ci, [vp, vp, u32, u64]),
        "dnaf_set_rng_row0": (ci, [vp, u64]),
        "dnaf_set_rng_divisor": (ci, [vp, u32]),
        "dnaf_last_kernel": (ci, [vp, vp, u64]),
        "dnaf_fetch_bgzf": (ci, [vp, vp, vp]),
        "dnaf_fetch_text": (ci, [vp, vp, vp]),
        "dnaf_read_text_range": (ci, [vp, u64, u64, vp]),
        "dnaf_allele_counts": (ci, [vp, vp, u64]),
        "dnaf_export_bed": (ci, [vp, vp]),
        "dnaf_read_bed": (ci, [vp, vp, u64, vp]),
        "dnaf_format_snps_json": (ci, [u64, vp, vp, u32, vp, vp, u32, vp, u32, vp, vp, vp, vp, vp, u64, vp]),
        "dnaf_parse_snps_json": (ci, [vp, u64, u64, vp, vp, u32, vp, vp, u32, vp, vp, vp]),
        "dnaf_mt_jump_host": (ci, [vp, u64, vp]),
        "dnaf_mt_characteristic_polynomial": (ci, [vp, vp]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here = the library does not export what the header declares
        fn.restype, fn.argtypes = sig[name]
    if lib.dnaf_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, binding expects %d" % (LIB_PATH, lib.dnaf_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def device_count():
    n = ctypes.c_int(0)
    rc = lib.dnaf_device_count(ctypes.byref(n))
    if rc != OK:
        raise DnafError(rc, lib.dnaf_last_error(None).decode())
    return n.value


def numpy_legacy_choice(pop_size, size):
    """numpy.random.choice(pop_size, size, replace=False) on numpy's global legacy generator -- the same indices, the same
    generator state afterwards -- without the O(pop_size) shuffle of an array (dnaf_numpy_legacy_choice)."""
    import numpy

    st = numpy.random.get_state()
    # the replay traces each wanted entry back through the shuffle's swaps: O(pop_size x size) compares -- right for the
    # handful of weights a deleterious group has, not for hundreds (a group with 200 weights over 40 M SNPs: numpy's own
    # shuffle, 2 s, is the faster one from about 32 entries up)
    if st[0] != "MT19937" or size > pop_size or pop_size >= 2**32 or pop_size == 0 or size > 32:
        return numpy.random.choice(pop_size, size, replace=False)
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_uint32(int(st[2]))
    out = np.empty(size, dtype=np.uint64)
    rc = lib.dnaf_numpy_legacy_choice(ptr(key), ctypes.byref(pos), ctypes.c_uint64(pop_size), ctypes.c_uint32(size), ptr(out))
    if rc != OK:
        raise DnafError(rc, "dnaf_numpy_legacy_choice")
    numpy.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out.astype(np.int64)


def format_prefixes(chrom, pos, snp_id, ref, alt):
    """Fixed columns of each body line (pop_factory.py:503-507) -> (blob u8[], off u64[n+1])."""
    n = len(pos)
    chrom = np.ascontiguousarray(chrom)
    ref = np.ascontiguousarray(ref)
    alt = np.ascontiguousarray(alt)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    snp_id = np.ascontiguousarray(snp_id, dtype=np.int64)
    for a in (chrom, ref, alt):
        if a.dtype.kind != "S" or len(a) != n:
            raise ValueError("chrom/ref/alt must be numpy bytes ('S<w>') arrays of length n")
    off = np.zeros(n + 1, dtype=np.uint64)
    args = (n, ptr(chrom), chrom.dtype.itemsize, ptr(pos), ptr(snp_id), ptr(ref), ref.dtype.itemsize, ptr(alt),
            alt.dtype.itemsize)
    rc = lib.dnaf_format_prefixes(*args, None, 0, ptr(off))
    if rc != OK:
        raise DnafError(rc, "dnaf_format_prefixes (sizing)")
    blob = np.empty(max(int(off[n]), 1), dtype=np.uint8)
    rc = lib.dnaf_format_prefixes(*args, ptr(blob), int(off[n]), ptr(off))
    if rc != OK:
        raise DnafError(rc, "dnaf_format_prefixes")
    return blob[: int(off[n])], off


def format_snps_json(snp_id, chrom, pos, ref, alt, n_alleles, cum):
    """The text of snps.json.gz for these rows (one `str(SNPTuples)` per line, pop_factory.py:118-124) as bytes.
    Returns None when a field would need JSON escaping (the caller then formats in Python)."""
    n = len(snp_id)
    chrom, ref, alt = np.ascontiguousarray(chrom), np.ascontiguousarray(ref), np.ascontiguousarray(alt)
    for a in (chrom, ref, alt):
        raw = a.view(np.uint8)
        if raw.size and (np.any(raw == 0x22) or np.any(raw == 0x5C) or np.any((raw != 0) & (raw < 0x20)) or np.any(raw > 0x7E)):
            return None
    snp_id = np.ascontiguousarray(snp_id, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    n_alleles = np.ascontiguousarray(n_alleles, dtype=np.uint8)
    cum = np.ascontiguousarray(cum, dtype=np.float64)
    uniq, inv = np.unique(cum, return_inverse=True)
    if np.any(~np.isfinite(uniq)):
        return None
    strs = [repr(float(u)).encode() for u in uniq]  # json.dumps(float) is float.__repr__
    num_off = np.zeros(len(strs) + 1, dtype=np.uint64)
    num_off[1:] = np.cumsum([len(x) for x in strs])
    num_blob = np.frombuffer(b"".join(strs) or b"\0", dtype=np.uint8)
    cum_idx = np.ascontiguousarray(inv.reshape(cum.shape), dtype=np.int32)
    nb = ctypes.c_uint64(0)
    args = (n, ptr(snp_id), ptr(chrom), chrom.dtype.itemsize, ptr(pos), ptr(ref), ref.dtype.itemsize, ptr(alt),
            alt.dtype.itemsize, ptr(n_alleles), ptr(cum_idx), ptr(num_blob), ptr(num_off))
    rc = lib.dnaf_format_snps_json(*args, None, 0, ctypes.byref(nb))
    if rc != OK:
        raise DnafError(rc, "dnaf_format_snps_json (sizing)")
    out = np.empty(max(nb.value, 1), dtype=np.uint8)
    rc = lib.dnaf_format_snps_json(*args, ptr(out), nb.value, ctypes.byref(nb))
    if rc != OK:
        raise DnafError(rc, "dnaf_format_snps_json")
    return out[: nb.value]


def parse_snps_json(text, chrom_w=8, allele_w=16):
    """Columns of a snps.json text (bytes): (id, chrom S, pos, alleles S[n][4], cum f64[n][4], n_alleles) or None when
    a line is outside the fast path's format."""
    buf = np.frombuffer(text, dtype=np.uint8)
    max_rows = int(np.count_nonzero(buf == 0x0A)) + 1
    sid = np.zeros(max_rows, dtype=np.int64)
    pos = np.zeros(max_rows, dtype=np.int64)
    chrom = np.zeros(max_rows, dtype="S%d" % chrom_w)
    alleles = np.zeros((max_rows, 4), dtype="S%d" % allele_w)
    cum = np.ones((max_rows, 4), dtype=np.float64)
    na = np.zeros(max_rows, dtype=np.uint8)
    n = ctypes.c_uint64(0)
    rc = lib.dnaf_parse_snps_json(ptr(buf) if len(buf) else None, len(buf), max_rows, ptr(sid), ptr(chrom), chrom_w, ptr(pos),
                                  ptr(alleles), allele_w, ptr(cum), ptr(na), ctypes.byref(n)) if len(buf) else OK
    if rc == ERR_UNSUPPORTED:
        return None
    if rc != OK:
        raise DnafError(rc, "dnaf_parse_snps_json")
    k = n.value
    return sid[:k], chrom[:k], pos[:k], alleles[:k], cum[:k], na[:k]


class Context:
    """One dnaf_ctx: one GPU, one stream.  Not thread-safe."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p(None)
        rc = lib.dnaf_create(int(device), ctypes.byref(self._h))
        if rc != OK:
            raise DnafError(rc, lib.dnaf_last_error(None).decode())
        self.device = int(device)

    def _check(self, rc):
        if rc != OK:
            raise DnafError(rc, lib.dnaf_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib.dnaf_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_samples(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(lib.dnaf_set_samples(self._h, ptr(flags), len(flags)))

    def set_forced(self, snp_ids, carrier_off, carrier_idx):
        snp_ids = np.ascontiguousarray(snp_ids, dtype=np.int64)
        carrier_off = np.ascontiguousarray(carrier_off, dtype=np.uint64)
        carrier_idx = np.ascontiguousarray(carrier_idx, dtype=np.uint32)
        if len(snp_ids) and len(carrier_off) != len(snp_ids) + 1:
            raise ValueError("carrier_off must have n_slots+1 entries")
        self._check(lib.dnaf_set_forced(self._h, ptr(snp_ids), len(snp_ids), ptr(carrier_off), ptr(carrier_idx)))

    def set_rng(self, mode, seed, rng_row0=0):
        self._check(lib.dnaf_set_rng(self._h, mode, int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_row0)))

    def set_mt_state(self, key, pos, rng_row0=0):
        """MT19937 mode from numpy.random.get_state()[1:3] (the stream position the reference's worker starts at)."""
        key = np.ascontiguousarray(key, dtype=np.uint32)
        if key.shape != (624,):
            raise ValueError("key must hold 624 words")
        self._check(lib.dnaf_set_mt_state(self._h, ptr(key), int(pos), int(rng_row0)))

    def set_rng_row0(self, rng_row0):
        """Only the RNG row of table row 0 changes (the next tile of a larger job); mode, seed and cached state stay."""
        self._check(lib.dnaf_set_rng_row0(self._h, int(rng_row0)))

    def last_kernel(self):
        """Name of the row kernel the last generate() launched."""
        buf = ctypes.create_string_buffer(64)
        self._check(lib.dnaf_last_kernel(self._h, buf, 64))
        return buf.value.decode()

    def set_rng_divisor(self, n):
        """Table rows k n .. k n + n - 1 share RNG row k: the stream of the reference's `-n n` forked workers."""
        self._check(lib.dnaf_set_rng_divisor(self._h, int(n)))

    def set_snps(self, snp_id, chrom_class, n_alleles, cum_freq, prefix_bytes, prefix_off):
        n = len(snp_id)
        snp_id = np.ascontiguousarray(snp_id, dtype=np.int64)
        chrom_class = np.ascontiguousarray(chrom_class, dtype=np.uint8)
        n_alleles = np.ascontiguousarray(n_alleles, dtype=np.uint8)
        cum_freq = np.ascontiguousarray(cum_freq, dtype=np.float64)
        prefix_bytes = np.ascontiguousarray(prefix_bytes, dtype=np.uint8)
        prefix_off = np.ascontiguousarray(prefix_off, dtype=np.uint64)
        if cum_freq.shape != (n, MAX_ALLELES) or len(chrom_class) != n or len(n_alleles) != n or len(prefix_off) != n + 1:
            raise ValueError("SNP table columns have inconsistent shapes")
        if n and int(prefix_off[n]) > len(prefix_bytes):
            raise ValueError("prefix_off runs past prefix_bytes")
        t = SnpTableStruct(n, ptr(snp_id).value, ptr(chrom_class).value, ptr(n_alleles).value, ptr(cum_freq).value,
                           ptr(prefix_bytes).value if len(prefix_bytes) else None, ptr(prefix_off).value)
        self._check(lib.dnaf_set_snps(self._h, ctypes.byref(t)))

    def text_bytes(self, row_begin, n_rows):
        v = ctypes.c_uint64(0)
        self._check(lib.dnaf_text_bytes(self._h, row_begin, n_rows, ctypes.byref(v)))
        return v.value

    def row_offsets(self, row_begin, n_rows):
        off = np.zeros(n_rows + 1, dtype=np.uint64)
        self._check(lib.dnaf_row_offsets(self._h, row_begin, n_rows, ptr(off)))
        return off

    def generate(self, row_begin, n_rows, what=OUT_TEXT | OUT_PACKED):
        self._check(lib.dnaf_generate(self._h, row_begin, n_rows, what))

    def reserve_output(self, text_bytes, packed_words):
        self._check(lib.dnaf_reserve_output(self._h, int(text_bytes), int(packed_words)))

    def sync(self):
        self._check(lib.dnaf_sync(self._h))

    def last_generate_ms(self):
        v = ctypes.c_float(0)
        self._check(lib.dnaf_last_generate_ms(self._h, ctypes.byref(v)))
        return v.value

    def timing_begin(self):
        self._check(lib.dnaf_timing_begin(self._h))

    def timing_end(self):
        """-> (summed kernel ms, launches recorded, launches made)"""
        tot = ctypes.c_double(0)
        nr, nl = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(lib.dnaf_timing_end(self._h, ctypes.byref(tot), ctypes.byref(nr), ctypes.byref(nl)))
        return tot.value, nr.value, nl.value

    def read_text(self, n_bytes):
        buf = np.empty(max(n_bytes, 1), dtype=np.uint8)
        got = ctypes.c_uint64(0)
        self._check(lib.dnaf_read_text(self._h, ptr(buf), n_bytes, ctypes.byref(got)))
        return buf[: got.value]

    def allele_counts(self, n_rows):
        """(control alt-allele count, case alt-allele count) per row of the last generated packed matrix: u32[n][2]."""
        out = np.zeros((n_rows, 2), dtype=np.uint32)
        self._check(lib.dnaf_allele_counts(self._h, ptr(out.reshape(-1)) if n_rows else None, n_rows))
        return out

    def read_text_range(self, byte_begin, n_bytes):
        buf = np.empty(max(n_bytes, 1), dtype=np.uint8)
        self._check(lib.dnaf_read_text_range(self._h, int(byte_begin), int(n_bytes), ptr(buf)))
        return buf[:n_bytes]

    def read_packed(self, n_rows, words_per_row):
        buf = np.empty((n_rows, words_per_row), dtype=np.uint32)
        got = ctypes.c_uint64(0)
        flat = buf.reshape(-1) if buf.size else np.empty(1, dtype=np.uint32)
        self._check(lib.dnaf_read_packed(self._h, ptr(flat), buf.size, ctypes.byref(got)))
        return buf

    def bgzf_compress(self, reuse_tables=False, text_domain=False):
        """Compress the last generated text on the GPU; returns the number of BGZF bytes.  Members in biallelic
        autosome rows are made from the genotype matrix when it was generated too (text_domain=True: never)."""
        n = ctypes.c_uint64(0)
        flags = (BGZF_REUSE_TABLES if reuse_tables else 0) | (BGZF_TEXT_DOMAIN if text_domain else 0)
        self._check(lib.dnaf_bgzf_compress(self._h, flags, ctypes.byref(n)))
        return n.value

    def export_bed(self):
        """PLINK .bed rows (without the 3 magic bytes) of the last packed matrix: u8[n_rows][ceil(S/4)]."""
        n = ctypes.c_uint64(0)
        self._check(lib.dnaf_export_bed(self._h, ctypes.byref(n)))
        buf = np.empty(max(n.value, 1), dtype=np.uint8)
        self._check(lib.dnaf_read_bed(self._h, ptr(buf), n.value, ctypes.byref(n)))
        return buf[: n.value]

    def read_bgzf(self, n_bytes):
        buf = np.empty(max(n_bytes, 1), dtype=np.uint8)
        got = ctypes.c_uint64(0)
        self._check(lib.dnaf_read_bgzf(self._h, ptr(buf), n_bytes, ctypes.byref(got)))
        return buf[: got.value]

    def _fetch(self, fn):
        host = ctypes.c_void_p(None)
        n = ctypes.c_uint64(0)
        self._check(fn(self._h, ctypes.byref(host), ctypes.byref(n)))
        if n.value == 0:
            return np.empty(0, dtype=np.uint8)
        # a view of the library's pinned buffer: valid until the next fetch on this context
        return np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,))

    def fetch_bgzf(self):
        """The last compressed range as a numpy view of pinned host memory (no extra copy)."""
        return self._fetch(lib.dnaf_fetch_bgzf)

    def fetch_text(self):
        return self._fetch(lib.dnaf_fetch_text)

    def last_bgzf_ms(self):
        v = ctypes.c_float(0)
        self._check(lib.dnaf_last_bgzf_ms(self._h, ctypes.byref(v)))
        return v.value

    def packed_pitch_words(self):
        v = ctypes.c_uint64(0)
        self._check(lib.dnaf_packed_pitch_words(self._h, ctypes.byref(v)))
        return v.value

    def device_buffers(self):
        t, p = ctypes.c_void_p(None), ctypes.c_void_p(None)
        tb, pw = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(lib.dnaf_device_buffers(self._h, ctypes.byref(t), ctypes.byref(tb), ctypes.byref(p), ctypes.byref(pw)))
        return t.value, tb.value, p.value, pw.value
