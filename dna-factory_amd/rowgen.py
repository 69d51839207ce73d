"""Host-side mirror of the reference's worker/dispatcher seam on top of the C ABI.

`RowGenerator.write_vcf_snps(fam_data, snps, file)` has the meaning of
`PopulationFactory.write_vcf_snps` (pop_factory.py:417-469): append exactly
len(snps) VCF body lines to `file`, in order.  The rows are produced by the
gfx950 kernels (the work of `queue_vcf_snps`, pop_factory.py:471-513); the
stripe/fork/merge machinery of the reference has no counterpart because a
counter-based RNG makes every row independent of the partition.
"""
import io

import numpy as np

from . import capi
from .tables import SampleTable, SnpTable

RNG_MODES = {"philox": capi.RNG_PHILOX, "mt19937": capi.RNG_MT19937}


class RowGenerator:
    def __init__(self, device=0, seed=0, rng="philox", mt_state=None):
        """rng="philox": Philox4x32-10 keyed by `seed`.  rng="mt19937": numpy's legacy stream,
        either from `seed` (= numpy.random.seed(seed), what the unmodified reference worker draws when it is
        called right after seeding) or from an explicit `mt_state` = numpy.random.get_state()[1:3]."""
        if rng not in RNG_MODES:
            raise ValueError("rng must be one of %s" % sorted(RNG_MODES))
        if mt_state is not None and rng != "mt19937":
            raise ValueError("mt_state needs rng='mt19937'")
        self.ctx = capi.Context(device)
        self.seed = int(seed)
        self.rng = rng
        self.mt_state = None if mt_state is None else (np.array(mt_state[0], dtype=np.uint32), int(mt_state[1]))
        self._rng_is_set = False
        self._set_rng(0)
        self.n_samples = None
        self.n_rows = None

    def _set_rng(self, rng_row0):
        if self._rng_is_set:  # mode, seed / state are this generator's for life: later tables only move the RNG row
            self.ctx.set_rng_row0(rng_row0)
            return
        self._rng_is_set = True
        if self.mt_state is not None:
            self.ctx.set_mt_state(self.mt_state[0], self.mt_state[1], rng_row0)
        else:
            self.ctx.set_rng(RNG_MODES[self.rng], self.seed, rng_row0)

    @property
    def row_quantum(self):
        """Row ranges must start on a multiple of this many rows (1 for Philox; in MT19937 mode waves start on block
        boundaries of the sequential stream, every 156 / gcd(S, 156) rows)."""
        if self.rng != "mt19937" or not self.n_samples:
            return 1
        from math import gcd

        return 156 // gcd(self.n_samples, 156)

    def close(self):
        self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- inputs -------------------------------------------------------------
    def set_samples(self, samples):
        """samples: SampleTable or an iterable accepted by SampleTable.from_samples (the worker's fam_data)."""
        st = samples if isinstance(samples, SampleTable) else SampleTable.from_samples(samples)
        self.ctx.set_samples(st.flags)
        self.ctx.set_forced(st.forced_ids, st.carrier_off, st.carrier_idx)
        self.n_samples = len(st)
        self.n_rows = None
        return st

    def set_snps(self, snps, rng_row0=0):
        """snps: SnpTable or an iterable accepted by SnpTable.from_snps (the worker's work_q, in output order).
        rng_row0: RNG row index of the first row (global row index of a shard)."""
        if self.n_samples is None:
            raise capi.DnafError(capi.ERR_STATE, "set_samples must be called before set_snps")
        tb = snps if isinstance(snps, SnpTable) else SnpTable.from_snps(snps)
        blob, off = tb.prefixes()
        self._set_rng(rng_row0)
        self.ctx.set_snps(tb.snp_id, tb.chrom_class, tb.n_alleles, tb.cum, blob, off)
        self.n_rows = len(tb)
        return tb

    # ---- generation ---------------------------------------------------------
    @property
    def words_per_row(self):
        return (self.n_samples + 15) // 16

    def generate(self, row_begin=0, n_rows=None, text=True, packed=True):
        """Rows [row_begin, row_begin+n_rows) -> (text u8[] or None, row_off u64[n+1], packed u32[n][W] or None)."""
        if self.n_rows is None:
            raise capi.DnafError(capi.ERR_STATE, "set_snps must be called before generate")
        if n_rows is None:
            n_rows = self.n_rows - row_begin
        what = (capi.OUT_TEXT if text else 0) | (capi.OUT_PACKED if packed else 0)
        self.ctx.generate(row_begin, n_rows, what)
        off = self.ctx.row_offsets(row_begin, n_rows)
        t = self.ctx.read_text(int(off[n_rows])) if text else None
        p = self.ctx.read_packed(n_rows, self.words_per_row) if packed else None
        return t, off, p

    def generate_bgzf(self, row_begin=0, n_rows=None, reuse_tables=False, text_domain=False):
        """Rows [row_begin, row_begin+n_rows) as BGZF members compressed on the GPU (no EOF block): only the
        compressed bytes cross PCIe.  Returns (bgzf u8[], uncompressed text bytes).  text_domain=True keeps the
        genotype-domain encoder out (every member is made from the text)."""
        if self.n_rows is None:
            raise capi.DnafError(capi.ERR_STATE, "set_snps must be called before generate")
        if n_rows is None:
            n_rows = self.n_rows - row_begin
        self.ctx.generate(row_begin, n_rows, capi.OUT_TEXT | (0 if text_domain else capi.OUT_PACKED))
        n = self.ctx.bgzf_compress(reuse_tables=reuse_tables, text_domain=text_domain)
        return self.ctx.read_bgzf(n), self.ctx.text_bytes(row_begin, n_rows)

    def lines(self, row_begin=0, n_rows=None):
        t, off, _ = self.generate(row_begin, n_rows, text=True, packed=False)
        b = t.tobytes()
        return [b[int(off[j]): int(off[j + 1])].decode("ascii") for j in range(len(off) - 1)]

    def write_vcf_snps(self, fam_data, snps, file, tile_rows=None, rng_row0=0):
        """Drop-in for PopulationFactory.write_vcf_snps (pop_factory.py:417-418): appends len(snps) lines
        to `file` (text- or binary-mode, anything with .write) in order.  Returns the number of lines."""
        self.set_samples(fam_data)
        tb = self.set_snps(snps, rng_row0=rng_row0)
        n = len(tb)
        if tile_rows is None:
            # keep a tile's text near 256 MiB
            per_row = 64 + 4 * max(self.n_samples, 1)
            tile_rows = max(1, min(n, (256 << 20) // per_row)) if n else 1
        q = self.row_quantum
        tile_rows = max(q, tile_rows // q * q)
        as_text = isinstance(file, io.TextIOBase)
        for b in range(0, n, tile_rows):
            m = min(tile_rows, n - b)
            t, _, _ = self.generate(b, m, text=True, packed=False)
            data = t.tobytes()
            file.write(data.decode("ascii") if as_text else data)
        return n
