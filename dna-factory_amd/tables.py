"""Structure-of-arrays views of the worker's two inputs.

`SnpTable` holds what the worker reads of each `SNPTuples`
(pop_factory.py:74-133) and `SampleTable` what it reads of each `SampleInfo`
(pop_factory.py:47-71).  Both can be built from reference-shaped objects
(duck-typed) or directly from numpy columns for large jobs.
"""
import numpy as np

from . import capi

_CLASS_OF = {"X": capi.CLASS_X, "Y": capi.CLASS_Y, "MT": capi.CLASS_MT}


def chrom_class_of(chrom):
    """Row class = everything is_haploid (common/snp.py:102-109) and the female-on-Y rule
    (pop_factory.py:481) ask about a chromosome string."""
    return _CLASS_OF.get(chrom, capi.CLASS_AUTOSOME)


def _fit_bytes(values, min_width=1):
    enc = [v if isinstance(v, bytes) else str(v).encode() for v in values]
    w = max([min_width] + [len(e) for e in enc])
    return np.array(enc, dtype="S%d" % w) if enc else np.zeros(0, dtype="S%d" % w)


class SnpTable:
    """Columns: snp_id i64, chrom S<w>, position i64, ref S<w>, alt S<w> (already comma-joined,
    alt_alleles() pop_factory.py:111-116), n_alleles u8, cum f64[n][4]."""

    def __init__(self, snp_id, chrom, position, ref, alt, n_alleles, cum):
        self.snp_id = np.ascontiguousarray(snp_id, dtype=np.int64)
        self.chrom = np.ascontiguousarray(chrom)
        self.position = np.ascontiguousarray(position, dtype=np.int64)
        self.ref = np.ascontiguousarray(ref)
        self.alt = np.ascontiguousarray(alt)
        self.n_alleles = np.ascontiguousarray(n_alleles, dtype=np.uint8)
        self.cum = np.ascontiguousarray(cum, dtype=np.float64)
        n = len(self.snp_id)
        if self.cum.shape != (n, capi.MAX_ALLELES):
            raise ValueError("cum must be [n][%d]" % capi.MAX_ALLELES)
        for col in (self.chrom, self.position, self.ref, self.alt, self.n_alleles):
            if len(col) != n:
                raise ValueError("SNP table columns differ in length")
        for col in (self.chrom, self.ref, self.alt):
            if col.dtype.kind != "S":
                raise ValueError("chrom/ref/alt must be numpy bytes arrays")
        self._class = None
        self._prefix = None

    def __len__(self):
        return len(self.snp_id)

    @classmethod
    def from_snps(cls, snps):
        """snps: iterable of SNPTuples-like objects (.id .chromosome .position .tuples) or
        (id, chromosome, position, tuples) tuples; tuples = [(allele string, cumulative freq), ...]."""
        ids, chroms, poss, refs, alts, nal, cums = [], [], [], [], [], [], []
        for s in snps:
            if hasattr(s, "tuples"):
                sid, chrom, pos, tuples = s.id, s.chromosome, s.position, s.tuples
            else:
                sid, chrom, pos, tuples = s
            tuples = list(tuples)
            if not 1 <= len(tuples) <= capi.MAX_ALLELES:
                raise capi.DnafError(capi.ERR_UNSUPPORTED, "SNP %r has %d alleles (1..%d supported)"
                                     % (sid, len(tuples), capi.MAX_ALLELES))
            ids.append(int(sid))
            chroms.append(str(chrom))
            poss.append(int(pos))
            refs.append(str(tuples[0][0]))
            if len(tuples) == 1:
                alts.append(str(tuples[0][0]))
            elif len(tuples) == 2:
                alts.append(str(tuples[1][0]))
            else:
                alts.append(",".join(str(t[0]) for t in tuples[1:]))
            nal.append(len(tuples))
            row = [float(t[1]) for t in tuples] + [1.0] * (capi.MAX_ALLELES - len(tuples))
            cums.append(row)
        cum = np.array(cums, dtype=np.float64).reshape(len(ids), capi.MAX_ALLELES)
        return cls(ids, _fit_bytes(chroms), poss, _fit_bytes(refs), _fit_bytes(alts), nal, cum)

    @property
    def chrom_class(self):
        if self._class is None:
            c = np.zeros(len(self), dtype=np.uint8)
            for name, code in _CLASS_OF.items():
                c[self.chrom == name.encode()] = code
            self._class = c
        return self._class

    def prefixes(self):
        """(blob, off): each row's "chrom\\tpos\\trsID\\tREF\\tALT\\t40\\tPASS\\t.\\tGT\\t" (pop_factory.py:503-507)."""
        if self._prefix is None:
            self._prefix = capi.format_prefixes(self.chrom, self.position, self.snp_id, self.ref, self.alt)
        return self._prefix

    def take(self, order):
        """Rows re-ordered / selected by an index array."""
        return SnpTable(self.snp_id[order], self.chrom[order], self.position[order], self.ref[order], self.alt[order],
                        self.n_alleles[order], self.cum[order])

    def slice(self, begin, end):
        t = SnpTable(self.snp_id[begin:end], self.chrom[begin:end], self.position[begin:end], self.ref[begin:end],
                     self.alt[begin:end], self.n_alleles[begin:end], self.cum[begin:end])
        if self._prefix is not None:  # the formatted fixed columns go along (tiles of a table that was formatted once)
            blob, off = self._prefix
            begin, end = max(0, min(begin, len(self))), max(0, min(end, len(self)))
            if end > begin:
                o = off[begin:end + 1]
                t._prefix = (blob[int(o[0]):int(o[-1])], o - o[0])
        if self._class is not None:
            t._class = self._class[begin:end]
        return t


def _key_matches_int(k):
    # `snp.id not in sample.deleterious_snps` (pop_factory.py:485) is dict membership on an int id:
    # only keys that hash/compare equal to that int can match -- str keys (DeleteriousGroup.from_json,
    # pop_factory.py:617-618) never do.
    if isinstance(k, (bool, np.bool_)):
        return True
    if isinstance(k, (int, np.integer)):
        return True
    if isinstance(k, (float, np.floating)):
        return float(k).is_integer()
    return False


class SampleTable:
    """flags u8[S] (capi.SAMPLE_MALE | capi.SAMPLE_CONTROL) and the forced set transposed by SNP:
    forced_ids i64[n_slots], carrier_off u64[n_slots+1], carrier_idx u32[]."""

    def __init__(self, flags, forced_ids=None, carrier_off=None, carrier_idx=None, person_id=None):
        self.flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.forced_ids = np.zeros(0, np.int64) if forced_ids is None else np.ascontiguousarray(forced_ids, dtype=np.int64)
        self.carrier_off = (np.zeros(1, np.uint64) if carrier_off is None
                            else np.ascontiguousarray(carrier_off, dtype=np.uint64))
        self.carrier_idx = (np.zeros(0, np.uint32) if carrier_idx is None
                            else np.ascontiguousarray(carrier_idx, dtype=np.uint32))
        self.person_id = None if person_id is None else np.ascontiguousarray(person_id, dtype=np.int64)

    def __len__(self):
        return len(self.flags)

    @classmethod
    def from_samples(cls, samples):
        """samples: iterable of SampleInfo-like objects (.sex, .is_control, .deleterious_snps[, .person_id]) or
        (sex, is_control, carried ids or None) tuples, in fam order."""
        flags, pids, carriers = [], [], {}
        for i, s in enumerate(samples):
            if hasattr(s, "is_control"):
                sex, ctrl, carried = s.sex, s.is_control, s.deleterious_snps
                pids.append(int(getattr(s, "person_id", i)))
            else:
                sex, ctrl, carried = s
                pids.append(i)
            f = (capi.SAMPLE_MALE if sex == 1 else 0) | (capi.SAMPLE_CONTROL if ctrl else 0)
            flags.append(f)
            if not ctrl:
                if carried is None:
                    # the reference evaluates `snp.id not in None` -> TypeError for a case without a dict
                    raise TypeError("case sample %d has deleterious_snps=None (pop_factory.py:485 would raise)" % i)
                for k in (carried.keys() if hasattr(carried, "keys") else carried):
                    if _key_matches_int(k):
                        carriers.setdefault(int(k), []).append(i)
        ids = sorted(carriers)
        off = np.zeros(len(ids) + 1, dtype=np.uint64)
        idx = []
        for n, k in enumerate(ids):
            idx.extend(carriers[k])
            off[n + 1] = len(idx)
        return cls(flags, ids, off, np.asarray(idx, dtype=np.uint32), pids)
