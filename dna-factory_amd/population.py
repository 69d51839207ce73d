"""Host side of a population run: everything around the row generator that the reference does in
`PopulationFactory.generate_population` (pop_factory.py:225-256) -- SNP list, deleterious picks, sample table,
side files, VCF header -- on structure-of-arrays data, with the output layouts of SURVEY Appendix A.

Seed-level compatibility of the feeders: every random decision is taken with the same numpy legacy
(`numpy.random.*`, global MT19937) and stdlib `random` calls, in the same order and on the same arguments, as the
reference takes them.  After `numpy.random.seed(k)` / `random.seed(j)` the SNP list, `deleterious.json`,
`population.fam` and `pop_deleterious.txt` therefore come out byte-identical to the reference's
(tests/test_population_host.py checks this against tests/golden/full_flow_120x400).
"""
import gzip
import json
import os
import random
from datetime import datetime

import numpy
import numpy as np

from . import genome
from .tables import SampleTable, SnpTable

# ---------------------------------------------------------------------------------------------------------
# SNP list
# ---------------------------------------------------------------------------------------------------------


def _replay_random_choice3(count):
    """The indices `count` successive `random.choice(<3-element list>)` calls return, leaving Python's `random`
    exactly where those calls would leave it -- without a Python-level loop.

    CPython: choice(seq) = seq[_randbelow(3)], and _randbelow(3) draws getrandbits(2) = next MT19937 word >> 30
    until the value is below 3.  The words come from numpy's RandomState loaded with `random`'s own state (both wrap
    the same generator), so the stream and the final state are the interpreter's.  A prefix is cross-checked against
    real `random.choice` calls; on any disagreement (another interpreter's `random`) the plain loop is used."""
    if count == 0:
        return np.zeros(0, dtype=np.int64)
    state = random.getstate()

    def plain():
        return np.fromiter((random.choice((0, 1, 2)) for _ in range(count)), dtype=np.int64, count=count)

    if state[0] != 3 or len(state[1]) != 625:
        return plain()
    key = np.array(state[1][:624], dtype=np.uint32)
    rs = numpy.random.RandomState()
    rs.set_state(("MT19937", key, int(state[1][624])))
    picks, consumed = [], 0
    need = count
    while need > 0:
        words = rs.randint(0, 2**32, size=need + need // 2 + 64, dtype=np.uint32)
        r = (words >> np.uint32(30)).astype(np.int64)
        ok = np.flatnonzero(r < 3)
        if len(ok) >= need:
            ok = ok[:need]
            picks.append(r[ok])
            consumed += int(ok[-1]) + 1
            need = 0
        else:
            picks.append(r[ok])
            consumed += len(words)
            need -= len(ok)
    picks = np.concatenate(picks)
    # cross-check a prefix against the interpreter, then put `random` where the calls would have left it
    probe = min(count, 512)
    real = [random.choice((0, 1, 2)) for _ in range(probe)]
    random.setstate(state)
    if real != [int(x) for x in picks[:probe]]:
        return plain()
    rs.set_state(("MT19937", key, int(state[1][624])))
    rs.randint(0, 2**32, size=consumed, dtype=np.uint32)
    end = rs.get_state()
    random.setstate((3, tuple(int(x) for x in end[1]) + (int(end[2]),), state[2]))
    return picks


def generate_snp_table(size, min_maf=0.005):
    """The synthetic SNP list of SnpFactory.random_snp_tuples (pop_factory.py:172-193), sorted like
    pop_factory.py:245, as an SnpTable.  Consumes the global numpy / `random` streams exactly like the reference:
    choice(chromosomes, p) -> choice(mafs, p) -> random(size) -> choice(nucleotides) -> one random.choice per SNP."""
    maf_vals, pdf = genome.load_maf_pdf()
    start = 0
    for i, m in enumerate(maf_vals):  # first index with min_maf <= m, pop_factory.py:161-165
        if min_maf <= m:
            start = i
            break
    chrom_idx = numpy.random.choice(len(genome.CHROMOSOME_LIST), size=size, p=genome.CHROMOSOME_PROB)
    mafs = numpy.random.choice(maf_vals[start:], size=size, p=pdf[start:] * 1 / numpy.sum(pdf[start:]))
    position_randoms = numpy.random.random(size)
    ref_idx = numpy.random.choice(4, size=size)  # same draws as choice(["A","T","C","G"], size)
    others = [[n for n in range(4) if n != r] for r in range(4)]  # remaining_nt after .remove(ref), in list order
    pick = _replay_random_choice3(size)  # one random.choice(remaining_nt) per SNP, pop_factory.py:190
    alt_idx = np.asarray(others, dtype=np.int64)[ref_idx, pick]
    max_pos = np.array([genome.CHROMOSOME_MAX_POSITION[c] for c in genome.CHROMOSOME_LIST], dtype=np.int64)
    position = (position_randoms * max_pos[chrom_idx]).astype(np.int64)  # int(rand * max), pop_factory.py:186
    names = np.array([c.encode() for c in genome.CHROMOSOME_LIST], dtype="S2")
    nt = np.array([c.encode() for c in genome.NUCLEOTIDES], dtype="S1")
    # ordered_snps.sort(key=(chromosome, position)), pop_factory.py:245 -- done on the compact columns (one byte of
    # chromosome rank, the positions), and the table is then built in its final order: at 40 M SNPs the string sort and the
    # re-ordering of a finished table were half of the front end's time
    order = _sort_order(_string_rank(names)[chrom_idx], position)
    chrom_idx, position, ref_idx, alt_idx, mafs = chrom_idx[order], position[order], ref_idx[order], alt_idx[order], mafs[order]
    cum = np.empty((size, 4), dtype=np.float64)
    cum[:, 0] = 1 - mafs
    cum[:, 1:] = 1.0
    ids = order.astype(np.int64) + 1  # rs ids are assigned before the sort
    return SnpTable(ids, names[chrom_idx], position, nt[ref_idx], nt[alt_idx], np.full(size, 2, np.uint8), cum)


def _string_rank(names):
    """rank[i] = position of names[i] in the sorted list of the distinct names (bytewise = Python str order for ASCII)."""
    uniq, inv = np.unique(names, return_inverse=True)
    return inv.astype(np.uint8 if len(uniq) <= 256 else np.int64)


def _sort_order(rank, position):
    """Stable order by (rank, position) = list.sort(key=(chromosome, position)): a radix pass over the one-byte ranks,
    then the positions of every chromosome on their own (numpy's sorts release the GIL: one thread per chromosome)."""
    from concurrent.futures import ThreadPoolExecutor

    if rank.dtype != np.uint8 or len(rank) < 200000:
        return np.lexsort((position, rank))
    by_rank = np.argsort(rank, kind="stable")  # radix sort for 8-bit keys: rows grouped by chromosome, original order inside
    counts = np.bincount(rank, minlength=256)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    out = np.empty(len(rank), dtype=np.int64)

    def one(c):
        lo, hi = int(bounds[c]), int(bounds[c + 1])
        if hi > lo:
            idx = by_rank[lo:hi]
            out[lo:hi] = idx[np.argsort(position[idx], kind="stable")]

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        list(ex.map(one, [c for c in range(256) if counts[c]]))
    return out


def sort_snp_table(tb):
    """ordered_snps.sort(key=(chromosome, position)) -- chromosome compares as a *string* (SURVEY Q9); stable."""
    if len(tb) > 1:  # already in order (a table from generate_snp_table, or a snps.json.gz the reference wrote): nothing to move
        w = tb.chrom.dtype.itemsize
        key = tb.chrom.view(np.uint8).reshape(len(tb), w)
        same = (key[1:] == key[:-1]).all(axis=1)
        less = np.zeros(len(tb) - 1, dtype=bool)
        undecided = np.ones(len(tb) - 1, dtype=bool)
        for k in range(w):  # bytewise string comparison, most significant byte first
            less |= undecided & (key[:-1, k] < key[1:, k])
            undecided &= key[:-1, k] == key[1:, k]
        if bool(np.all(less | (same & (tb.position[:-1] <= tb.position[1:])))):
            return tb
    uniq, inv = np.unique(tb.chrom, return_inverse=True)  # uniq is sorted bytewise = Python str order for ASCII
    order = _sort_order(inv.astype(np.uint8) if len(uniq) <= 256 else inv, tb.position)
    return tb.take(order)


def _float_repr_column(values):
    """repr() of each float64 via the distinct values (MAF tables have < 100 of them)."""
    uniq, inv = np.unique(values, return_inverse=True)
    reprs = np.array([repr(float(u)) for u in uniq])
    return reprs[inv]


def _snps_json_lines_python(tb, b, e):
    lines = []
    ref = tb.ref[b:e].astype("U")
    chrom = tb.chrom[b:e].astype("U")
    alts = [a.split(",") for a in tb.alt[b:e].astype("U")]
    for j in range(e - b):
        na = int(tb.n_alleles[b + j])
        tuples = {ref[j]: float(tb.cum[b + j, 0])}
        for k in range(1, na):
            tuples[alts[j][k - 1]] = float(tb.cum[b + j, k])
        lines.append(json.dumps({"id": int(tb.snp_id[b + j]), "chromosome": chrom[j],
                                 "position": int(tb.position[b + j]), "tuples": tuples}))
    return ("\n".join(lines) + "\n").encode() if lines else b""


def _gzip_member(data, level):
    return gzip.compress(data, compresslevel=level, mtime=0)


def write_snps_file(tb, path, threads=8):
    """snps.json.gz: one `str(SNPTuples)` per line (pop_factory.py:118-124, :258-262), gzip level 5.  Lines are
    formatted by the library's C helper in slabs and compressed as independent gzip members on a thread pool
    (a multi-member file is what `gzip.open(...).read()` and the reference's load_snps_file read transparently)."""
    from concurrent.futures import ThreadPoolExecutor

    from . import capi

    n = len(tb)
    step = 250000
    with open(path, "wb") as f, ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        pending = []
        for b in range(0, max(n, 1), step):
            e = min(n, b + step)
            blob = capi.format_snps_json(tb.snp_id[b:e], tb.chrom[b:e], tb.position[b:e], tb.ref[b:e], tb.alt[b:e],
                                         tb.n_alleles[b:e], tb.cum[b:e]) if e > b else np.zeros(0, np.uint8)
            data = _snps_json_lines_python(tb, b, e) if blob is None else blob.tobytes()
            pending.append(pool.submit(_gzip_member, data, 5))
            while len(pending) > 2 * threads:
                f.write(pending.pop(0).result())
        for fut in pending:
            f.write(fut.result())


def load_snps_file(path):
    """Read snps.json.gz (SNPTuples.from_json, pop_factory.py:126-133, :264-272): key order of "tuples" is allele
    order.  Files in the layout dna-factory writes go through the library's C parser; anything else through json."""
    from . import capi

    with gzip.open(path, "rb") as f:
        text = f.read()
    cols = capi.parse_snps_json(text)
    if cols is not None:
        sid, chrom, pos, alleles, cum, na = cols
        if len(sid) and (na.min() < 1):
            cols = None  # a record without alleles: let from_snps report it
    if cols is not None:
        sid, chrom, pos, alleles, cum, na = cols
        ref = alleles[:, 0]
        alt = np.where(na == 1, alleles[:, 0], alleles[:, 1]).astype("S64")
        for j in np.flatnonzero(na > 2):  # ",".join of the other alleles (alt_alleles(), pop_factory.py:111-116)
            alt[j] = b",".join(alleles[j, 1: na[j]])
        cum = cum.copy()
        for k in range(4):
            cum[na <= k, k] = 1.0
        w = max([1] + [len(x) for x in alt[na > 2]]) if np.any(na > 2) else 1
        w = max(w, int(np.max(np.char.str_len(alt))) if len(alt) else 1)
        return SnpTable(sid, chrom, pos, ref, alt.astype("S%d" % w), na, cum)
    snps = []
    for line in text.decode().splitlines():
        if not line:
            continue
        o = json.loads(line)
        snps.append((o["id"], o["chromosome"], o["position"], list(o.get("tuples", {}).items())))
    return SnpTable.from_snps(snps)


# ---------------------------------------------------------------------------------------------------------
# Deleterious groups (pop_factory.py:541-635)
# ---------------------------------------------------------------------------------------------------------


class DeleteriousGroup:
    def __init__(self, name, population_weight):
        self.deleterious = {}  # snp id -> mutation weight, in pick order
        self.name = name
        self.population_weight = population_weight

    def to_json(self):
        return json.dumps({"deleterious": {str(k): v for k, v in self.deleterious.items()}, "name": self.name,
                           "population_weight": self.population_weight})

    @classmethod
    def from_json(cls, line, int_keys=True):
        """int_keys=False keeps the reference's behaviour (pop_factory.py:613-619): keys stay strings and never
        match an int SNP id, so nothing is forced."""
        o = json.loads(line)
        g = cls(o["name"], o["population_weight"])
        for k, w in o["deleterious"].items():
            g.deleterious[int(k) if int_keys else k] = w
        return g

    def select_mutations(self):
        """Shuffle, then take mutations until their weights reach 1 (pop_factory.py:621-635)."""
        items = list(self.deleterious.items())
        random.shuffle(items)
        picked, agg = {}, 0
        for k, w in items:
            picked[k] = w
            agg += w
            if agg >= 1:
                break
        return picked


def _filtered_ids(tb, lo, hi):
    """snp_ids_from_list (pop_factory.py:549-558): filter on (minor.cum - ref.cum) only if min > 0 or max < 0.5."""
    if lo > 0 or hi < 0.5:
        diff = tb.cum[:, 1] - tb.cum[:, 0]
        keep = (lo <= diff) & (diff <= hi)
        return tb.snp_id[keep]
    return tb.snp_id


def pick_deleterious(tb, config_path):
    """YAML config -> ordered {name: DeleteriousGroup} (pick_deleterious_snps / from_yml / init_with_snps,
    pop_factory.py:521-538, :560-608).  One numpy.random.choice(ids, k, replace=False) per group instance."""
    import yaml

    from . import capi

    with open(config_path, "r") as f:
        cfg = yaml.safe_load(f)
    groups = {}
    for name, attr in cfg.items():
        lo, hi = 0, 1
        if attr.get("min_minor_allele_freq"):
            if not 0 < attr["min_minor_allele_freq"] < 0.5:
                raise Exception("min_minor_allele_freq must be between 0 and 0.5. yml value = {}".format(attr["min_minor_allele_freq"]))
            lo = attr["min_minor_allele_freq"]
        if attr.get("max_minor_allele_freq"):
            if not 0 < attr["max_minor_allele_freq"] < 0.5:
                raise Exception("max_minor_allele_freq must be between 0 and 0.5. yml value = {}".format(attr["max_minor_allele_freq"]))
            hi = attr["max_minor_allele_freq"]
        n_inst = int(attr["num_instances"]) if attr.get("num_instances") else 1
        ids = _filtered_ids(tb, lo, hi)
        weights = attr["mutation_weights"]
        for i in range(n_inst):
            g = DeleteriousGroup("%s-%s" % (name, i), attr["population_weight"])
            if len(ids) == 0:
                raise Exception("No SNPs in list to choose from. SNPs must have all been filtered out by deleterious config.")
            # numpy.random.choice(a=ids, size=k, replace=False) = ids[permutation(len(ids))[:k]] on the legacy generator
            # (pop_factory.py:576); the library replays its draws without shuffling 40 M ids fourteen times
            for w, sid in zip(weights, ids[capi.numpy_legacy_choice(len(ids), len(weights))]):
                g.deleterious[int(sid)] = w
            groups[g.name] = g
    return groups


def write_deleterious_json(groups, path):
    with open(path, "w") as f:
        for g in groups.values():
            f.write(g.to_json() + "\n")


def load_deleterious_json(path, int_keys=True):
    groups = {}
    with open(path, "rt") as f:
        for line in f:
            g = DeleteriousGroup.from_json(line, int_keys=int_keys)
            groups[g.name] = g
    return groups


# ---------------------------------------------------------------------------------------------------------
# Sample table, FAM and side files (pop_factory.py:333-383)
# ---------------------------------------------------------------------------------------------------------


def build_samples(groups, control_size, test_size, male_odds, outdir, offset=0):
    """pick_deleterious_groups (random.choices) -> generate_fam_file (numpy rand for sex; select_mutations per case);
    writes population.fam and pop_deleterious.txt; returns (SampleTable, person ids)."""
    glist = list(groups.values())
    chosen = random.choices(population=glist, weights=[g.population_weight for g in glist], k=test_size) if test_size else []
    n = control_size + test_size
    randoms = numpy.random.rand(n)
    sex = np.where(randoms <= male_odds, 1, 2)
    person = np.empty(n, dtype=np.int64)
    person[:control_size] = 100000 + offset + np.arange(1, control_size + 1)
    person[control_size:] = 500000 + offset + np.arange(1, test_size + 1)
    carried = [None] * n
    with open(os.path.join(outdir, "pop_deleterious.txt"), "w") as pp:
        for j in range(test_size):
            picked = chosen[j].select_mutations()
            carried[control_size + j] = picked
            pp.write("%i\t%s\t" % (person[control_size + j], chosen[j].name) + "\t".join("rs" + str(k) for k in picked) + "\n")
    with open(os.path.join(outdir, "population.fam"), "w") as f:
        for i in range(n):
            f.write("%i\t%i\t%i\t%i\t%i\t%i\t\n" % (i + 1 + offset * 2, person[i], 0, 0, sex[i], 1 if i < control_size else 2))
    st = SampleTable.from_samples((int(sex[i]), i < control_size, carried[i]) for i in range(n))
    st.person_id = person
    return st


def write_bim(tb, path):
    """PLINK .bim for the rows of population.bed: chromosome, rs id, 0 cM, position, A1 = ALT (the first one of a
    multi-allelic row), A2 = REF -- the allele order dnaf_export_bed codes against."""
    chrom = np.char.decode(tb.chrom, "ascii")
    ref = np.char.decode(tb.ref, "ascii")
    alt = np.char.partition(np.char.decode(tb.alt, "ascii"), ",")[:, 0]
    with open(path, "w") as f:
        step = 200000
        for b in range(0, len(tb), step):
            e = min(len(tb), b + step)
            f.write("".join("%s\trs%d\t0\t%d\t%s\t%s\n" % t for t in zip(chrom[b:e], tb.snp_id[b:e].tolist(), tb.position[b:e].tolist(),
                                                                         alt[b:e], ref[b:e])))


def vcf_header(person_ids, now=None):
    """gen_vcf_header, pop_factory.py:36-44."""
    now = now or datetime.now()
    return ("##fileformat=VCFv4.3\n"
            "##filedate=%s\n" % now.strftime("%Y%m%d %H:%M")
            + "##source=PopFactory\n"
            + '##FILTER=<ID=q10,Description="Quality below 10">\n'
            + '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
            + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t"
            + "\t".join(str(int(p)) for p in person_ids) + "\n")
