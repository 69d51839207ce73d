#!/usr/bin/env python3
"""Generate tests/golden/*.json.gz by running the REAL reference worker.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs
/root/reference, which never travels to the GPU box).  The reference is
imported unmodified from where it lies; `Bio.bgzf` (biopython is not
installed here) is replaced by an empty stub module because
pop_factory.py:13 imports it at module level -- the hot path
(pop_factory.py:471-513) never touches it.

Two kinds of expected output are recorded per case:
  lines_mt      the unmodified worker after numpy.random.seed(seed)
  lines_philox  the same worker with numpy.random.rand replaced by the
                Philox4x32-10 draw of DESIGN.md (row r, draw n -> word n&3 of
                philox(ctr=(n>>2,0,r_lo,r_hi), key=seed) scaled by 2^-32)

A fixture is data only: inputs (SNP tuples, sample table) and the bytes the
reference produced for them.
"""
import gzip
import json
import os
import random
import sys
import tempfile
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("DNAF_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402  (numpy Philox draw function)


def import_reference():
    sys.dont_write_bytecode = True
    bio = types.ModuleType("Bio")
    bgzf = types.ModuleType("Bio.bgzf")
    bgzf.BgzfWriter = object  # never instantiated by the hot path
    bio.bgzf = bgzf
    sys.modules["Bio"] = bio
    sys.modules["Bio.bgzf"] = bgzf
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import pop_factory  # the reference module

    pop_factory.time.sleep = lambda *_a, **_k: None  # worker's exit sleep, pop_factory.py:512
    return pop_factory


class FakePipe:
    def __init__(self):
        self.msgs = []

    def send(self, m):
        self.msgs.append(m)

    def close(self):
        pass


def run_worker(pf_mod, fam_data, snps):
    pf = pf_mod.PopulationFactory(output_path=tempfile.gettempdir())
    pipe = FakePipe()
    pf.queue_vcf_snps(fam_data, list(enumerate(snps, 1)), pipe)
    assert pipe.msgs[-1] == "DONE"
    nums = [m[0] for m in pipe.msgs[:-1]]
    assert nums == list(range(1, len(snps) + 1))
    return [m[1] for m in pipe.msgs[:-1]]


def run_worker_mt(pf_mod, fam_data, snps, seed):
    numpy.random.seed(seed)
    return run_worker(pf_mod, fam_data, snps)


def run_worker_philox(pf_mod, fam_data, snps, seed, row0=0):
    state = {"row": row0}
    real_rand = numpy.random.rand

    def philox_rand(n):
        u = orc.philox_row_uniforms(seed, state["row"], n)
        state["row"] += 1
        return u

    pf_mod.numpy.random.rand = philox_rand
    try:
        return run_worker(pf_mod, fam_data, snps)
    finally:
        pf_mod.numpy.random.rand = real_rand


def make_snp(pf_mod, sid, chrom, pos, tuples):
    s = pf_mod.SNPTuples(sid, chrom, pos)
    for letter, cum in tuples:
        s.add_tuple(letter, cum)
    return s


def make_samples(pf_mod, spec):
    """spec: list of (iid, sex, is_control, carried dict or None)"""
    return [pf_mod.SampleInfo(i + 1, iid, 0, 0, sex, ctrl, carried) for i, (iid, sex, ctrl, carried) in enumerate(spec)]


def dump(name, snps, spec, seed, lines_mt, lines_philox, extra=None):
    doc = {
        "name": name,
        "seed": seed,
        "snps": [[s.id, s.chromosome, s.position, [[t[0], float(t[1])] for t in s.tuples]] for s in snps],
        "samples": [[iid, sex, bool(ctrl), (sorted(int(k) for k in carried) if carried is not None else None)]
                    for (iid, sex, ctrl, carried) in spec],
        "lines_mt": lines_mt,
        "lines_philox": lines_philox,
    }
    if extra:
        doc.update(extra)
    path = os.path.join(GOLD, name + ".json.gz")
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    print("wrote %s (%d rows x %d samples, %d bytes)" % (path, len(snps), len(spec), os.path.getsize(path)))


def case_mixed(pf_mod, name, n_rows, S, seed, rs):
    """All chromosome classes, both sexes, controls/cases, forced cells (incl. female-on-Y), multi-allelic rows."""
    chroms = [str(c) for c in range(1, 23)] + ["X", "Y", "MT"]
    letters = ["A", "C", "G", "T"]
    snps = []
    for j in range(n_rows):
        chrom = chroms[j % len(chroms)] if j < 2 * len(chroms) else chroms[rs.randint(0, len(chroms))]
        kind = j % 11
        perm = list(rs.permutation(letters))
        maf = 0.005 * rs.randint(1, 100)
        if kind == 7:  # three alleles
            tuples = [(perm[0], 0.70), (perm[1], 0.90), (perm[2], 1.0)]
        elif kind == 9:  # four alleles
            tuples = [(perm[0], 0.55), (perm[1], 0.80), (perm[2], 0.93), (perm[3], 1.0)]
        elif kind == 10 and j % 22 == 10:  # single tuple: ALT printed as REF
            tuples = [(perm[0], 1.0)]
        else:
            tuples = [(perm[0], 1 - maf), (perm[1], 1.0)]
        sid = 1000 + ((j * 7919) % 100003)  # ids not monotone in row order
        snps.append(make_snp(pf_mod, sid, chrom, int(rs.randint(0, 250000000)), tuples))
    ids = [s.id for s in snps]
    y_ids = [s.id for s in snps if s.chromosome == "Y"]
    n_ctrl = S // 2
    spec = []
    for i in range(S):
        ctrl = i < n_ctrl
        sex = 1 if rs.rand() <= 0.5 else 2
        carried = None
        if not ctrl:
            k = int(rs.randint(0, 6))
            chosen = [int(x) for x in rs.choice(ids, size=k, replace=False)] if k else []
            if y_ids and i % 3 == 0:
                chosen.append(int(y_ids[i % len(y_ids)]))  # female case "carrying" a Y SNP -> '.' wins (Q7)
            carried = {c: 0.5 for c in chosen}
        spec.append((100001 + i if ctrl else 500001 + i, sex, ctrl, carried))
    fam = make_samples(pf_mod, spec)
    dump(name, snps, spec, seed, run_worker_mt(pf_mod, fam, snps, seed), run_worker_philox(pf_mod, fam, snps, seed))


def case_headline_shape(pf_mod, name="headline_10x20000", S=20000, seed=20261004, chroms=("1", "2", "X", "3", "Y", "4", "MT", "5", "22", "X"), rs_seed=5):
    """Ten rows at the headline shape (20,000 samples: one row per wave in the line-owning GPU kernel, whose waves also
    write the first bytes of the next wave's row), every chromosome class among them, a three-allele row, forced cells.
    Also called for four rows at BASELINE config 5's 50,000 samples (rows of seven 2,048-group segments)."""
    rs = numpy.random.RandomState(rs_seed)
    chroms = list(chroms)
    letters = ["A", "C", "G", "T"]
    snps = []
    for j, chrom in enumerate(chroms):
        perm = list(rs.permutation(letters))
        maf = 0.005 * rs.randint(1, 100)
        tuples = [(perm[0], 0.70), (perm[1], 0.90), (perm[2], 1.0)] if j == 7 or (len(chroms) < 8 and j == 1) else [(perm[0], 1 - maf), (perm[1], 1.0)]
        snps.append(make_snp(pf_mod, 7000 + 13 * j, chrom, int(rs.randint(0, 250000000)), tuples))
    ids = [s.id for s in snps]
    n_ctrl = S // 2
    spec = []
    for i in range(S):
        ctrl = i < n_ctrl
        sex = 1 if rs.rand() <= 0.5 else 2
        carried = None
        if not ctrl:
            carried = {int(ids[k % len(ids)]): 0.5 for k in ((1, 4, 9) if i % 97 == 0 else (3,) if i % 389 == 0 else ())}
        spec.append((100001 + i if ctrl else 500001 + i, sex, ctrl, carried))
    fam = make_samples(pf_mod, spec)
    dump(name, snps, spec, seed, run_worker_mt(pf_mod, fam, snps, seed), run_worker_philox(pf_mod, fam, snps, seed))


def case_threshold_equality(pf_mod):
    """Rolls exactly equal to a threshold pick the LOWER index (`>=`), in both RNG modes."""
    S, seed = 12, 4242
    spec = [(100001 + i, 1 + (i % 2), True, None) for i in range(S)]
    fam = make_samples(pf_mod, spec)
    # MT mode: thresholds taken from the very doubles the worker will draw
    numpy.random.seed(seed)
    d = [numpy.random.rand(2 * S) for _ in range(4)]
    snps_mt = [
        make_snp(pf_mod, 1, "1", 100, [("A", float(d[0][6])), ("C", 1.0)]),
        make_snp(pf_mod, 2, "2", 200, [("G", float(d[1][9])), ("T", 1.0)]),
        make_snp(pf_mod, 3, "3", 300, [("A", float(numpy.nextafter(d[2][4], 0.0))), ("T", 1.0)]),
        make_snp(pf_mod, 4, "4", 400, [("A", float(d[3][0])), ("C", float(d[3][1])), ("G", 1.0)]),
    ]
    lines_mt = run_worker_mt(pf_mod, fam, snps_mt, seed)
    p = [orc.philox_row_uniforms(seed, r, 2 * S) for r in range(4)]
    snps_ph = [
        make_snp(pf_mod, 1, "1", 100, [("A", float(p[0][6])), ("C", 1.0)]),
        make_snp(pf_mod, 2, "2", 200, [("G", float(p[1][9])), ("T", 1.0)]),
        make_snp(pf_mod, 3, "3", 300, [("A", float(numpy.nextafter(p[2][4], 0.0))), ("T", 1.0)]),
        make_snp(pf_mod, 4, "4", 400, [("A", float(p[3][0])), ("C", float(p[3][1])), ("G", 1.0)]),
    ]
    lines_ph = run_worker_philox(pf_mod, fam, snps_ph, seed)
    dump("threshold_equality_mt", snps_mt, spec, seed, lines_mt, None)
    dump("threshold_equality_philox", snps_ph, spec, seed, None, lines_ph)


def case_ragged(pf_mod):
    """Sample counts around every packing boundary, including S = 0 (line = prefix + newline)."""
    rs = numpy.random.RandomState(99)
    for S in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 63, 64, 65, 255, 257):
        snps = [
            make_snp(pf_mod, 11, "1", 12345, [("A", 0.8), ("G", 1.0)]),
            make_snp(pf_mod, 12, "X", 777, [("C", 0.6), ("T", 1.0)]),
            make_snp(pf_mod, 13, "Y", 31, [("G", 0.5), ("A", 1.0)]),
            make_snp(pf_mod, 14, "MT", 9, [("T", 0.9), ("C", 1.0)]),
            make_snp(pf_mod, 15, "22", 50806828, [("T", 0.3), ("C", 0.6), ("A", 1.0)]),
        ]
        spec = []
        for i in range(S):
            ctrl = i % 2 == 0
            spec.append((100001 + i, 1 if rs.rand() <= 0.5 else 2, ctrl, None if ctrl else ({12: 1.0, 13: 1.0} if i % 4 == 1 else {})))
        fam = make_samples(pf_mod, spec)
        seed = 7 + S
        dump("ragged_S%d" % S, snps, spec, seed, run_worker_mt(pf_mod, fam, snps, seed),
             run_worker_philox(pf_mod, fam, snps, seed))


def case_full_flow(pf_mod):
    """The reference's own feeders end to end (SnpFactory -> sort -> deleterious.yml picks -> fam -> worker),
    recorded so the host side can be checked against it: SNP list, deleterious.json lines, fam lines,
    pop_deleterious.txt, header, numpy MT state at worker entry, and the body lines."""
    tmp = tempfile.mkdtemp(prefix="dnaf_golden_")
    numpy.random.seed(20261)
    random.seed(97531)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        pf = pf_mod.PopulationFactory(num_processes=1, generate_snps=True, male_odds=0.5,
                                      deleterious_config=os.path.join(REF, "deleterious.yml"), output_path=tmp)
        fac = pf_mod.SnpFactory.init_from_cdf_file()
        pf.ordered_snps = fac.random_snp_tuples(400, min_maf=0.01)
        pf.ordered_snps.sort(key=lambda x: (x.chromosome, x.position))
        pf.pick_deleterious_snps(pf.ordered_snps, pf.deleterious_config)
        groups = pf_mod.PopulationFactory.pick_deleterious_groups(list(pf.deleterious.values()), 60)
        fam = pf.generate_fam_file(60, 60, 0.5, groups)
    finally:
        os.chdir(cwd)
    st = numpy.random.get_state()
    assert st[0] == "MT19937"
    lines_mt = run_worker(pf_mod, fam, pf.ordered_snps)  # continues from the state above
    spec = [(s.person_id, s.sex, s.is_control, s.deleterious_snps) for s in fam]
    lines_ph = run_worker_philox(pf_mod, fam, pf.ordered_snps, 20261)
    header = pf_mod.gen_vcf_header(fam).split("\n")
    header[1] = "##filedate=FROZEN"
    extra = {
        "mt_state_key": [int(x) for x in st[1]],
        "mt_state_pos": int(st[2]),
        "deleterious_json": open(os.path.join(tmp, "deleterious.json")).read(),
        "population_fam": open(os.path.join(tmp, "population.fam")).read(),
        "pop_deleterious_txt": open(os.path.join(tmp, "pop_deleterious.txt")).read(),
        "vcf_header": "\n".join(header),
        "numpy_seed": 20261,
        "python_random_seed": 97531,
    }
    dump("full_flow_120x400", pf.ordered_snps, spec, 20261, lines_mt, lines_ph, extra)


def run_reference_cli(pf_mod, tag, n_controls, n_cases, max_snps, yml_text=None, num_processes=1, min_freq=0.01,
                      snps_path=None, deleterious_list_path=None, offset=None):
    """One run of the UNMODIFIED reference's `main` flow -- PopulationFactory(...).generate_population(...), exactly what
    pop_factory.py:673-693 does with the parsed flags -- with the three things a test cannot otherwise fix pinned from
    outside: the wall clock the reference takes its numpy seed and its ##filedate from (pop_factory.datetime frozen at
    12:34:56 -> numpy.random.seed(123456), pop_factory.py:235), python's `random` (seeded with the same number; the
    reference never seeds it), and Bio.bgzf.BgzfWriter (biopython is not installed: a stand-in that hashes the text it
    is given -- parity is on the decompressed stream).  -> (output directory, dict of SHA-256 digests)."""
    import datetime as real_datetime
    import hashlib

    class FrozenDatetime(real_datetime.datetime):
        @classmethod
        def now(cls, tz=None):
            return cls(2026, 10, 4, 12, 34, 56)

    class HashingWriter:
        digest = None
        n_lines = 0
        n_bytes = 0
        first_body_line = None

        def __init__(self, filename=None, mode="w", fileobj=None, compresslevel=6):
            self.h = hashlib.sha256()

        def write(self, text):
            data = text.encode() if isinstance(text, str) else text
            for line in data.splitlines(keepends=True):
                if line.startswith(b"##filedate="):
                    continue  # wall clock
                if not line.startswith(b"#") and HashingWriter.first_body_line is None:
                    HashingWriter.first_body_line = line[:120].decode()
                self.h.update(line)
                HashingWriter.n_lines += 1
                HashingWriter.n_bytes += len(line)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            HashingWriter.digest = self.h.hexdigest()
            return False

    seed = 123456
    tmp = tempfile.mkdtemp(prefix="dnaf_%s_" % tag)
    old_dt, old_writer = pf_mod.datetime, pf_mod.bgzf.BgzfWriter
    pf_mod.datetime = FrozenDatetime
    pf_mod.bgzf.BgzfWriter = HashingWriter
    random.seed(seed)
    cwd = os.getcwd()
    os.chdir(REF)  # -p deleterious.yml is cwd-relative
    yml = "deleterious.yml"
    if yml_text is not None:  # a deleterious-group file of this case's own (kept in the fixture, the test writes it out again)
        yml = os.path.join(tmp, "groups.yml")
        with open(yml, "w") as f:
            f.write(yml_text)
    try:
        pf = pf_mod.PopulationFactory(num_processes=num_processes, generate_snps=True, male_odds=0.5, deleterious_config=yml,
                                      deleterious_list_path=deleterious_list_path, sample_id_offset=offset,
                                      snps_path=snps_path, output_path=tmp)
        pf.generate_population(n_controls, n_cases, min_freq, max_snps, 6)
    finally:
        os.chdir(cwd)
        pf_mod.datetime, pf_mod.bgzf.BgzfWriter = old_dt, old_writer

    def sha(path, opener=open):
        with opener(path, "rb") as f:
            data = f.read()
        return {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}

    flags = "-s %d -c %d" % (n_cases, n_controls)
    if max_snps is not None:
        flags += " -x %d" % max_snps
    flags += " -f %s -n %d" % (min_freq, num_processes)
    if yml_text is not None:
        flags += " -p groups.yml"
    if snps_path:
        flags += " --snps_file <step 1>/snps.json.gz"
    if deleterious_list_path:
        flags += " --deleterious_file <step 1>/deleterious.json"
    if offset:
        flags += " --offset %d" % offset
    doc = {
        "command": "pop_factory.py %s  (reference, unmodified; clock frozen at 12:34:56)" % flags,
        "numpy_seed": seed, "python_random_seed": seed,
        "vcf_without_filedate": {"sha256": HashingWriter.digest, "bytes": HashingWriter.n_bytes, "lines": HashingWriter.n_lines,
                                 "first_body_line_head": HashingWriter.first_body_line},
        "population_fam": sha(os.path.join(tmp, "population.fam")),
        "pop_deleterious_txt": sha(os.path.join(tmp, "pop_deleterious.txt")),
    }
    # written only when the run picked / generated them itself (pop_factory.py:246-250)
    if os.path.exists(os.path.join(tmp, "deleterious.json")):
        doc["deleterious_json"] = sha(os.path.join(tmp, "deleterious.json"))
    if os.path.exists(os.path.join(tmp, "snps.json.gz")):
        doc["snps_json_decompressed"] = sha(os.path.join(tmp, "snps.json.gz"), gzip.open)
    if yml_text is not None:
        doc["deleterious_yml"] = yml_text
    return tmp, doc


def case_config1(pf_mod, name="config1", n_controls=100, n_cases=100, max_snps=100000, yml_text=None, num_processes=1):
    """BASELINE config 1 at its stated size through the UNMODIFIED reference: `pop_factory.py -s 100 -c 100 -x 100000
    -f 0.01 -n 1` (run_reference_cli above).  With other arguments: BASELINE config 2 (`-s 1000 -c 1000 -x 1000000`,
    about an hour of the reference's worker), configs 3 / 5 at 1 % of their SNPs, a run that crosses the reference's
    2M-SNP chunking line (`chunks_2000001`: two work chunks, the stream restarts, pop_factory.py:402-411), and the
    reference's DEFAULT worker count (`workers2_*`: -n 2, both forked workers draw the same stream).
    Only SHA-256 digests and sizes are stored (the VCF body alone is 83 MB)."""
    _tmp, doc = run_reference_cli(pf_mod, name, n_controls, n_cases, max_snps, yml_text, num_processes)
    path = os.path.join(GOLD, "%s_sha256.json" % name)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote %s: %s" % (path, json.dumps(doc["vcf_without_filedate"])))


def case_workflow(pf_mod, name="workflow", num_processes=1, n_snps=20000):
    """The reference README's multi-job workflow (README.md:69-103) through the unmodified reference, three runs:
      step 1  -s 0 -c 0 -x 20000 -f 0.01            selects SNPs and deleterious SNPs, "empty" VCF (lines without cells)
      step 2  -s 50 -c 50 --snps_file <1>/snps.json.gz --deleterious_file <1>/deleterious.json
      step 3  the same with --offset 1000
    (pop_factory.py:237-250 load paths, :264-272, :515-519 / :613-619 with its string-key quirk SURVEY Q8, :341-383 ids).
    `num_processes=2` records the same three commands at the reference's default -n (README's literal commands)."""
    out = {}
    d1, out["step1"] = run_reference_cli(pf_mod, name + "1", 0, 0, n_snps, num_processes=num_processes)
    snps, dele = os.path.join(d1, "snps.json.gz"), os.path.join(d1, "deleterious.json")
    # -x absent, -f at its default 0.005 (pop_factory.py:644-648): neither is used when the SNPs are loaded
    _d, out["step2"] = run_reference_cli(pf_mod, name + "2", 50, 50, None, num_processes=num_processes, min_freq=0.005,
                                         snps_path=snps, deleterious_list_path=dele)
    _d, out["step3"] = run_reference_cli(pf_mod, name + "3", 50, 50, None, num_processes=num_processes, min_freq=0.005,
                                         snps_path=snps, deleterious_list_path=dele, offset=1000)
    path = os.path.join(GOLD, "%s_sha256.json" % name)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s: %s" % (path, json.dumps({k: v["vcf_without_filedate"]["sha256"][:12] for k, v in out.items()})))


# four deleterious groups for the config-5-shaped run (polygenic stress): different sizes, weights and frequency windows
FOUR_GROUPS_YML = """\
stress_a:
  mutation_weights: [1, 1]
  num_instances: 1
  population_weight: 1
stress_b:
  mutation_weights: [0.5, 0.5, 0.5]
  num_instances: 1
  population_weight: 2
  max_minor_allele_freq: 0.30
stress_c:
  mutation_weights: [0.2, 0.2, 0.3, 0.3, 0.4]
  num_instances: 1
  population_weight: 1
  min_minor_allele_freq: 0.05
stress_d:
  mutation_weights: [1]
  num_instances: 1
  population_weight: 1
"""


def main():
    os.makedirs(GOLD, exist_ok=True)
    pf_mod = import_reference()
    if "--only-config1" in sys.argv:
        case_config1(pf_mod)
        return
    if "--only-config2" in sys.argv:  # not part of a default run: an hour
        case_config1(pf_mod, "config2", 1000, 1000, 1000000)
        return
    if "--only-config5-1pct" in sys.argv:  # BASELINE config 5's 50,000 samples and four deleterious groups at 1 % of its SNPs
        case_config1(pf_mod, "config5_1pct", 25000, 25000, 10000, FOUR_GROUPS_YML)
        return
    if "--only-config3-1pct" in sys.argv:  # BASELINE config 3's samples and flow at 1 % of its SNPs (the whole would be 40 h)
        case_config1(pf_mod, "config3_1pct", 10000, 10000, 50000)
        return
    if "--only-chunks" in sys.argv:  # 2,000,001 SNPs x 4 samples: two of the reference's work chunks (a few minutes)
        case_config1(pf_mod, "chunks_2000001", 2, 2, 2000001)
        return
    if "--only-workflow" in sys.argv:  # README.md:69-103, at -n 1 and at the reference's default -n 2
        case_workflow(pf_mod, "workflow", 1)
        case_workflow(pf_mod, "workflow_n2", 2)
        return
    if "--only-workers" in sys.argv:  # the reference's default -n 2 (and -n 3): forked workers share one stream
        case_config1(pf_mod, "workers2_100x20000", 50, 50, 20000, num_processes=2)
        case_config1(pf_mod, "workers3_100x20001", 50, 50, 20001, num_processes=3)
        return
    if "--only-long-rows" in sys.argv:  # rows long enough for the line-owning GPU kernel (S >= 2,033), several rows per wave
        case_mixed(pf_mod, "mixed_60x2100", 60, 2100, 424242, numpy.random.RandomState(3))
        case_mixed(pf_mod, "mixed_90x1000", 90, 1000, 31337, numpy.random.RandomState(4))  # short-row kernel, four trips per row
        case_headline_shape(pf_mod)
        case_headline_shape(pf_mod, "config5_shape_4x50000", 50000, 77, ("7", "X", "19", "MT"), 6)
        return
    # the reference's own known answers, test/unit/pop_factory_test.py:24-28
    t = make_snp(pf_mod, 100, "1", 50000, [("G", 0.70), ("A", 0.90), ("T", 1.0)])
    assert t.pick_allele_index(0.95) == 2 and t.pick_allele_index(0.4) == 0
    case_mixed(pf_mod, "mixed_64x97", 64, 97, 12345, numpy.random.RandomState(1))
    case_mixed(pf_mod, "mixed_150x260", 150, 260, 86399, numpy.random.RandomState(2))
    case_mixed(pf_mod, "mixed_60x2100", 60, 2100, 424242, numpy.random.RandomState(3))
    case_mixed(pf_mod, "mixed_90x1000", 90, 1000, 31337, numpy.random.RandomState(4))
    case_headline_shape(pf_mod)
    case_headline_shape(pf_mod, "config5_shape_4x50000", 50000, 77, ("7", "X", "19", "MT"), 6)
    case_threshold_equality(pf_mod)
    case_ragged(pf_mod)
    case_full_flow(pf_mod)
    if "--skip-config1" not in sys.argv:
        case_config1(pf_mod)  # about a minute: the reference's own worker over 2x10^7 genotypes


if __name__ == "__main__":
    main()
