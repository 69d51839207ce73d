"""Command line: flag surface (the reference's own parse test) and, on a GPU, the output directory end to end."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as ge

    ge.build()
    from dna_factory_amd import cli

    return cli


def test_parse_args_like_the_reference(cli):
    # test/unit/pop_factory_test.py:80-104, verbatim expectations
    cmd = ("-s 10 -c 20 -n 5 -z 3 -p path_config.yml -f 0.1 -m 0.7 -x 2500 -l "
           "--deleterious_file /home/ochrzan/workspace/deleterious.json "
           "--offset 300 --snps_file my_snps.json --outdir myoutput/tuesday")
    a = cli.parse_cmd_args(cmd.split(" "))
    assert (a.size, a.control_size, a.num_processes, a.compression_level) == (10, 20, 5, 3)
    assert (a.deleterious_config, a.min_freq, a.male_odds, a.max_snps) == ("path_config.yml", 0.1, 0.7, 2500)
    assert a.generate_snps is False
    assert a.deleterious_file == "/home/ochrzan/workspace/deleterious.json"
    assert (a.offset, a.snps_file, a.outdir) == (300, "my_snps.json", "myoutput/tuesday")
    a = cli.parse_cmd_args("-s 10 -c 20 -x 2500".split(" "))
    assert a.deleterious_file is None
    # defaults of pop_factory.py:646-659
    assert (a.deleterious_config, a.min_freq, a.male_odds, a.num_processes, a.compression_level, a.generate_snps) == (
        "deleterious.yml", 0.005, 0.5, 2, 6, True)
    with pytest.raises(SystemExit):
        cli.parse_cmd_args("-s 10 -c 20 -z 10".split(" "))
    with pytest.raises(SystemExit):
        cli.parse_cmd_args("-c 20".split(" "))


def test_args_from_file(cli, tmp_path):
    f = tmp_path / "args.txt"
    f.write_text("-s\n5\n-c\n6\n-x\n77\n")
    a = cli.parse_cmd_args(["@" + str(f)])  # fromfile_prefix_chars='@', pop_factory.py:639
    assert (a.size, a.control_size, a.max_snps) == (5, 6, 77)


def _run_cli(args, cwd=REPO):
    env = dict(os.environ, PYTHONPATH=REPO)
    return subprocess.run([sys.executable, os.path.join(REPO, "pop_factory.py")] + args, cwd=cwd, env=env,
                          capture_output=True, text=True, timeout=600)


def _expected_body_inputs(outdir):
    """(None, samples) rebuilt from population.fam and pop_deleterious.txt."""
    carried = {}
    for line in open(os.path.join(outdir, "pop_deleterious.txt")):
        parts = line.rstrip("\n").split("\t")
        carried[int(parts[0])] = [int(x[2:]) for x in parts[2:] if x]
    samples = []
    for line in open(os.path.join(outdir, "population.fam")):
        fid, iid, _, _, sex, pheno = line.split("\t")[:6]
        ctrl = pheno == "1"
        samples.append((int(sex), ctrl, None if ctrl else carried[int(iid)]))
    return None, samples


def _expected_body(outdir, seed, orc):
    """Rebuild the worker's inputs from the side files the run wrote, and run the CPU oracle on them."""
    snps = []
    for line in gzip.open(os.path.join(outdir, "snps.json.gz"), "rt"):
        o = json.loads(line)
        snps.append((o["id"], o["chromosome"], o["position"], list(o["tuples"].items())))
    carried = {}
    for line in open(os.path.join(outdir, "pop_deleterious.txt")):
        parts = line.rstrip("\n").split("\t")
        carried[int(parts[0])] = [int(x[2:]) for x in parts[2:] if x]
    samples = []
    for line in open(os.path.join(outdir, "population.fam")):
        fid, iid, _, _, sex, pheno = line.split("\t")[:6]
        ctrl = pheno == "1"
        samples.append((int(sex), ctrl, None if ctrl else carried[int(iid)]))
    rec = orc.snp_records(snps)
    sex, ctrl, off, ids = orc.sample_arrays(samples)
    text, _, _ = orc.generate(rec, sex, ctrl, off, ids, rng_mode=orc.RNG_PHILOX, seed=seed)
    return text, samples


@pytest.mark.gpu
def test_cli_end_to_end_matches_oracle(oracle, tmp_path):
    out = tmp_path / "run1"
    r = _run_cli(["-s", "70", "-c", "90", "-x", "1500", "-f", "0.01", "-n", "1", "-z", "4", "--seed", "4242",
                  "--outdir", str(out), "--tile-rows", "400"])
    assert r.returncode == 0, r.stderr
    for name in ("population.vcf.gz", "population.fam", "pop_deleterious.txt", "deleterious.json", "snps.json.gz"):
        assert os.path.getsize(out / name) > 0
    vcf = gzip.open(out / "population.vcf.gz", "rb").read()
    head, body = vcf.split(b"#CHROM", 1)
    col_line, body = body.split(b"\n", 1)
    assert head.startswith(b"##fileformat=VCFv4.3\n##filedate=") and b"##source=PopFactory\n" in head
    want, samples = _expected_body(str(out), 4242, oracle)
    assert body == want
    iids = [l.split("\t")[1] for l in open(out / "population.fam")]
    assert col_line == ("\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(iids)).encode()
    assert len(samples) == 160 and body.count(b"\n") == 1500
    # BGZF structure: every member carries the BC field, the file ends with the 28-byte EOF block
    from dna_factory_amd import bgzf

    sizes = bgzf.block_sizes(str(out / "population.vcf.gz"))
    assert sizes[-1] == 28 and max(sizes) <= 65536

    # --host-bgzf (zlib at -z, like the reference) produces the same decompressed file
    out3 = tmp_path / "run3"
    r = _run_cli(["-s", "70", "-c", "90", "-x", "1500", "-f", "0.01", "-z", "2", "--seed", "4242", "--host-bgzf",
                  "--outdir", str(out3)])
    assert r.returncode == 0, r.stderr
    vcf3 = gzip.open(out3 / "population.vcf.gz", "rb").read()
    assert vcf3.split(b"#CHROM", 1)[1] == vcf.split(b"#CHROM", 1)[1]

    # same seed, SNP list and deleterious set on two shards (two contexts) -> identical decompressed VCF body
    out2 = tmp_path / "run2"
    r = _run_cli(["-s", "70", "-c", "90", "--seed", "4242", "--snps_file", str(out / "snps.json.gz"),
                  "--deleterious_file", str(out / "deleterious.json"), "--devices", "0,0", "--outdir", str(out2)])
    assert r.returncode == 0, r.stderr
    # (sample sexes / carried sets are re-drawn from the same seeds in the same order only if the feeder call
    # sequence is identical; here the SNP generation is skipped, so compare against this run's own side files)
    vcf2 = gzip.open(out2 / "population.vcf.gz", "rb").read()
    body2 = vcf2.split(b"#CHROM", 1)[1].split(b"\n", 1)[1]
    import shutil

    shutil.copy(out / "snps.json.gz", out2 / "snps.json.gz")
    want2, _ = _expected_body(str(out2), 4242, oracle)
    assert body2 == want2
    assert bgzf.block_sizes(str(out2 / "population.vcf.gz"))[-1] == 28


def test_reference_chunk_ranges(cli):
    # split_list semantics, common/snp.py:71-79 with chunks = int(n / 1e6) or 1 (pop_factory.py:407-410)
    assert cli.reference_chunks(10) == [(0, 10)]
    assert cli.reference_chunks(1999999) == [(0, 1999999)]
    assert cli.reference_chunks(2000001) == [(0, 1000000), (1000000, 2000001)]
    assert cli.reference_chunks(3500000) == [(0, 1166667), (1166667, 2333334), (2333334, 3500000)]


@pytest.mark.gpu
def test_cli_mt19937_continues_the_numpy_stream(oracle, tmp_path):
    """--rng mt19937: the VCF body is what the reference's worker draws next from the host side's numpy stream.
    The expected state is obtained by replaying the same feeder calls (they are pinned against the reference in
    tests/test_population_host.py)."""
    import random

    import numpy

    from dna_factory_amd import population as pop

    out = tmp_path / "mt"
    r = _run_cli(["-s", "30", "-c", "22", "-x", "700", "-f", "0.01", "--seed", "86399", "--rng", "mt19937",
                  "--outdir", str(out), "--tile-rows", "90", "--devices", "0,0"])
    assert r.returncode == 0, r.stderr
    numpy.random.seed(86399)
    random.seed(86399)
    tb = pop.sort_snp_table(pop.generate_snp_table(700, min_maf=0.01))
    groups = pop.pick_deleterious(tb, os.path.join(REPO, "deleterious.yml"))
    pop.build_samples(groups, 22, 30, 0.5, str(tmp_path))
    st = numpy.random.get_state()
    snps = []
    for line in gzip.open(out / "snps.json.gz", "rt"):
        o = json.loads(line)
        snps.append((o["id"], o["chromosome"], o["position"], list(o["tuples"].items())))
    _, samples = _expected_body(str(out), 0, oracle)
    rec = oracle.snp_records(snps)
    sex, ctrl, off, ids = oracle.sample_arrays(samples)
    want, _, _ = oracle.generate(rec, sex, ctrl, off, ids, rng_mode=oracle.RNG_MT19937,
                                 mt_state=(st[1].astype(np.uint32), int(st[2])))
    body = gzip.open(out / "population.vcf.gz", "rb").read().split(b"#CHROM", 1)[1].split(b"\n", 1)[1]
    assert body == want


@pytest.mark.gpu
def test_cli_refuses_the_database_path(tmp_path):
    r = _run_cli(["-s", "1", "-c", "1", "-l", "--outdir", str(tmp_path / "x")])
    assert r.returncode != 0 and "not supported" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_cli_chunk_restart_matches_the_reference_chunking(oracle, tmp_path):
    """--compat-chunk-restart: from 2M SNPs up the reference forks fresh workers per work chunk, so every chunk draws
    from the start of the stream again.  Checked on 2,000,001 SNPs x 4 samples against per-chunk oracle runs."""
    out = tmp_path / "chunks"
    r = _run_cli(["-s", "2", "-c", "2", "-x", "2000001", "--seed", "77", "--compat-chunk-restart", "--outdir", str(out)])
    assert r.returncode == 0, r.stderr
    snps = []
    for line in gzip.open(out / "snps.json.gz", "rt"):
        o = json.loads(line)
        snps.append((o["id"], o["chromosome"], o["position"], list(o["tuples"].items())))
    _, samples = _expected_body_inputs(str(out))
    rec = oracle.snp_records(snps)
    sex, ctrl, off, ids = oracle.sample_arrays(samples)
    body = gzip.open(out / "population.vcf.gz", "rb").read().split(b"#CHROM", 1)[1].split(b"\n", 1)[1]
    pos = 0
    for lo, hi in ((0, 1000000), (1000000, 2000001)):
        want, _, _ = oracle.generate(rec[lo:hi], sex, ctrl, off, ids, rng_mode=oracle.RNG_PHILOX, seed=77, rng_row0=0,
                                     want_packed=False)
        assert body[pos: pos + len(want)] == want
        pos += len(want)
    assert pos == len(body)


@pytest.mark.gpu
def test_plink_bed_matches_the_vcf(oracle, tmp_path):
    """--plink: population.bed / .bim made on the GPU from the packed matrix agree, genotype by genotype, with the
    VCF body the ORACLE produces for the run's inputs (which the run's own VCF must equal) under PLINK's import rules
    (A1 = ALT, A2 = REF; haploid calls stored as homozygous, '.' as missing).  plink itself is not installed here:
    parity with its own converter is unpinned."""
    import gzip

    import numpy as np

    out = tmp_path / "o"
    rc = subprocess.run([sys.executable, os.path.join(REPO, "pop_factory.py"), "-s", "37", "-c", "30", "-x", "3000", "-f", "0.01", "--seed", "5",
                         "--plink", "--outdir", str(out)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr[-2000:]
    S = 67
    raw = (out / "population.bed").read_bytes()
    assert raw[:3] == bytes([0x6C, 0x1B, 0x01])
    bed = np.frombuffer(raw[3:], dtype=np.uint8).reshape(3000, (S + 3) // 4)
    bim = [l.split("\t") for l in (out / "population.bim").read_text().splitlines()]
    want_body, _ = _expected_body(str(out), 5, oracle)
    lines = [l for l in gzip.open(out / "population.vcf.gz", "rt").read().splitlines() if not l.startswith("#")]
    assert ("\n".join(lines) + "\n").encode() == want_body  # the expectations below are the oracle's cells
    assert len(lines) == len(bim) == 3000
    fam = (out / "population.fam").read_text().splitlines()
    assert len(fam) == S
    seen = set()
    for j, line in enumerate(lines):
        f = line.split("\t")
        assert [f[0], f[2], f[1], f[4].split(",")[0], f[3]] == [bim[j][0], bim[j][1], bim[j][3], bim[j][4], bim[j][5].strip()] and bim[j][2] == "0"
        codes = (bed[j][:, None] >> (2 * np.arange(4))[None, :]).reshape(-1) & 3
        assert not codes[S:].any()  # padding bits of the last byte are zero
        for i, cell in enumerate(f[9:]):
            if cell == ".":
                want = 1
            elif "/" in cell:
                a, b = cell.split("/")
                want = {0: 3, 1: 2, 2: 0}[(a != "0") + (b != "0")]
            else:
                want = 3 if cell == "0" else 0
            assert codes[i] == want, (j, i, cell)
            seen.add((f[0] in ("X", "Y", "MT"), cell))
    assert {c for _, c in seen} >= {"0/0", "0/1", "1/1", "0", "1", "."}


@pytest.mark.gpu
def test_two_shards_give_the_same_vcf_and_bed(tmp_path):
    """--devices 0,0: two contexts write their SNP ranges to shard files that are appended in the kernel
    (copy_file_range); VCF text and .bed bytes equal the one-device run's."""
    import gzip

    base = [sys.executable, os.path.join(REPO, "pop_factory.py"), "-s", "90", "-c", "80", "-x", "5000", "-f", "0.01", "--seed", "9", "--plink"]
    one, two = tmp_path / "one", tmp_path / "two"
    for out, extra in ((one, []), (two, ["--devices", "0,0", "--tile-rows", "700"])):
        rc = subprocess.run(base + ["--outdir", str(out)] + extra, capture_output=True, text=True)
        assert rc.returncode == 0, rc.stderr[-2000:]
    body = lambda p: [l for l in gzip.open(p / "population.vcf.gz", "rt").read().splitlines() if not l.startswith("##filedate")]
    assert body(one) == body(two)
    assert (one / "population.bed").read_bytes() == (two / "population.bed").read_bytes()
    assert (one / "population.bim").read_bytes() == (two / "population.bim").read_bytes()
    assert not [f for f in os.listdir(two) if "shard" in f]


# ---- BASELINE configs 1 and 2 at their stated sizes, against the unmodified reference CLI (tests/golden/config*_sha256.json) ----
CONFIGS = {"config1": (100, 100, 100000), "config2": (1000, 1000, 1000000),
           "config3_1pct": (10000, 10000, 50000),  # config 3's 20,000 samples and flow at 1 % of its SNPs (all of it: 40 h of the reference)
           "config5_1pct": (25000, 25000, 10000),  # config 5's 50,000 samples and four deleterious groups (the fixture holds the YAML)
           # 2,000,001 SNPs: the reference cuts them into two work chunks and its forked workers restart the random stream at
           # each (pop_factory.py:402-411) -- what it writes for BASELINE configs 3 and 4 (5M / 40M SNPs)
           "chunks_2000001": (2, 2, 2000001)}
# -n 2 / -n 3: the reference's forked workers draw one and the same stream, reproduced by --compat-workers
WORKER_CONFIGS = {"workers2_100x20000": (50, 50, 20000, 2), "workers3_100x20001": (50, 50, 20001, 3)}
EXTRA_FLAGS = {"chunks_2000001": ["--compat-chunk-restart"]}


def _config_golden(name):
    path = os.path.join(REPO, "tests", "golden", "%s_sha256.json" % name)
    if not os.path.exists(path):
        pytest.skip("%s: no hash fixture (oracle/make_golden.py --only-%s)" % (name, name.replace("_", "-")))
    with open(path) as f:
        return json.load(f)


def _sha(data):
    import hashlib

    return {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_config_host_files_equal_the_reference(cli, tmp_path, config):
    """`-s 100 -c 100 -x 100000 -f 0.01` (config 1; config 2: `-s 1000 -c 1000 -x 1000000`) at the reference's seeds (numpy 123456 from its frozen clock, python random the
    same): snps.json.gz, deleterious.json, population.fam and pop_deleterious.txt of this build hash to what the
    unmodified reference wrote (oracle/make_golden.py case_config1).  Host side only: runs without a GPU."""
    import random

    from dna_factory_amd import population as pop

    gold = _config_golden(config)
    n_ctrl, n_cases, n_snps = CONFIGS[config]
    seed = gold["numpy_seed"]
    np.random.seed(seed)
    random.seed(gold["python_random_seed"])
    out = str(tmp_path) + os.path.sep
    tb = pop.sort_snp_table(pop.generate_snp_table(n_snps, min_maf=0.01))
    pop.write_snps_file(tb, out + "snps.json.gz")
    yml = os.path.join(REPO, "deleterious.yml")
    if "deleterious_yml" in gold:
        yml = out + "groups.yml"
        with open(yml, "w") as f:
            f.write(gold["deleterious_yml"])
    groups = pop.pick_deleterious(tb, yml)
    pop.write_deleterious_json(groups, out + "deleterious.json")
    pop.build_samples(groups, n_ctrl, n_cases, 0.5, out, offset=0)
    assert _sha(gzip.open(out + "snps.json.gz", "rb").read()) == gold["snps_json_decompressed"]
    assert _sha(open(out + "deleterious.json", "rb").read()) == gold["deleterious_json"]
    assert _sha(open(out + "population.fam", "rb").read()) == gold["population_fam"]
    assert _sha(open(out + "pop_deleterious.txt", "rb").read()) == gold["pop_deleterious_txt"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", sorted(CONFIGS) + ["config3_1pct@4shards"])
def test_config_whole_output_directory_equals_the_reference(cli, tmp_path, config):
    """BASELINE config 1 end to end: `pop_factory.py -s 100 -c 100 -x 100000 -f 0.01 -n 1 --rng mt19937 --seed 123456`
    writes the VCF (200 samples x 100,000 SNPs, every byte but the ##filedate line) and the four side files of the
    unmodified reference run at the same seeds.  Config 2 likewise (2,000 samples x 1,000,000 SNPs: 8 GB of VCF text)."""
    extra = []
    if config.endswith("@4shards"):  # BASELINE config 4's mechanism: SNP-range shards (four contexts on GPU 0), files concatenated
        config = config.split("@")[0]
        extra = ["--devices", "0,0,0,0"]
    gold = _config_golden(config)
    n_ctrl, n_cases, n_snps = CONFIGS[config]
    out = str(tmp_path / config)
    extra += EXTRA_FLAGS.get(config, [])
    if "deleterious_yml" in gold:
        (tmp_path / "groups.yml").write_text(gold["deleterious_yml"])
        extra += ["-p", str(tmp_path / "groups.yml")]
    p = _run_cli(["-s", str(n_ctrl), "-c", str(n_cases), "-x", str(n_snps), "-f", "0.01", "-n", "1", "--rng", "mt19937", "--seed",
                  str(gold["numpy_seed"]), "--outdir", out] + extra)
    assert p.returncode == 0, p.stderr[-3000:]
    import hashlib

    _assert_directory_equals(out, gold)


def _vcf_digest(out):
    """What oracle/make_golden.py's stand-in for biopython's writer records: SHA-256 of the decompressed VCF without its
    ##filedate line, byte and line counts, head of the first body line."""
    import hashlib

    h, n_bytes, n_lines, first = hashlib.sha256(), 0, 0, None
    with gzip.open(os.path.join(out, "population.vcf.gz"), "rb") as f:
        for line in f:
            if line.startswith(b"##filedate="):
                continue
            if first is None and not line.startswith(b"#"):
                first = line[:120].decode()
            h.update(line)
            n_bytes += len(line)
            n_lines += 1
    return {"sha256": h.hexdigest(), "bytes": n_bytes, "lines": n_lines, "first_body_line_head": first}


def _assert_directory_equals(out, gold):
    """Every file the reference run wrote (the fixture lists them) hashes the same here; files the reference did not
    write in that mode (snps.json.gz / deleterious.json when they were passed in) must be absent here too."""
    got, want = _vcf_digest(out), gold["vcf_without_filedate"]
    assert got["first_body_line_head"] == want["first_body_line_head"]
    assert (got["lines"], got["bytes"]) == (want["lines"], want["bytes"])
    assert got["sha256"] == want["sha256"]
    if "snps_json_decompressed" in gold:
        assert _sha(gzip.open(os.path.join(out, "snps.json.gz"), "rb").read()) == gold["snps_json_decompressed"]
    else:
        assert not os.path.exists(os.path.join(out, "snps.json.gz"))
    for name, key in (("deleterious.json", "deleterious_json"), ("population.fam", "population_fam"),
                      ("pop_deleterious.txt", "pop_deleterious_txt")):
        if key in gold:
            assert _sha(open(os.path.join(out, name), "rb").read()) == gold[key], name
        else:
            assert not os.path.exists(os.path.join(out, name)), name


@pytest.mark.gpu
@pytest.mark.parametrize("config", sorted(WORKER_CONFIGS))
def test_default_worker_count_of_the_reference_is_reproduced(cli, tmp_path, config):
    """`pop_factory.py -s 50 -c 50 -x 20000 -f 0.01 -n 2` -- the reference's DEFAULT worker count -- through the unmodified
    reference (its two forked workers draw the same stream: rows 2k+1 and 2k+2 come from the same uniforms) against
    `--rng mt19937 --compat-workers -n 2` here: the whole output directory, hash for hash.  Likewise -n 3 with a row
    count that is not a multiple of three."""
    gold = _config_golden(config)
    n_ctrl, n_cases, n_snps, n_workers = WORKER_CONFIGS[config]
    out = str(tmp_path / config)
    p = _run_cli(["-s", str(n_cases), "-c", str(n_ctrl), "-x", str(n_snps), "-f", "0.01", "-n", str(n_workers), "--compat-workers",
                  "--rng", "mt19937", "--seed", str(gold["numpy_seed"]), "--outdir", out, "--tile-rows", "7000"])
    assert p.returncode == 0, p.stderr[-3000:]
    _assert_directory_equals(out, gold)


@pytest.mark.gpu
def test_compat_workers_with_philox_matches_the_oracle_stripe_by_stripe(oracle, tmp_path):
    """The same row -> stream-row map with the Philox stream (every kernel form takes rng_row = row // N): stripe w of the
    output -- rows w, w + N, ... -- equals the oracle's lines for those rows drawn from stream rows 0, 1, ..."""
    for S_half, n_snps, N in ((30, 1501, 3), (1100, 260, 2)):
        out = tmp_path / ("w%d" % N)
        r = _run_cli(["-s", str(S_half), "-c", str(S_half), "-x", str(n_snps), "-f", "0.01", "-n", str(N), "--compat-workers", "--seed", "99",
                      "--outdir", str(out)])
        assert r.returncode == 0, r.stderr[-3000:]
        snps = []
        for line in gzip.open(out / "snps.json.gz", "rt"):
            o = json.loads(line)
            snps.append((o["id"], o["chromosome"], o["position"], list(o["tuples"].items())))
        _, samples = _expected_body_inputs(str(out))
        rec = oracle.snp_records(snps)
        sex, ctrl, off, ids = oracle.sample_arrays(samples)
        body = gzip.open(out / "population.vcf.gz", "rb").read().split(b"#CHROM", 1)[1].split(b"\n", 1)[1]
        lines = body.split(b"\n")[:-1]
        assert len(lines) == n_snps
        for w in range(N):
            want, _, _ = oracle.generate(rec[w::N], sex, ctrl, off, ids, rng_mode=oracle.RNG_PHILOX, seed=99, rng_row0=0, want_packed=False)
            assert b"".join(l + b"\n" for l in lines[w::N]) == want, (N, w)


# ---- the reference README's multi-job workflow (README.md:69-103), three runs, against the unmodified reference ----
def _workflow_commands(base, offset_step3=1000):
    d1, d2, d3 = (os.path.join(base, d) for d in ("step1", "step2", "step3"))
    reuse = ["--snps_file", os.path.join(d1, "snps.json.gz"), "--deleterious_file", os.path.join(d1, "deleterious.json")]
    return [("step1", d1, ["-s", "0", "-c", "0", "-x", "20000", "-f", "0.01"]),
            ("step2", d2, ["-s", "50", "-c", "50"] + reuse),
            ("step3", d3, ["-s", "50", "-c", "50", "--offset", str(offset_step3)] + reuse)]


@pytest.mark.gpu
def test_readme_multi_job_workflow_equals_the_reference(cli, tmp_path):
    """Step 1 `-s 0 -c 0 -x 20000 -f 0.01` (selects SNPs and deleterious SNPs, a VCF of lines without cells), step 2
    `-s 50 -c 50 --snps_file <1>/snps.json.gz --deleterious_file <1>/deleterious.json`, step 3 the same with `--offset
    1000`: with `--rng mt19937 --compat-deleterious-file-bug` every output directory is what the unmodified reference
    wrote for the same commands (tests/golden/workflow_sha256.json; S = 0 through the CLI and the GPU encoder,
    load_snps_file / load_deleterious pop_factory.py:264-272, :515-519, :613-619, ids with an offset :341-383).
    Without the compat flag the ids of a loaded deleterious.json are integers, so forcing works: the
    sample files stay the reference's, the VCF differs in the carriers' cells."""
    gold = _config_golden("workflow")
    common = ["-n", "1", "--rng", "mt19937", "--seed", str(gold["step1"]["numpy_seed"])]
    for step, out, args in _workflow_commands(str(tmp_path)):
        p = _run_cli(args + common + ["--compat-deleterious-file-bug", "--outdir", out])
        assert p.returncode == 0, p.stderr[-3000:]
        _assert_directory_equals(out, gold[step])
    # default: integer keys
    step, _out, args = _workflow_commands(str(tmp_path))[1]
    fixed = str(tmp_path / "step2_forcing")
    p = _run_cli(args + common + ["--outdir", fixed])
    assert p.returncode == 0, p.stderr[-3000:]
    for name, key in (("population.fam", "population_fam"), ("pop_deleterious.txt", "pop_deleterious_txt")):
        assert _sha(open(os.path.join(fixed, name), "rb").read()) == gold[step][key], name
    got, want = _vcf_digest(fixed), gold[step]["vcf_without_filedate"]
    assert (got["lines"], got["bytes"]) == (want["lines"], want["bytes"]) and got["sha256"] != want["sha256"]
    a = gzip.open(os.path.join(fixed, "population.vcf.gz"), "rt").read().splitlines()
    b = gzip.open(os.path.join(str(tmp_path), "step2", "population.vcf.gz"), "rt").read().splitlines()
    carried = {}
    for line in open(os.path.join(fixed, "pop_deleterious.txt")):
        parts = line.rstrip("\n").split("\t")
        carried[parts[0]] = {x for x in parts[2:] if x}
    iids = a[5].split("\t")[9:]
    n_forced = 0
    for la, lb in zip(a[6:], b[6:]):
        if la == lb:
            continue
        fa, fb = la.split("\t"), lb.split("\t")
        assert fa[:9] == fb[:9]
        for iid, ca, cb in zip(iids, fa[9:], fb[9:]):
            if ca != cb:  # only carriers' cells change, and only to the forced genotype (pop_factory.py:495-499)
                assert fa[2] in carried[iid] and ca in ("1/1", "1"), (fa[2], iid, ca, cb)
                n_forced += 1
    assert n_forced > 0


@pytest.mark.gpu
def test_readme_workflow_at_the_default_worker_count(cli, tmp_path):
    """The README's three commands as written there -- no -n, i.e. the reference's default of two workers -- against
    `--compat-workers` (tests/golden/workflow_n2_sha256.json)."""
    gold = _config_golden("workflow_n2")
    common = ["--compat-workers", "--rng", "mt19937", "--seed", str(gold["step1"]["numpy_seed"]), "--compat-deleterious-file-bug"]
    for step, out, args in _workflow_commands(str(tmp_path)):
        p = _run_cli(args + common + ["--outdir", out])
        assert p.returncode == 0, p.stderr[-3000:]
        _assert_directory_equals(out, gold[step])


def test_workflow_sample_files_equal_the_reference(cli, tmp_path):
    """Host side of the same three runs, without a GPU: snps.json.gz / deleterious.json written by step 1, read back by
    steps 2 and 3 (string keys like the reference's from_json, or integers), population.fam and pop_deleterious.txt with
    and without --offset."""
    import random

    from dna_factory_amd import population as pop

    gold = _config_golden("workflow")
    seed = gold["step1"]["numpy_seed"]
    d1 = str(tmp_path / "s1") + os.path.sep
    os.makedirs(d1)
    np.random.seed(seed)
    random.seed(gold["step1"]["python_random_seed"])
    tb = pop.sort_snp_table(pop.generate_snp_table(20000, min_maf=0.01))
    pop.write_snps_file(tb, d1 + "snps.json.gz")
    groups = pop.pick_deleterious(tb, os.path.join(REPO, "deleterious.yml"))
    pop.write_deleterious_json(groups, d1 + "deleterious.json")
    st = pop.build_samples(groups, 0, 0, 0.5, d1, offset=0)
    assert len(st) == 0
    for name, key in (("deleterious.json", "deleterious_json"), ("population.fam", "population_fam"),
                      ("pop_deleterious.txt", "pop_deleterious_txt")):
        assert _sha(open(d1 + name, "rb").read()) == gold["step1"][key], name
    assert _sha(gzip.open(d1 + "snps.json.gz", "rb").read()) == gold["step1"]["snps_json_decompressed"]
    for step, offset in (("step2", 0), ("step3", 1000)):
        for int_keys in (False, True):
            d = str(tmp_path / (step + str(int_keys))) + os.path.sep
            os.makedirs(d)
            np.random.seed(seed)
            random.seed(gold[step]["python_random_seed"])
            tb2 = pop.sort_snp_table(pop.load_snps_file(d1 + "snps.json.gz"))
            assert len(tb2) == 20000
            g2 = pop.load_deleterious_json(d1 + "deleterious.json", int_keys=int_keys)
            pop.build_samples(g2, 50, 50, 0.5, d, offset=offset)
            for name, key in (("population.fam", "population_fam"), ("pop_deleterious.txt", "pop_deleterious_txt")):
                assert _sha(open(d + name, "rb").read()) == gold[step][key], (step, int_keys, name)


def test_cli_rejects_an_empty_device_list(tmp_path):
    r = _run_cli(["-s", "1", "-c", "1", "-x", "10", "--devices", "", "--outdir", str(tmp_path / "d")])
    assert r.returncode != 0 and "--devices" in (r.stderr + r.stdout)
    r = _run_cli(["-s", "1", "-c", "1", "-x", "10", "--devices", "0,x", "--outdir", str(tmp_path / "e")])
    assert r.returncode != 0 and "--devices" in (r.stderr + r.stdout)
