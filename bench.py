#!/usr/bin/env python3
"""Benchmark of the row-generation hot path (BASELINE.json: SNP-rows/s at fixed #samples).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one tile of `--tile-rows` SNP rows for all `--samples` samples
(text + packed 2-bit matrix, both left resident in HBM).  The default workload is BASELINE config 3
("10k cases + 10k controls x 5M SNPs, -f 0.01"): 20 steps x 250,000 rows = 5M rows per GPU.  With N GPUs every
rank generates its own 5M-row SNP range of one global table (SNP-range sharding, no collective on the data
path: weak scaling); `value` = rows all ranks generated / max-over-ranks wall time.

The JSON line also carries `roofline` (algorithmic bytes per launch / mean kernel time from HIP events on the
launch stream, vs the 8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6, help="untimed steps first (the clocks take four or five launches to settle)")
    ap.add_argument("--samples", type=int, default=20000, help="cases + controls (half each)")
    ap.add_argument("--tile-rows", type=int, default=250000, help="SNP rows per step")
    ap.add_argument("--table-rows", type=int, default=5000000, help="SNP rows per GPU (tiles cycle through them)")
    ap.add_argument("--min-maf", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=12345, help="row RNG seed")
    ap.add_argument("--table-seed", type=int, default=20261004, help="seed of the synthetic SNP / sample tables")
    ap.add_argument("--what", choices=["text+packed", "text", "packed"], default="text+packed")
    ap.add_argument("--rng", choices=["philox", "mt19937"], default="philox",
                    help="row RNG; mt19937 = numpy-compatible stream (tile rows are rounded to whole rounds of 4096 waves of its row quantum)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--bgzf-tiles", type=int, default=2,
                    help="after the timed region, compress this many tiles with the on-GPU BGZF encoder and report "
                         "its rate and ratio (0 = skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="HBM bytes per launch from a rocprofv3 --pmc run of this same command (optional)")
    return ap.parse_args()


def synthetic_samples(S, n_rows_table, snp_ids, rs):
    """Controls first, then cases (pop_factory.py:357-358); sex ~ Bernoulli(0.5) (:365).  Forced set shaped like
    the default deleterious.yml: 66 SNPs over 14 group instances; each case belongs to one instance and carries a
    random subset of its SNPs (select_mutations, pop_factory.py:621-635)."""
    from dna_factory_amd import capi
    from dna_factory_amd.tables import SampleTable

    n_ctrl = S // 2
    flags = (rs.random_sample(S) <= 0.5).astype(np.uint8) * capi.SAMPLE_MALE
    flags[:n_ctrl] |= capi.SAMPLE_CONTROL
    group_sizes = [3, 3, 3, 3, 10, 10, 10, 4, 4, 4, 4, 4, 1, 3]  # deleterious.yml:8-68
    n_slots = sum(group_sizes)
    slot_ids = rs.choice(snp_ids, size=n_slots, replace=False)
    bounds = np.concatenate([[0], np.cumsum(group_sizes)])
    case_group = rs.randint(0, len(group_sizes), size=S - n_ctrl)
    carriers = [[] for _ in range(n_slots)]
    for k, g in enumerate(case_group):
        lo, hi = bounds[g], bounds[g + 1]
        take = rs.randint(1, hi - lo + 1)
        for s in rs.choice(np.arange(lo, hi), size=take, replace=False):
            carriers[s].append(n_ctrl + k)
    order = np.argsort(slot_ids)
    off = np.zeros(n_slots + 1, dtype=np.uint64)
    idx = []
    for n, s in enumerate(order):
        idx.extend(sorted(carriers[s]))
        off[n + 1] = len(idx)
    return SampleTable(flags, slot_ids[order], off, np.asarray(idx, dtype=np.uint32))


def kernel_source_hash():
    """Content hash of everything the library is built from (build.source_hash(): all of csrc/ and include/dnaf.h) -- the
    launch shape lives in dnaf_api.hip and the short-row shapes run rowmix_kernels.hip.h, so a PMC traffic figure has to
    have been measured on this very build."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_dnaf_build", os.path.join(ROOT, "dna-factory_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()[:16]


def load_traffic(a, S):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py) -- only when they were taken
    on this very kernel source and workload; otherwise null, with the reason."""
    if not os.path.exists(a.traffic_json):
        return None, "no PMC file (%s)" % os.path.relpath(a.traffic_json, ROOT)
    try:
        tj = json.load(open(a.traffic_json))
    except (OSError, ValueError) as e:
        return None, "unreadable PMC file: %s" % e
    if not (tj.get("samples") == S and tj.get("tile_rows") == a.tile_rows and tj.get("what") == a.what):
        return None, "PMC file is for another workload"
    if tj.get("kernel_source_sha16") != kernel_source_hash():
        return None, "PMC file was measured on other kernel sources (%s, now %s)" % (tj.get("kernel_source_sha16"), kernel_source_hash())
    return tj.get("hbm_bytes_per_launch"), "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command, %s" % os.path.relpath(a.traffic_json, ROOT)


def cpu_baseline(tb, st, seed, target_seconds):
    """Time the CPU oracle (oracle/rowgen_oracle.c, the checker -- never the product) on cores-1 host threads over
    a bounded prefix of the same workload."""
    from oracle import oracle as orc

    orc.build()
    S = len(st)
    try:
        avail = len(os.sched_getaffinity(0))  # the cores this process may actually run on
    except (AttributeError, OSError):
        avail = os.cpu_count() or 2
    try:  # a container may see every core but be limited to a CPU quota (cgroup v2 cpu.max = "quota period")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, avail - 1)  # the reference's advice: never more workers than cores - 1 (README.md:59-68)
    # oracle inputs from the SoA tables
    def records(lo, hi):
        rec = np.zeros(hi - lo, dtype=orc.SNP_DTYPE)
        rec["chrom"] = tb.chrom[lo:hi]
        rec["position"] = tb.position[lo:hi]
        rec["id"] = tb.snp_id[lo:hi]
        rec["ref"] = tb.ref[lo:hi]
        rec["alt"] = tb.alt[lo:hi]
        rec["n_alleles"] = tb.n_alleles[lo:hi]
        rec["cum"][:, :4] = tb.cum[lo:hi]
        return rec

    from dna_factory_amd import capi

    sex = np.where(st.flags & capi.SAMPLE_MALE, 1, 2).astype(np.uint8)
    ctrl = ((st.flags & capi.SAMPLE_CONTROL) != 0).astype(np.uint8)
    per_sample = [[] for _ in range(S)]
    for k, sid in enumerate(st.forced_ids):
        for i in st.carrier_idx[int(st.carrier_off[k]): int(st.carrier_off[k + 1])]:
            per_sample[int(i)].append(int(sid))
    doff = np.zeros(S + 1, dtype=np.uint64)
    dids = []
    for i in range(S):
        dids.extend(per_sample[i])
        doff[i + 1] = len(dids)
    dids = np.asarray(dids, dtype=np.int64)
    # calibrate on a small sample, then size the run for ~target_seconds
    n0, dt0 = 8 * cores, 0.0
    while True:
        n0 = min(len(tb), n0)
        t0 = time.perf_counter()
        orc.generate_threads(records(0, n0), sex, ctrl, doff, dids, seed, cores)
        dt0 = max(time.perf_counter() - t0, 1e-6)
        if dt0 >= 0.5 or n0 >= len(tb):
            break
        n0 *= 4
    n = int(min(len(tb), max(n0, n0 * target_seconds / dt0)))
    rec = records(0, n)
    t0 = time.perf_counter()
    orc.generate_threads(rec, sex, ctrl, doff, dids, seed, cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "SNP-rows/s", "cores": cores, "kind": "port",
            "genotypes_per_s": n * S / dt,
            "sample": "first %d rows x %d samples of the same table, CPU oracle (C restatement of "
                      "pop_factory.py:471-513, Philox draws) on %d threads, %.1f s" % (n, S, cores, dt)}


class _Rank:
    """One GPU's share of the benchmark: its own SNP range of the global table, its own context."""

    def __init__(self, a, rank, device):
        from dna_factory_amd import capi
        from dna_factory_amd.genome import synthetic_snp_table
        from dna_factory_amd.rowgen import RowGenerator

        self.a, self.rank = a, rank
        S = a.samples
        rows = max(a.table_rows, a.tile_rows)
        rs = np.random.RandomState(a.table_seed + 1000 * rank)
        tb = synthetic_snp_table(rows, a.min_maf, rs)
        self.st = synthetic_samples(S, rows, tb.snp_id, np.random.RandomState(a.table_seed))  # same samples on every rank
        self.gen = gen = RowGenerator(device=device, seed=a.seed, rng=a.rng)
        gen.set_samples(self.st)
        q = gen.row_quantum
        if a.rng == "mt19937" and q > 1:
            # a wave owns q rows; 8192 waves are resident at once (5 KB of LDS, 8 waves per SIMD): tiles of multiples of 4096 waves
            q *= 4096 if a.tile_rows >= 4096 * q else 1
            self.tile_rows = max(q, (a.tile_rows + q // 2) // q * q)
        else:
            self.tile_rows = max(q, a.tile_rows // q * q)
        rows = max(self.tile_rows, rows // self.tile_rows * self.tile_rows)
        if rows < len(tb):
            tb = tb.slice(0, rows)
        self.tb, self.rows = tb, rows
        gen.set_snps(tb, rng_row0=rank * rows)  # this rank's SNP range of the global table
        self.what = {"text+packed": capi.OUT_TEXT | capi.OUT_PACKED, "text": capi.OUT_TEXT, "packed": capi.OUT_PACKED}[a.what]
        self.n_tiles = max(1, rows // self.tile_rows)
        # size the output buffers for the largest tile up front: no allocation inside the timed region
        gen.ctx.reserve_output(max(gen.ctx.text_bytes(t * self.tile_rows, self.tile_rows) for t in range(self.n_tiles)),
                               self.tile_rows * gen.words_per_row)

    def step(self, i):
        self.gen.ctx.generate((i % self.n_tiles) * self.tile_rows, self.tile_rows, self.what)

    def run(self, barrier):
        """W untimed steps, then exactly K steps between two barriers; -> (wall seconds, kernel ms, launches recorded)."""
        a = self.a
        for i in range(a.warmup):
            self.step(i)
        self.gen.ctx.sync()
        barrier()
        self.gen.ctx.timing_begin()
        t0 = time.perf_counter()
        for i in range(a.steps):
            self.step(a.warmup + i)
        self.gen.ctx.sync()
        t1 = time.perf_counter()
        barrier()
        kern_ms, n_rec, _ = self.gen.ctx.timing_end()
        return t1 - t0, kern_ms, n_rec


def main():
    a = parse_args()
    n = max(a.gpus, 1)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world > 1 and env_world != n:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, env_world))
    share_gpu = bool(os.environ.get("DNAF_BENCH_SHARE_GPU"))  # rehearsal on a one-GPU box: every rank on GPU 0

    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ROOT, "dna-factory_amd", "csrc", "libdnaf_hip.so")):
        ge.build()
    from dna_factory_amd import capi

    # never report n_gpus GPUs without having used that many
    have = capi.device_count()
    if not share_gpu and have < n:
        raise SystemExit("--gpus %d but this node shows %d GPU(s)" % (n, have))

    per_rank_s = None
    if env_world > 1:
        # one process per GPU under a launcher (torchrun): the shards never exchange data, so the only collectives are
        # the benchmark's own barrier and max -- on the CPU (gloo), no RCCL
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo")
        r = _Rank(a, rank, 0 if share_gpu else local_rank)
        wall, kern_ms, n_rec = r.run(dist.barrier)
        t = torch.tensor([wall], dtype=torch.float64)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(env_world)]
        dist.all_gather(gathered, t)
        per_rank_s = [float(x.item()) for x in gathered]
        world, launcher = env_world, "one process per GPU (torchrun), gloo barrier"
    elif n > 1:
        # stand-alone: one context per GPU, one host thread each (the product's own model, cli.py), a thread barrier
        import threading

        bar = threading.Barrier(n)
        ranks, results, errors = [None] * n, [None] * n, []

        def work(g):
            try:
                ranks[g] = _Rank(a, g, 0 if share_gpu else g)
                results[g] = ranks[g].run(bar.wait)
            except BaseException as e:  # noqa: BLE001 -- a dead rank must not leave the others at the barrier
                errors.append((g, e))
                bar.abort()

        threads = [threading.Thread(target=work, args=(g,)) for g in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise SystemExit("rank %d failed: %r" % errors[0])
        per_rank_s = [res[0] for res in results]
        r, (wall, kern_ms, n_rec) = ranks[0], results[0]
        world, launcher = n, "one process, one host thread and one context per GPU"
    else:
        r = _Rank(a, 0, 0)
        wall, kern_ms, n_rec = r.run(lambda: None)
        per_rank_s = [wall]
        world, launcher = 1, "single GPU"
    elapsed = max(per_rank_s)

    gen, tb, st, S, what = r.gen, r.tb, r.st, a.samples, r.what
    a.tile_rows, rows, n_tiles = r.tile_rows, r.rows, r.n_tiles
    if rank == 0:
        # algorithmic bytes of one launch: packed row + text row + 32-byte SNP record, per row
        tiles_used = [(a.warmup + i) % n_tiles for i in range(a.steps)]
        text_b = np.mean([gen.ctx.text_bytes(t * a.tile_rows, a.tile_rows) for t in tiles_used])
        bytes_per_launch = 32.0 * a.tile_rows
        if what & capi.OUT_TEXT:
            bytes_per_launch += text_b
        if what & capi.OUT_PACKED:
            bytes_per_launch += a.tile_rows * 4.0 * gen.words_per_row
        mean_kernel_s = (kern_ms / max(n_rec, 1)) * 1e-3
        achieved = bytes_per_launch / mean_kernel_s / 1e9
        traffic, traffic_source = load_traffic(a, S)
        total_rows = a.steps * a.tile_rows * world
        out = {
            "metric": "SNP-rows/sec at fixed #samples",
            "value": total_rows / elapsed,
            "unit": "SNP-rows/s",
            "genotypes_per_s": total_rows * S / elapsed,
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 10k cases + 10k controls x 5M SNPs, -f 0.01 (per GPU), "
                                   "streamed in row tiles; outputs resident in HBM",
                       "samples": S, "tile_rows": a.tile_rows, "table_rows_per_gpu": rows, "min_maf": a.min_maf,
                       "outputs": a.what, "rng": "philox4x32-10" if a.rng == "philox" else "mt19937 (numpy legacy)", "seed": a.seed, "table_seed": a.table_seed,
                       "sharding": "SNP-range per GPU, no collective", "launcher": launcher,
                       "per_gpu_wall_s": per_rank_s},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_6290_GBs": achieved / 6290.0, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "bytes_per_launch": bytes_per_launch, "kernel_ms_mean": mean_kernel_s * 1e3,
                         "launches_timed": n_rec, "kernel": gen.ctx.last_kernel() + (" (+ mt_jump_kernel)" if a.rng == "mt19937" else ""),
                         "draws_per_s": 2.0 * S * a.tile_rows / mean_kernel_s},
        }
        if a.bgzf_tiles > 0 and (what & capi.OUT_TEXT):
            # the sink right after the path: BGZF members encoded on the GPU, outside the timed region
            z_ms, z_in, z_out = 0.0, 0, 0
            for t in range(a.bgzf_tiles):
                gen.ctx.generate((t % n_tiles) * a.tile_rows, a.tile_rows, capi.OUT_TEXT | capi.OUT_PACKED)
                z_out += gen.ctx.bgzf_compress(reuse_tables=t > 0)
                z_in += gen.ctx.text_bytes((t % n_tiles) * a.tile_rows, a.tile_rows)
                z_ms += gen.ctx.last_bgzf_ms()
            out["bgzf"] = {"tiles": a.bgzf_tiles, "text_GB_per_s": z_in / (z_ms * 1e-3) / 1e9,
                           "rows_per_s": a.bgzf_tiles * a.tile_rows / (z_ms * 1e-3), "ratio": z_in / z_out,
                           "bytes_per_genotype": z_out / (a.bgzf_tiles * a.tile_rows * S),
                           "kernel_ms_per_tile": z_ms / a.bgzf_tiles,
                           "encoder": "members in biallelic autosome rows from the 2-bit matrix (LZ77 over cells), the rest from the text"}
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(tb, st, a.seed, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    for rk in (ranks if (env_world == 1 and n > 1) else [r]):  # contexts are closed, not left to the interpreter's exit
        if rk is not None:
            rk.gen.close()
    if env_world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
