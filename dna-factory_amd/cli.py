"""Command line of the generator: the flag surface of the reference's pop_factory.py (pop_factory.py:638-693)
and its output directory layout, with the VCF body produced on MI355X GPUs.

Reference flags keep their names, dests and defaults (pinned by the reference's own
test/unit/pop_factory_test.py:80-104).  Additive flags: --seed, --rng, --devices, --tile-rows, --threads,
--host-bgzf, --compat-deleterious-file-bug.
"""
import argparse
import os
import random
import sys
import threading
import time
from datetime import datetime

import numpy

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTPUT_DIR = os.path.join(ROOT_DIR, "populations")


def parse_cmd_args(args):
    p = argparse.ArgumentParser(fromfile_prefix_chars="@", prog="DNA Factory",
                                description="Generates genetic populations using simulated SNP data.")
    p.add_argument("-s", type=int, dest="size", help="size of afflicted/case group", required=True)
    p.add_argument("-c", type=int, dest="control_size", help="size of control group", required=True)
    p.add_argument("-x", type=int, dest="max_snps", help="max number of snps to load/generate")
    p.add_argument("-p", type=str, default="deleterious.yml", dest="deleterious_config",
                   help="location of deleterious config yaml file (default is deleterious.yml)")
    p.add_argument("-f", type=float, default=0.005, dest="min_freq",
                   help="min minor allele frequency for a SNP to be included, default is 0.005")
    p.add_argument("-m", type=float, default=0.5, dest="male_odds",
                   help="odds of a population member being male (default 0.5)")
    p.add_argument("-n", type=int, default=2, dest="num_processes",
                   help="number of worker processes of the reference; accepted for compatibility, the GPU path "
                        "has no worker processes")
    p.add_argument("-z", type=int, dest="compression_level", default=6, choices=range(1, 10),
                   help="gzip compression level (1=least 9=most) default 6")
    p.add_argument("-l", action="store_const", const=False, default=True, dest="generate_snps",
                   help="load from the RefSNP database instead of simulated snps (not supported by this build)")
    p.add_argument("--deleterious_file", type=str,
                   help="<path> to a deleterious.json file that specifies the exact snps to use as deleterious")
    p.add_argument("--snps_file", type=str, help="<path> location of snps.json.gz file to use as selected snps")
    p.add_argument("--outdir", type=str, help="<path> directory to use for output files")
    p.add_argument("--offset", type=int,
                   help="offset to add to all sample ids. Useful for creating VCF files that can be merged")
    # --- extensions ---
    p.add_argument("--seed", type=int, help="RNG seed (default: wall clock HHMMSS, like the reference); also seeds "
                                            "python's `random` so that a run is reproducible")
    p.add_argument("--rng", choices=["philox", "mt19937"], default="philox",
                   help="row RNG: philox = counter-based Philox4x32-10 (default, fastest); mt19937 = numpy's legacy "
                        "stream continued from the host side's numpy state, i.e. the numbers the reference itself "
                        "would draw with the same seeds (DESIGN.md section 2)")
    p.add_argument("--devices", type=str, default="0", help="comma-separated GPU ids; SNP ranges are sharded over them")
    p.add_argument("--tile-rows", type=int, default=0, help="SNP rows generated per kernel launch (0 = auto)")
    p.add_argument("--threads", type=int, default=0, help="BGZF compression threads per shard (0 = auto)")
    p.add_argument("--host-bgzf", action="store_true",
                   help="compress on the host with zlib at level -z (like the reference) instead of the on-GPU BGZF "
                        "encoder, which ignores -z: smaller files (zlib -6: 0.23 B/genotype against 0.31 at 20k samples; "
                        "1.8 x smaller at 200 samples), at zlib's speed")
    p.add_argument("--plink", action="store_true",
                   help="also write population.bed and population.bim (PLINK 1 binary genotypes made on the GPU from the "
                        "2-bit matrix; population.fam is the matching .fam), so PLINK need not convert the VCF")
    p.add_argument("--compat-chunk-restart", action="store_true",
                   help="restart the random stream at every 1M-SNP work chunk like the reference does "
                        "(pop_factory.py:402-411; only matters from 2M SNPs up)")
    p.add_argument("--compat-workers", action="store_true",
                   help="reproduce what the reference's -n N does to the random stream: its N forked workers all draw the "
                        "parent's stream, so rows kN+1 .. kN+N of a work chunk are made from the same uniforms "
                        "(pop_factory.py:426-434); N is taken from -n (default 2, like the reference); implies "
                        "--compat-chunk-restart")
    p.add_argument("--compat-deleterious-file-bug", action="store_true",
                   help="keep the reference's behaviour for --deleterious_file: ids stay strings, so nothing is forced")
    return p.parse_args(args)


def _auto_tile_rows(n_samples, n_rows, rng="philox"):
    """Rows per launch: about 512 MiB of text.  In MT19937 mode a wave owns 156 / gcd(S, 156) whole rows of the
    sequential stream, so a tile must hold a few thousand waves' worth of rows to fill the chip (up to 16 GiB of text)."""
    per_row = 64 + 4 * max(n_samples, 1)
    rows = (512 << 20) // per_row
    if rng == "mt19937" and n_samples:
        from math import gcd

        rows = max(rows, min(4096 * (156 // gcd(n_samples, 156)), (16 << 30) // per_row))
    return max(1, min(max(n_rows, 1), rows))


def split_ranges(n, chunks):
    """Index ranges of split_list(l, chunks) for len(l) == n (common/snp.py:71-79): chunk size round(n / chunks),
    the last chunk takes the remainder."""
    size = round(n / chunks)
    return [(i * size, n if i + 1 == chunks else (i + 1) * size) for i in range(chunks)]


def reference_chunks(n):
    """Row ranges of the reference's work chunks (output_vcf_population, pop_factory.py:402-411: int(n / 1e6) chunks
    of split_list).  Its workers fork from the same parent state for every chunk, so the random stream restarts at
    each chunk start."""
    return split_ranges(n, int(n / 1000000) or 1)


def _generate_shard(device, seed, rng, mt_state, st, tb, row_lo, row_hi, rng_row_lo, tile_rows, sink, progress, host_bgzf,
                    bed_sink=None, rng_divisor=1):
    """Rows [row_lo, row_hi) of the table on one GPU -> the BGZF sink, in order.  By default the text is
    compressed on the GPU and only BGZF members cross PCIe; with host_bgzf the text is copied out and
    compressed by the sink's zlib workers."""
    from . import capi
    from .rowgen import RowGenerator

    with RowGenerator(device=device, seed=seed, rng=rng, mt_state=mt_state) as gen:
        gen.set_samples(st)
        gen.ctx.set_rng_divisor(rng_divisor)
        gen.set_snps(tb.slice(row_lo, row_hi), rng_row0=rng_row_lo)
        n = row_hi - row_lo
        q = gen.row_quantum * rng_divisor  # tiles begin where a wave of the stream and a group of rows sharing a stream row begin
        tile_rows = max(q, tile_rows // q * q)
        # the sink's I/O threads drain tiles k-2 .. k from the library's pinned staging buffers (four, used in turn) while the
        # GPU generates, compresses and copies out tile k+1: a tile's write has to be over before the fourth fetch after its own
        inflight = []
        trace = os.environ.get("DNAF_TRACE_TILES")  # per-tile seconds: GPU work, device -> host copy, wait for the writer
        for b in range(0, n, tile_rows):
            m = min(tile_rows, n - b)
            t0 = time.time()
            gen.ctx.generate(b, m, capi.OUT_TEXT if host_bgzf and bed_sink is None else capi.OUT_TEXT | capi.OUT_PACKED)
            if bed_sink is not None:
                bed_sink.write(gen.ctx.export_bed())  # .bed rows of this tile, straight from the packed matrix
            if not host_bgzf:
                gen.ctx.bgzf_compress(reuse_tables=b > 0)
            else:
                gen.ctx.sync()
            t1 = time.time()
            while len(inflight) >= 3:
                inflight.pop(0).result()
            t2 = time.time()
            if host_bgzf:
                sink.write(gen.ctx.fetch_text())  # (copied into the compressor's blocks before write returns)
            else:
                fut = sink.write_members(gen.ctx.fetch_bgzf())
                if fut is not None:
                    inflight.append(fut)
            if trace:
                print("tile %d: gpu %.3f s, writer wait %.3f s, copy + hand-over %.3f s" % (b // tile_rows, t1 - t0, t2 - t1, time.time() - t2), flush=True)
            progress(m)
        for fut in inflight:
            fut.result()


def _write_round_robin(path, header, st, tb, seed, rng, devices, compresslevel, tile_rows, threads, host_bgzf, mt_state, bed_path, progress):
    """Tile k of the table goes to device k mod G; the calling thread writes the tiles' BGZF members (and .bed rows) in
    order as they arrive.  A device works up to three tiles ahead of the writer: the library uses four pinned staging
    buffers in turn, so a device's i-th fetch reuses the buffer of its (i-4)-th; every tile's write (on the sink's I/O
    threads) is acknowledged when it is over and a device collects the acknowledgement of tile i-4 before it fetches tile i."""
    import queue

    from . import bgzf, capi
    from .rowgen import RowGenerator

    n, G = len(tb), len(devices)
    q = 1
    if rng == "mt19937" and len(st):
        from math import gcd

        q = 156 // gcd(len(st), 156)  # tiles start on a wave boundary of the sequential stream
    tile_rows = max(q, tile_rows // q * q)
    tiles = [(lo, min(n, lo + tile_rows)) for lo in range(0, n, tile_rows)]
    tb.prefixes()  # the rows' fixed columns are formatted once; the tiles' tables are views of them
    slots = [queue.Queue() for _ in devices]
    acks = [threading.Semaphore(0) for _ in devices]
    stop, written = threading.Event(), threading.Event()
    trace = os.environ.get("DNAF_TRACE_TILES")

    def produce(g):
        try:
            with RowGenerator(device=devices[g], seed=seed, rng=rng, mt_state=mt_state) as gen:
                gen.set_samples(st)
                for k in range(g, len(tiles), G):
                    lo, hi = tiles[k]
                    t0 = time.time()
                    gen.set_snps(tb.slice(lo, hi), rng_row0=lo)
                    gen.ctx.generate(0, hi - lo, capi.OUT_TEXT if host_bgzf and bed_path is None else capi.OUT_TEXT | capi.OUT_PACKED)
                    bed = bytes(gen.ctx.export_bed()) if bed_path else None
                    if not host_bgzf:
                        gen.ctx.bgzf_compress(reuse_tables=k >= G)
                    t1 = time.time()
                    if k >= 4 * G:  # this fetch overwrites the staging buffer of this device's tile k - 4G: written by now?
                        while not acks[g].acquire(timeout=0.5):
                            if stop.is_set():
                                return
                    t2 = time.time()
                    if host_bgzf:
                        gen.ctx.sync()
                        view = gen.ctx.fetch_text()
                    else:
                        view = gen.ctx.fetch_bgzf()
                    slots[g].put((k, view, bed))
                    if trace:
                        print("tile %d on device slot %d: produced in %.3f s, waited %.3f s for the writer, copied out in %.3f s" % (k, g, t1 - t0, t2 - t1, time.time() - t2), flush=True)
                written.wait()  # the views handed over live in this context's pinned buffers: keep it until the writer is through
        except BaseException as e:  # surfaced in the writer, never swallowed
            slots[g].put((-1, e, None))

    ths = [threading.Thread(target=produce, args=(g,), daemon=True) for g in range(G)]
    for t in ths:
        t.start()
    bed_f = open(bed_path, "wb") if bed_path else None
    try:
        if bed_f:
            bed_f.write(BED_MAGIC)
        with bgzf.BgzfWriter(path, compresslevel=compresslevel, threads=threads) as f:
            f.write(header)
            for k, (lo, hi) in enumerate(tiles):
                got, view, bed = slots[k % G].get()
                if got != k:
                    raise view if got == -1 else RuntimeError("tile %d arrived in place of tile %d" % (got, k))
                fut = None
                if host_bgzf:
                    f.write(view)  # (copied into the compressor's blocks before write returns)
                else:
                    fut = f.write_members(view)
                if fut is None:
                    acks[k % G].release()
                else:
                    fut.add_done_callback(lambda _f, g=k % G: acks[g].release())
                if bed_f:
                    bed_f.write(bed)
                progress(hi - lo)
    finally:
        stop.set()
        written.set()
        for t in ths:
            t.join()
        if bed_f:
            bed_f.close()


BED_MAGIC = bytes([0x6C, 0x1B, 0x01])  # PLINK 1 .bed, SNP-major


def write_vcf(path, header, st, tb, seed, rng, devices, compresslevel, tile_rows=0, threads=0, host_bgzf=False,
              mt_state=None, chunk_restart=False, bed_path=None, workers=1):
    """population.vcf.gz = header + one body line per SNP (output_vcf_population, pop_factory.py:385-415).
    With several devices the tiles of SNP rows are dealt round-robin to the GPUs and this thread appends the finished
    tiles' BGZF members in order (_write_round_robin): the reference's stripe + ordered merge, pop_factory.py:426-469."""
    from . import bgzf

    n = len(tb)
    if n == 0:
        raise Exception("No SNPs to Process! Exiting.")  # pop_factory.py:395-396
    tile_rows = tile_rows or _auto_tile_rows(len(st), n, rng)
    threads = threads or max(1, min(16, (os.cpu_count() or 4) // max(1, len(devices))))
    done = [0]
    lock = threading.Lock()
    t0 = time.time()

    def progress(m):
        with lock:
            done[0] += m
            print("%s Output %i/%i VCF lines." % (datetime.now().strftime("%Y-%m-%d %H:%M:%S"), done[0], n), flush=True)

    print("Outputing VCF lines", flush=True)
    # units of work: (first row, end row, RNG row of the first row).  By default one stream covers the whole table;
    # with chunk_restart the stream restarts at every reference work chunk, as the reference's does.
    units = [(lo, hi, 0) for lo, hi in reference_chunks(n)] if chunk_restart else [(0, n, 0)]
    if len(devices) == 1:
        bed = open(bed_path, "wb") if bed_path else None
        if bed:
            bed.write(BED_MAGIC)
        with bgzf.BgzfWriter(path, compresslevel=compresslevel, threads=threads) as f:
            f.write(header)
            for lo, hi, r0 in units:
                # the reference uses one worker for a chunk with fewer rows than workers (pop_factory.py:422-424)
                _generate_shard(devices[0], seed, rng, mt_state, st, tb, lo, hi, r0, tile_rows, f, progress, host_bgzf, bed,
                                rng_divisor=workers if hi - lo >= workers else 1)
        if bed:
            bed.close()
    else:
        # Several devices: the reference's own shape (pop_factory.py:426-469) -- the work is dealt round-robin (there:
        # rows to worker processes, here: tiles of rows to GPUs) and ONE writer appends the results in order.  Every
        # byte is written once, straight into population.vcf.gz; rows keep their global RNG row, so the bytes do not
        # depend on how many devices took part.
        if chunk_restart or workers > 1:
            raise SystemExit("--compat-chunk-restart / --compat-workers are only supported on one device")
        _write_round_robin(path, header, st, tb, seed, rng, devices, compresslevel, tile_rows, threads, host_bgzf, mt_state,
                           bed_path, progress)
    print("Finished VCF file output. (%.1f s)" % (time.time() - t0), flush=True)


def generate_population(args):
    from . import population as pop

    try:
        devices = [int(d) for d in args.devices.split(",") if d.strip() != ""]
    except ValueError:
        raise SystemExit("--devices takes comma-separated GPU ids, got %r" % args.devices)
    if not devices or min(devices) < 0:
        raise SystemExit("--devices needs at least one GPU id (>= 0), got %r" % args.devices)
    if not args.host_bgzf and args.compression_level != 6:
        # the reference hands -z to zlib; the on-GPU encoder has one setting (between zlib -1 and -6 in size)
        print("note: -z %d does not apply to the on-GPU BGZF encoder (its output lies between zlib -1 and zlib -6 in size); "
              "use --host-bgzf to compress with zlib at that level like the reference" % args.compression_level, flush=True)
    t_start = time.time()
    seed = args.seed if args.seed is not None else int(datetime.now().strftime("%H%M%S"))  # pop_factory.py:235
    numpy.random.seed(seed)
    if args.seed is not None:
        random.seed(seed)
    if args.outdir:
        outdir = args.outdir if args.outdir.endswith(os.path.sep) else args.outdir + os.path.sep
    else:
        outdir = os.path.join(OUTPUT_DIR, datetime.now().strftime("%Y%m%d%H%M")) + os.path.sep
    os.makedirs(outdir, exist_ok=True)

    if args.snps_file:
        tb = pop.load_snps_file(args.snps_file)
    elif args.generate_snps:
        if not args.max_snps or args.max_snps < 1:
            raise SystemExit("-x <max_snps> is required when SNPs are simulated")
        tb = pop.generate_snp_table(args.max_snps, min_maf=args.min_freq)
    else:
        raise SystemExit("-l (load from the RefSNP database) is not supported: the database path is outside this "
                         "generator's scope; use simulated SNPs or --snps_file")
    tb = pop.sort_snp_table(tb)
    snps_writer = None
    if not args.snps_file:
        # snps.json.gz only reads the table: it is written on a side thread while the deleterious picks, the sample
        # files and the VCF are made (its formatter and gzip workers release the GIL)
        snps_error = []

        def write_snps():
            try:
                t = time.time()
                pop.write_snps_file(tb, outdir + "snps.json.gz")
                print("Time to write snps file %0.4f seconds" % (time.time() - t), flush=True)
            except BaseException as e:  # re-raised after the join below
                snps_error.append(e)

        snps_writer = threading.Thread(target=write_snps)
        snps_writer.start()

    if args.deleterious_file:
        groups = pop.load_deleterious_json(args.deleterious_file, int_keys=not args.compat_deleterious_file_bug)
    else:
        t = time.time()
        groups = pop.pick_deleterious(tb, args.deleterious_config)
        pop.write_deleterious_json(groups, outdir + "deleterious.json")
        print("Elapsed pick_deleterious_snps %0.2f sec" % (time.time() - t), flush=True)

    if len(tb) == 0:
        raise Exception("No SNPs to Process! Exiting.")
    st = pop.build_samples(groups, args.control_size, args.size, args.male_odds, outdir, offset=args.offset or 0)
    header = pop.vcf_header(st.person_id)
    mt_state = None
    if args.rng == "mt19937":
        # the reference's worker forks here and keeps drawing from the parent's numpy stream (pop_factory.py:432-434)
        state = numpy.random.get_state()
        mt_state = (state[1], int(state[2]))
    write_vcf(outdir + "population.vcf.gz", header, st, tb, seed, args.rng, devices, args.compression_level,
              tile_rows=args.tile_rows, threads=args.threads, host_bgzf=args.host_bgzf, mt_state=mt_state,
              chunk_restart=args.compat_chunk_restart or args.compat_workers, bed_path=outdir + "population.bed" if args.plink else None,
              workers=max(1, args.num_processes) if args.compat_workers else 1)
    if args.plink:
        pop.write_bim(tb, outdir + "population.bim")
    if snps_writer is not None:
        snps_writer.join()
        if snps_error:
            raise snps_error[0]
    print("Finished Generating Population in %0.4f secs." % (time.time() - t_start), flush=True)
    return outdir


def main(sys_args):
    args = parse_cmd_args(sys_args)
    return generate_population(args)


if __name__ == "__main__":
    main(sys.argv[1:])
