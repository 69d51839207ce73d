"""The N > 1 path on CPU: two gloo ranks each produce their SNP range (with the CPU oracle standing in for the
kernel -- the checker, as in every test) using the product's shard bounds and RNG-row convention; rank 0 checks
that the concatenation is the unsharded output and that the timing reduction is the max over ranks."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from dna_factory_amd.sharding import max_over_ranks, shard_range
    from helpers import random_case
    from oracle import oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(77)  # same tables on every rank
        snps, samples = random_case(rs, 101, 57)
        rec = orc.snp_records(snps)
        sex, ctrl, off, ids = orc.sample_arrays(samples)
        lo, hi = shard_range(len(snps), rank, world)
        text, _, packed = orc.generate(rec[lo:hi], sex, ctrl, off, ids, rng_mode=orc.RNG_PHILOX, seed=5, rng_row0=lo)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, text, packed))
        slowest = max_over_ranks(1.0 + rank, dist)
        if rank == 0:
            whole_text, _, whole_packed = orc.generate(rec, sex, ctrl, off, ids, rng_mode=orc.RNG_PHILOX, seed=5)
            ok = b"".join(g[2] for g in gathered) == whole_text
            ok = ok and np.array_equal(np.concatenate([g[3] for g in gathered]), whole_packed)
            ok = ok and [g[0] for g in gathered] == [0, 50] and gathered[-1][1] == 101
            q.put((ok, slowest))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_the_whole_job():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, slowest = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert slowest == 2.0


def test_shard_bounds_cover_every_row_once():
    sys.path.insert(0, REPO)
    from dna_factory_amd.sharding import shard_bounds, shard_range

    for n in (0, 1, 7, 8, 5000000, 40000000):
        for w in (1, 2, 3, 4, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
            assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1
            assert [shard_range(n, r, w) for r in range(w)] == [(b[r], b[r + 1]) for r in range(w)]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
