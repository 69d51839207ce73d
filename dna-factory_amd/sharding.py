"""SNP-range sharding: the only parallel decomposition of the path (the reference stripes rows over worker
processes, common/snp.py:82-88 / pop_factory.py:426; here every GPU gets one contiguous range of the sorted
SNP list so that its output is a contiguous slice of the VCF).  No data-path collective exists: a shard's
bytes depend only on (seed, its rows, their global row indices)."""


def shard_bounds(n_rows, world):
    """[b_0=0, b_1, ..., b_world=n_rows]: rank g owns rows [b_g, b_g+1)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    return [n_rows * g // world for g in range(world + 1)]


def shard_range(n_rows, rank, world):
    b = shard_bounds(n_rows, world)
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return b[rank], b[rank + 1]


def max_over_ranks(value, dist=None, device=None):
    """The slowest rank's time (bench.py reports rows of all ranks / this)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
