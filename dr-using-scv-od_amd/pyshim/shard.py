"""Sequence sharding across GPUs (one process per GPU) -- shared by bench.py and the gloo tests.

The hot path partitions by scan: Patchwork / binning / voxel descriptors are independent per scan
(SSC::reset clears all per-scan state, src/ssc.cpp:79-86), so a rank owns a contiguous block of scans
and no data-path collective is needed.  Only the run summary (per-rank counters, max-over-ranks time)
crosses ranks."""
import numpy as np


def block_range(n_items, rank, world):
    """Contiguous block [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    q, r = divmod(int(n_items), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def round_robin(n_items, rank, world):
    """Indices rank, rank+world, ... (BASELINE.json configs[3]: scans round-robin over GPUs)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def aggregate(dist, device, seconds, scans, points):
    """MAX over ranks of the timed seconds, SUM of the processed units.  dist may be None (1 rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds), float(scans), float(points)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    a = torch.tensor([float(scans), float(points)], dtype=torch.float64, device=device)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    return float(t.item()), float(a[0].item()), float(a[1].item())
