"""Sequence sharding across GPUs (one process per GPU) -- shared by bench.py and the gloo tests.

The hot path partitions by scan: Patchwork / binning / voxel descriptors are independent per scan
(SSC::reset clears all per-scan state, src/ssc.cpp:79-86), so a rank owns a contiguous block of scans
and no data-path collective is needed.  Only the run summary (per-rank counters, max-over-ranks time)
crosses ranks -- plus, when ONE sequence is split into contiguous blocks, the single real exchange step of the path:
the scan-vs-next-scan probe (SSC::tracking, src/ssc.cpp:1274-1321) of a block's LAST scan needs the voxel table of the
next block's FIRST scan.  That is a point-to-point message of a few thousand ints to the left neighbour
(`exchange_boundary_table`), not a collective."""
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


def exchange_boundary_table(dist, device, first_keys, first_labels=None):
    """Block-sharded sequence: every rank sends the sorted voxel key table (and optional labels) of its FIRST scan to
    rank-1 and receives the table of rank+1's first scan, which is what the tracking probe of its LAST scan runs
    against.  Returns (keys, labels) as int32 numpy arrays, or (None, None) on the last rank / without a process group.
    Point-to-point over the job's backend (RCCL on the GPUs, gloo in the CPU tests); sizes first, then payload."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None, None
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    keys = np.ascontiguousarray(first_keys, np.int32)
    labels = np.ascontiguousarray(first_labels if first_labels is not None else np.zeros(len(keys), np.int32), np.int32)
    assert len(labels) == len(keys)
    mine = torch.tensor([len(keys)], dtype=torch.int64, device=device)
    theirs = torch.zeros(1, dtype=torch.int64, device=device)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, mine, rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.irecv, theirs, rank + 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    payload = torch.from_numpy(np.stack([keys, labels])).to(device)          # [2, n] int32
    n_in = int(theirs.item()) if rank < world - 1 else 0
    incoming = torch.zeros((2, n_in), dtype=torch.int32, device=device)
    ops = []
    if rank > 0 and len(keys):
        ops.append(dist.P2POp(dist.isend, payload, rank - 1))
    if rank < world - 1 and n_in:
        ops.append(dist.P2POp(dist.irecv, incoming, rank + 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank == world - 1:
        return None, None
    got = incoming.cpu().numpy()
    return got[0].copy(), got[1].copy()


def gather_static_map(dist, device, local_xyzi, root_only=False):
    """The sequence-level merge `*map += *cloud_i` (the reference accumulates per-scan clouds into one map, ssc.cpp:554 /
    :1460-1480, in scan order): every rank contributes the world-frame static points of ITS block of scans, already in
    scan order; the result is their concatenation in rank order = the single-process accumulation order.  Variable sizes:
    one all_gather of the lengths, one all_gather of the padded payloads (RCCL over xGMI on the GPUs, gloo on CPU).
    Returns an [n, 4] float32 numpy array (None on non-root ranks when root_only)."""
    x = np.ascontiguousarray(local_xyzi, np.float32).reshape(-1, 4)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(t.item()) for t in sizes]
    cap = max(max(sizes), 1)
    mine = torch.zeros((cap, 4), dtype=torch.float32, device=device)
    if x.shape[0]:
        mine[:x.shape[0]] = torch.from_numpy(x).to(device)
    parts = [torch.empty((cap, 4), dtype=torch.float32, device=device) for _ in range(world)]
    dist.all_gather(parts, mine)
    if root_only and rank != 0:
        return None
    return np.concatenate([p[:k].cpu().numpy() for p, k in zip(parts, sizes)], 0)
