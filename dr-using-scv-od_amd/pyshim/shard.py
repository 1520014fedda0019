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


SEQ_ORDER = (5, 0, 2, 8, 9, 10, 1, 6, 7, 3, 4)  # seq 05 first: one rank = BASELINE.json configs[1], eight ranks ~ configs[3] (seq 00-10)


def plan_job(world, scans_per_rank, seq_len, blocks_per_rank=8, seq_order=SEQ_ORDER, skip=1):
    """Sequence-sharded job (BASELINE.json configs[3]): the scans of the sequences `seq_order` (lengths `seq_len`), in
    order, truncated to world * scans_per_rank, cut into world * blocks_per_rank blocks of consecutive scans that are dealt
    round-robin: block k belongs to rank k % world.  Tracking (SSC::tracking, ssc.cpp:1449-1451) pairs a scan with the scan
    `skip` indices later in ITS sequence (config `skip_`: the reference loads every skip-th scan, ssc.cpp:1041,1443,
    and tracks consecutive loaded frames; pairing (i, i + skip) for every i runs its `skip` interleaved sub-sequences in
    one pass); where that successor lives on another rank the owner of the successor sends the voxel table of that scan
    to the left neighbour (`skip` messages per block boundary).

    Returns a list (one entry per rank) of dicts:
      scans      [(seq, idx)] local scans in processing order (the rank's blocks, concatenated)
      next_scan  int32 [n]: local index of the successor, -1 = none (last scan of a sequence / of the job), -2 - e = the
                 e-th table received from rank (r + 1) % world
      send_scans local indices whose tables go to rank (r - 1) % world, in message order
      n_recv     number of tables received from rank (r + 1) % world
    """
    total = int(world) * int(scans_per_rank)
    glob = []
    for q in seq_order:
        for i in range(int(seq_len[q])):
            if len(glob) == total:
                break
            glob.append((q, i))
    if len(glob) < total:
        raise ValueError(f"the sequences hold {len(glob)} scans, fewer than world * scans_per_rank = {total}")
    n_blocks = int(world) * int(blocks_per_rank)
    cuts = [(k * total) // n_blocks for k in range(n_blocks + 1)]
    owner = np.empty(total, np.int32)
    local = np.empty(total, np.int32)
    ranks = [dict(scans=[], next_scan=None, send_scans=[], n_recv=0, blocks=[]) for _ in range(world)]
    for k in range(n_blocks):
        r = k % world
        for g in range(cuts[k], cuts[k + 1]):
            owner[g] = r
            local[g] = len(ranks[r]["scans"])
            ranks[r]["scans"].append(glob[g])
        ranks[r]["blocks"].append((cuts[k], cuts[k + 1]))
    for r in range(world):
        ranks[r]["next_scan"] = np.full(len(ranks[r]["scans"]), -1, np.int32)
    skip = int(skip)
    assert skip >= 1 and all(c1 - c0 >= skip for c0, c1 in zip(cuts[:-1], cuts[1:])), "blocks shorter than the tracking stride"
    for g in range(total - skip):
        h = g + skip
        if glob[h][0] != glob[g][0]:
            continue  # one of the last scans of its sequence
        r, r2 = int(owner[g]), int(owner[h])
        if r == r2:
            ranks[r]["next_scan"][local[g]] = local[h]
        else:
            assert r2 == (r + 1) % world
            ranks[r]["next_scan"][local[g]] = -2 - ranks[r]["n_recv"]
            ranks[r]["n_recv"] += 1
            ranks[r2]["send_scans"].append(int(local[h]))
    for r in ranks:
        r["skip"] = skip
    return ranks


def torch_empty_like_cpu(t):
    import torch
    return torch.empty(t.shape, dtype=t.dtype, device="cpu")


def exchange_tables(dist, send_buf, recv_buf):
    """One step of the boundary exchange: `send_buf` [n_send, cap, 4] int32 (tables exported by scvod_batch_export_table)
    goes to rank - 1, `recv_buf` [n_recv, cap, 4] is filled by rank + 1 (ring, device buffers: RCCL P2P over xGMI on the
    GPUs, gloo in the CPU tests).  Message counts match by construction of plan_job."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    staged = dist.get_backend() == "gloo" and send_buf is not None and send_buf.is_cuda  # CPU dry runs of the GPU job
    sb = send_buf.cpu() if staged else send_buf
    rb = torch_empty_like_cpu(recv_buf) if staged else recv_buf
    ops = []
    if sb is not None and sb.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, sb, (rank - 1) % world))
    if rb is not None and rb.shape[0] > 0:
        ops.append(dist.P2POp(dist.irecv, rb, (rank + 1) % world))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged and rb.shape[0] > 0:
        recv_buf.copy_(rb)


def gather_map_records(dist, records, root=0):
    """Static-map reduce: every rank's exported cell records [n, 2] int64 travel to `root` (sizes first, then one padded
    gather; padding key -1 = ~0 is skipped by scvod_map_merge).  Returns on the root the list of the OTHER ranks' padded
    record tensors, elsewhere an empty list.  Device tensors (RCCL) or CPU tensors (gloo)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    home = records.device
    if dist.get_backend() == "gloo" and records.is_cuda:  # CPU dry runs of the GPU job
        records = records.cpu()
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = max(int(max(int(t.item()) for t in sizes)), 1)
    mine = torch.full((cap, 2), -1, dtype=torch.int64, device=records.device)
    mine[:records.shape[0]] = records
    if rank == root:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, parts, dst=root)
        return [p.to(home) for r, p in enumerate(parts) if r != root]
    dist.gather(mine, None, dst=root)
    return []


def reduce_scatter_map(dist, records, counts):
    """Static-map reduce over xGMI as an all-to-all: `records` [n, 2] int64 grouped by owner rank (scvod_map_export_parts),
    `counts` the group sizes.  Every rank keeps its own group and receives the groups the other ranks hold for it (sizes
    first, then one point-to-point pair per peer: 7 links x 1/8 of a rank's map instead of 7 maps converging on rank 0).
    Returns the list of record tensors this rank owns (its own group first).  Device tensors (RCCL) or, under gloo, staged
    through the host."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [records]
    rank, world = dist.get_rank(), dist.get_world_size()
    home = records.device
    staged = dist.get_backend() == "gloo" and records.is_cuda
    rec = records.cpu() if staged else records
    mine = torch.tensor(counts, dtype=torch.int64, device=rec.device)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)                               # allc[j][r] = what rank j holds for rank r
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + int(c))
    out = [rec[offs[rank]:offs[rank + 1]]]
    ops, recvs = [], []
    for j in range(world):
        if j == rank:
            continue
        if counts[j]:
            ops.append(dist.P2POp(dist.isend, rec[offs[j]:offs[j + 1]].contiguous(), j))
        n_in = int(allc[j][rank].item())
        if n_in:
            buf = torch.empty((n_in, 2), dtype=torch.int64, device=rec.device)
            recvs.append(buf)
            ops.append(dist.P2POp(dist.irecv, buf, j))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    out += recvs
    return [t.to(home) for t in out] if staged else out


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
