"""Sharding of a multi-sequence job across GPUs (one process per GPU) -- shared by bench.py and the gloo tests.

What shards: Patchwork / binning / voxel descriptors / clustering are independent per scan (SSC::reset clears all per-scan
state, src/ssc.cpp:79-86).  What does not: SSC::segDF tracks frame i against frame i + 1 IN ORDER and every call mutates
the successor before the next call walks it (src/ssc.cpp:1449-1451, 1354-1419) -- a chain per sequence, which the device
replays exactly (csrc/scvod_chain.hip).  A chain cannot cross ranks without serialising them, so the unit of work is the
SEQUENCE: whole sequences are dealt to the ranks (longest first, to the least loaded rank); inside a rank its sequences
(and the `skip` interleaved sub-sequences of each, config skip_) are independent chains of one batch.  No tracking data
crosses ranks.  The one exchange step of the path is the static map: every rank accumulates its scans into its own map
and the maps are reduce-scattered (records grouped by owner rank, one all-to-all of equal-sized padded slots over
RCCL/xGMI); the cell rule is order-independent, so the merged map is bit-identical to the single-rank map."""
import numpy as np

SEQ_ORDER = (5, 0, 2, 8, 9, 10, 1, 6, 7, 3, 4)  # seq 05 first: one rank = BASELINE.json configs[1]


def weak_scaling_sequences(n_sequences, scans_per_sequence, seq_order=SEQ_ORDER):
    """bench.py's job: n_sequences sequences of scans_per_sequence scans each (seq 05-, 00-, 02-, ... shaped: the sequence
    id seeds the scene), normally one per rank: weak scaling, one rank = BASELINE.json configs[1]."""
    return [(int(seq_order[k % len(seq_order)]) + 100 * (k // len(seq_order)), 0, int(scans_per_sequence)) for k in range(int(n_sequences))]


def kitti_sequences(seq_len, seq_order=SEQ_ORDER):
    """BASELINE.json configs[3]: SemanticKITTI seq 00-10 at their real lengths."""
    return [(int(q), 0, int(seq_len[q])) for q in seq_order]


def plan_job(world, sequences, skip=1):
    """sequences: [(seq_id, first_idx, count)].  Whole sequences are dealt longest first to the rank with the fewest scans
    so far (ties: lowest rank) -- deterministic, every rank computes the same plan.  Tracking pairs scan i with scan
    i + skip of ITS sequence (the reference loads every skip-th scan, ssc.cpp:1041,1443, and tracks consecutive loaded
    frames; pairing (i, i + skip) for every i runs the `skip` interleaved sub-sequences in one pass).

    Returns a list (one entry per rank) of dicts:
      sequences  the (seq_id, first_idx, count) triples of the rank, in processing order
      scans      [(seq_id, idx)] local scans in processing order (the rank's sequences, concatenated)
      next_scan  int32 [n]: local index of the successor, -1 = none (the last `skip` scans of a sequence)
    """
    world, skip = int(world), int(skip)
    assert world >= 1 and skip >= 1
    order = sorted(range(len(sequences)), key=lambda k: (-int(sequences[k][2]), k))
    load = [0] * world
    mine = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        mine[r].append(k)
        load[r] += int(sequences[k][2])
    ranks = []
    for r in range(world):
        scans, nxt, seqs = [], [], []
        for k in sorted(mine[r]):  # processing order: the job's own order
            q, first, count = (int(v) for v in sequences[k])
            base = len(scans)
            seqs.append((q, first, count))
            for j in range(count):
                scans.append((q, first + j))
                nxt.append(base + j + skip if j + skip < count else -1)
        ranks.append(dict(sequences=seqs, scans=scans, next_scan=np.asarray(nxt, np.int32).reshape(-1), skip=skip))
    return ranks


def reduce_scatter_map(dist, send, recv=None):
    """Static-map reduce over xGMI as ONE all-to-all of equal-sized slots: `send` [world, cap, 2] int64, slot j = the records
    this rank holds for owner j (scvod_map_export_parts_padded: padded with key -1, which scvod_map_merge skips).  Returns
    [world, cap, 2]: slot j = what rank j held for THIS rank.  No sizes travel and nothing is read on the host: the slots
    are fixed.  RCCL: all_to_all_single on device tensors; gloo (CPU tests, same-device dry runs) has no all-to-all: one
    isend / irecv pair per peer, staged through the host when the tensors live on a device."""
    import torch
    if dist is None or not dist.is_initialized():
        return send
    world, rank = dist.get_world_size(), dist.get_rank()
    assert send.shape[0] == world and send.is_contiguous()
    if recv is None:
        recv = torch.empty_like(send)
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(recv.view(world, -1), send.view(world, -1))
        return recv
    staged = send.is_cuda
    sb = send.cpu() if staged else send
    rb = torch.empty_like(sb)
    rb[rank].copy_(sb[rank])
    ops = []
    for j in range(world):
        if j != rank:
            ops.append(dist.P2POp(dist.isend, sb[j], j))
            ops.append(dist.P2POp(dist.irecv, rb[j], j))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged:
        recv.copy_(rb)
        return recv
    return rb


def aggregate(dist, device, seconds, scans, points):
    """MAX over ranks of the timed seconds, SUM of the processed units.  dist may be None (1 rank, no process group)."""
    if dist is None or not dist.is_initialized():
        return float(seconds), float(scans), float(points)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    a = torch.tensor([float(scans), float(points)], dtype=torch.float64, device=device)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    return float(t.item()), float(a[0].item()), float(a[1].item())
