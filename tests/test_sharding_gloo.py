"""N > 1 path on CPU: two gloo ranks take their share of a three-sequence job from the plan bench.py uses (whole sequences
per rank), each runs the (oracle) path INCLUDING the sequential tracking chain on its sequences, the per-rank static maps
(numpy restatement of the cell rule) are reduce-scattered through shard.reduce_scatter_map, and per-scan results, the
merged map and the summary reduction (SUM of units, MAX of time) equal the single-process run.  Not gpu."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOB = [(3, 0, 4), (6, 10, 3), (7, 5, 2)]  # (sequence, first scan, scans)
SKIP = 1
LEAF = 0.5


def _cells_of(x, lab, pose, scvod_py):
    """static cells of one scan: world cell key -> smallest packed offset (order-independent rule of csrc/scvod_map.hip, coarser)"""
    keep = (lab != 1) & (lab != 3)
    T = scvod_py.pose_matrix(pose)
    p = x[keep]
    w = np.stack([((T[4 * i] * p[:, 0] + T[4 * i + 1] * p[:, 1]) + T[4 * i + 2] * p[:, 2]) + T[4 * i + 3] for i in range(3)], 1)
    c = np.floor(w / LEAF).astype(np.int64) + (1 << 20)
    key = (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    off = np.clip(((w / LEAF - np.floor(w / LEAF)) * 1024).astype(np.int64), 0, 1023)
    val = (off[:, 0] << 20) | (off[:, 1] << 10) | off[:, 2]
    return key, val


def _run_sequences(seqs, orc, P, scvod_py, synth):
    """the oracle path over whole sequences: per-scan counters, per-point labels (sequential chain), static cells"""
    per_scan, keys, vals = {}, [], []
    for (q, first, count) in seqs:
        scans = [synth.make_scan(q, first + j, "PARK") for j in range(count)]
        x = np.concatenate([s[0].numpy() for s in scans])
        offs = np.concatenate([[0], np.cumsum([len(s[0]) for s in scans])]).astype(np.int32)
        poses = np.asarray([s[2] for s in scans], np.float32)
        _, lab, _ = orc.time_sequence(P, x, offs, poses)
        for j in range(count):
            l = lab[offs[j]:offs[j + 1]]
            per_scan[(q, first + j)] = [int(offs[j + 1] - offs[j]), int((l == 1).sum()), int((l == 3).sum())]
            k, v = _cells_of(x[offs[j]:offs[j + 1]], l, poses[j], scvod_py)
            keys.append(k)
            vals.append(v)
    key, val = np.concatenate(keys), np.concatenate(vals)
    order = np.lexsort((val, key))
    key, val = key[order], val[order]
    first = np.ones(len(key), bool)
    first[1:] = key[1:] != key[:-1]
    return per_scan, key[first], val[first]


def _owner(key, world):
    return ((key * np.int64(0x9E3779B1)) >> 13) % world


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    import scvod_py
    import shard
    import synth
    orc = oracle_py.load()
    P = scvod_py.make_params("parkinglot")
    plan = shard.plan_job(world, JOB, skip=SKIP)[rank]
    per_scan, key, val = _run_sequences(plan["sequences"], orc, P, scvod_py, synth)
    # reduce-scatter of the map: records grouped by owner into equal padded slots (what scvod_map_export_parts_padded writes)
    own = _owner(key, world)
    cap_t = torch.tensor([max(int((own == j).sum()) for j in range(world))], dtype=torch.int64)
    dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
    cap = int(cap_t.item()) + 3
    send = torch.full((world, cap, 2), -1, dtype=torch.int64)
    for j in range(world):
        m = own == j
        send[j, :int(m.sum()), 0] = torch.from_numpy(key[m])
        send[j, :int(m.sum()), 1] = torch.from_numpy(val[m])
    recv = shard.reduce_scatter_map(dist, send)
    rec = recv.reshape(-1, 2).numpy()
    rec = rec[rec[:, 0] != -1]
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    rec = rec[order]
    first = np.ones(len(rec), bool)
    first[1:] = rec[1:, 0] != rec[:-1, 0]
    mine = rec[first]
    dt, scans, pts_total = shard.aggregate(dist, torch.device("cpu"), 1.0 + rank, len(plan["scans"]), sum(v[0] for v in per_scan.values()))
    gathered = [None] * world
    dist.all_gather_object(gathered, (plan["sequences"], per_scan, mine.tolist()))
    if rank == 0:
        q.put(dict(dt=dt, scans=scans, pts=pts_total, gathered=gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_whole_sequences_chain_and_map_reduce(oracle, scvod):
    import shard
    import synth
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_scans = sum(c for _, _, c in JOB)
    assert res["dt"] == 2.0 and res["scans"] == n_scans          # MAX of (1.0, 2.0), SUM of shard sizes
    got_seqs = sorted(s for g in res["gathered"] for s in g[0])
    assert got_seqs == sorted(JOB)                                # every sequence whole, on exactly one rank
    assert sorted(len(g[0]) for g in res["gathered"]) == [1, 2]   # longest first to the least loaded rank: {4} | {3, 2}
    P = scvod.make_params("parkinglot")
    per_scan, key, val = _run_sequences(JOB, oracle, P, scvod, synth)
    sharded = {}
    for g in res["gathered"]:
        sharded.update(g[1])
    assert sharded == per_scan                                    # counters and chain labels of every scan
    assert res["pts"] == sum(v[0] for v in per_scan.values())
    assert sum(v[1] for v in per_scan.values()) > 0               # the job has dynamic points
    merged = np.asarray(sorted(tuple(r) for g in res["gathered"] for r in g[2]), np.int64).reshape(-1, 2)
    assert np.array_equal(merged[:, 0], key) and np.array_equal(merged[:, 1], val)   # the reduce-scattered map == the single-process map
    owners = [np.unique(_owner(np.asarray([r[0] for r in g[2]], np.int64), world)) for g in res["gathered"] if g[2]]
    assert all(len(o) == 1 for o in owners)                       # every rank ended up with the cells it owns, and only those
