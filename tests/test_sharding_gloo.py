"""N > 1 path on CPU: two gloo ranks shard a sequence with the same helpers bench.py uses, each rank
runs the (oracle) hot path on its shard, and the summary reduction (SUM of units, MAX of time) matches
the single-process run.  Not gpu."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_scans, q):
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
    lo, hi = shard.block_range(n_scans, rank, world)
    counts = []
    for i in range(lo, hi):
        pts, _, _ = synth.make_scan(3, i, "PARK")
        x = pts.numpy()
        o = orc.patchwork(P, x, 1)
        b = orc.bin(P, x[o["nonground_idx"]], True)
        v = orc.voxelize(P, b["apri"])
        counts.append([x.shape[0], len(o["ground_idx"]), len(b["apri"]), len(v["vox_key"])])
    counts = np.asarray(counts, np.int64).reshape(-1, 4)
    dt, scans, pts_total = shard.aggregate(dist, torch.device("cpu"), 1.0 + rank, hi - lo, int(counts[:, 0].sum()))
    # gather the per-scan counters to check nothing was lost or duplicated
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, counts.tolist()))
    rr = shard.round_robin(n_scans, rank, world)
    allrr = [None] * world
    dist.all_gather_object(allrr, rr.tolist())
    if rank == 0:
        q.put(dict(dt=dt, scans=scans, pts=pts_total, gathered=gathered, rr=allrr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction(oracle, scvod):
    import shard
    import synth
    n_scans, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scans, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["dt"] == 2.0 and res["scans"] == n_scans          # MAX of (1.0, 2.0), SUM of shard sizes
    blocks = sorted((lo, hi) for lo, hi, _ in res["gathered"])
    assert blocks[0][0] == 0 and blocks[-1][1] == n_scans and blocks[0][1] == blocks[1][0]
    assert sorted(sum(res["rr"], [])) == list(range(n_scans))
    P = scvod.make_params("parkinglot")
    single = []
    for i in range(n_scans):
        pts, _, _ = synth.make_scan(3, i, "PARK")
        x = pts.numpy()
        o = oracle.patchwork(P, x, 1)
        b = oracle.bin(P, x[o["nonground_idx"]], True)
        v = oracle.voxelize(P, b["apri"])
        single.append([x.shape[0], len(o["ground_idx"]), len(b["apri"]), len(v["vox_key"])])
    sharded = sum((c for _, _, c in sorted(res["gathered"])), [])
    assert sharded == single
    assert res["pts"] == sum(s[0] for s in single)


def test_block_range_properties():
    import shard
    for n in (0, 1, 7, 2761):
        for w in (1, 2, 3, 8):
            rs = [shard.block_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def _boundary_worker(rank, world, port, n_scans, q):
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
    lo, hi = shard.block_range(n_scans, rank, world)

    def tables(i):
        pts, _, pose = synth.make_scan(3, i, "PARK")
        x = pts.numpy()
        o = orc.patchwork(P, x, 1)
        b = orc.bin(P, x[o["nonground_idx"]], True)
        v = orc.voxelize(P, b["apri"])
        return b["apri"], v["vox_key"], pose

    first_apri, first_keys, _ = tables(lo)
    nxt_keys, nxt_labels = shard.exchange_boundary_table(dist, torch.device("cpu"), first_keys, np.arange(len(first_keys), dtype=np.int32) % 7 - 1)
    out = None
    if rank < world - 1:
        # the probe of this block's LAST scan against the NEXT block's first table (ssc.cpp:1274-1321)
        apri, _, pose_a = tables(hi - 1)
        _, _, pose_b = tables(hi)                      # only the pose is needed locally (poses are replicated input)
        T = orc.pose_delta(pose_a, pose_b)
        m = np.arange(0, len(apri), 5)
        xyzi = np.stack([apri["x"][m], apri["y"][m], apri["z"][m], apri["intensity"][m]], 1).astype(np.float32)
        offs = np.arange(0, len(m) + 1, 50, dtype=np.int32)
        offs[-1] = len(m)
        hit, uq, ub = orc.track_probe(P, xyzi, offs, T, nxt_keys, nxt_labels)
        out = (hi - 1, hit.tolist(), ub.tolist(), nxt_keys.tolist(), nxt_labels.tolist())
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_boundary_pair_uses_the_neighbours_table(oracle, scvod):
    """the one real exchange step of the path: a block's last scan is probed against the first voxel table of the next
    block, received point-to-point; the result equals the single-process probe of the same pair."""
    import synth
    n_scans, world = 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 400)
    procs = [ctx.Process(target=_boundary_worker, args=(r, world, port, n_scans, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and res[0] is not None          # the last rank has no right neighbour
    last, hit, ub, keys, labels = res[0]
    assert last == 1                                      # scans [0, 2) on rank 0, [2, 4) on rank 1
    P = scvod.make_params("parkinglot")

    def tables(i):
        pts, _, pose = synth.make_scan(3, i, "PARK")
        x = pts.numpy()
        o = oracle.patchwork(P, x, 1)
        b = oracle.bin(P, x[o["nonground_idx"]], True)
        return b["apri"], oracle.voxelize(P, b["apri"])["vox_key"], pose

    apri, _, pose_a = tables(1)
    _, keys2, pose_b = tables(2)
    assert keys == keys2.tolist() and labels == (np.arange(len(keys2)) % 7 - 1).tolist()
    m = np.arange(0, len(apri), 5)
    xyzi = np.stack([apri["x"][m], apri["y"][m], apri["z"][m], apri["intensity"][m]], 1).astype(np.float32)
    offs = np.arange(0, len(m) + 1, 50, dtype=np.int32)
    offs[-1] = len(m)
    rhit, _, rub = oracle.track_probe(P, xyzi, offs, oracle.pose_delta(pose_a, pose_b), keys2, (np.arange(len(keys2)) % 7 - 1).astype(np.int32))
    assert hit == rhit.tolist() and ub == rub.tolist() and (rhit >= 0).any()


def _map_worker(rank, world, port, n_scans, q):
    sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shard
    lo, hi = shard.block_range(n_scans, rank, world)
    rng = [np.random.default_rng(100 + i) for i in range(n_scans)]
    local = [rng[i].normal(size=(50 + 37 * i, 4)).astype(np.float32) for i in range(lo, hi)]   # ragged per-scan clouds
    local = np.concatenate(local) if local else np.zeros((0, 4), np.float32)
    full = shard.gather_static_map(dist, torch.device("cpu"), local)
    only_root = shard.gather_static_map(dist, torch.device("cpu"), local, root_only=True)
    gathered = [None] * world
    dist.all_gather_object(gathered, (full.tobytes(), only_root is None))
    if rank == 0:
        q.put((gathered, only_root.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_static_map_gather_keeps_scan_order():
    """three ranks (one of them with an empty block): the gathered map is the single-process accumulation, bit for bit"""
    n_scans, world = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_map_worker, args=(r, world, port, n_scans, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, root = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.concatenate([np.random.default_rng(100 + i).normal(size=(50 + 37 * i, 4)).astype(np.float32) for i in range(n_scans)])
    assert all(g[0] == ref.tobytes() for g in gathered) and root == ref.tobytes()
    assert [g[1] for g in gathered] == [False, True, True]
