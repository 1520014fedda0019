"""The sequence-sharded job of bench.py (pyshim/shard.py: plan_job, exchange_tables, gather_map_records) on CPU: the plan's
invariants for 1 .. 8 ranks, and three gloo ranks running the two real exchange steps of the path -- boundary tables to the
left neighbour, map records to the root -- with payloads that name their scan, so every message is checked to arrive where
the single-process job would look it up.  Not gpu."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))

SEQ_LEN = {5: 23, 0: 31, 2: 40, 8: 200, 9: 10, 10: 10, 1: 10, 6: 10, 7: 10, 3: 10, 4: 10}


def test_plan_invariants():
    import shard
    for world in (1, 2, 3, 8):
        for skip in (1, 5):
            spr = 24
            plan = shard.plan_job(world, spr, SEQ_LEN, blocks_per_rank=3, skip=skip)
            seen = [q for r in plan for q in r["scans"]]
            glob = [(q, i) for q in shard.SEQ_ORDER for i in range(SEQ_LEN[q])][: world * spr]
            assert sorted(seen) == sorted(glob) and len(set(seen)) == len(seen)         # every scan exactly once
            assert max(len(r["scans"]) for r in plan) - min(len(r["scans"]) for r in plan) <= 3
            for r, p in enumerate(plan):
                assert len(p["send_scans"]) == plan[(r - 1) % world]["n_recv"]
                ext = 0
                for j, (q, i) in enumerate(p["scans"]):
                    v = int(p["next_scan"][j])
                    succ = (q, i + skip)
                    if succ not in set(glob):
                        assert v == -1                                                  # end of its sequence / of the job
                    elif v >= 0:
                        assert p["scans"][v] == succ                                    # local successor
                    else:
                        assert v == -2 - ext                                            # e-th table from the right neighbour
                        right = plan[(r + 1) % world]
                        assert right["scans"][right["send_scans"][ext]] == succ
                        ext += 1
                assert ext == p["n_recv"]
            if world == 1:
                assert all(p["n_recv"] == 0 and not p["send_scans"] for p in plan)


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shard
    skip, cap = 2, 16
    plan = shard.plan_job(world, 12, SEQ_LEN, blocks_per_rank=2, skip=skip)[rank]
    # a "table" that names its scan: header {n, seq, idx, 0}, then n records
    send = torch.zeros((len(plan["send_scans"]), cap, 4), dtype=torch.int32)
    for m, s in enumerate(plan["send_scans"]):
        qq, i = plan["scans"][s]
        n = 3 + (i % 5)
        send[m, 0] = torch.tensor([n, qq, i, 0])
        send[m, 1:1 + n, 0] = torch.arange(n) + 1000 * i
    recv = torch.full((plan["n_recv"], cap, 4), -7, dtype=torch.int32)
    for _ in range(2):  # a second step reuses the buffers
        shard.exchange_tables(dist, send, recv)
    ok = True
    for j, (qq, i) in enumerate(plan["scans"]):
        v = int(plan["next_scan"][j])
        if v <= -2:
            h = recv[-2 - v]
            n = int(h[0, 0])
            ok &= (int(h[0, 1]), int(h[0, 2])) == (qq, i + skip) and n == 3 + ((i + skip) % 5)
            ok &= bool((h[1:1 + n, 0] == torch.arange(n) + 1000 * (i + skip)).all())
    # map records: every rank contributes rank + 2 records tagged with its rank; the root sees all of them, padded with -1
    rec = torch.stack([torch.arange(rank + 2, dtype=torch.int64) + 100 * rank, torch.full((rank + 2,), rank, dtype=torch.int64)], 1)
    others = shard.gather_map_records(dist, rec, root=0)
    got = None
    if rank == 0:
        allrec = torch.cat([rec] + others)
        allrec = allrec[allrec[:, 0] != -1]
        got = sorted(map(tuple, allrec.tolist()))
    else:
        ok &= others == []
    # reduce-scatter of the map: rank r holds r + j + 1 records for owner j (tagged r, j); every owner ends up with its own
    counts = [rank + j + 1 for j in range(world)]
    rs = torch.cat([torch.stack([torch.full((c,), 1000 * rank + j, dtype=torch.int64), torch.arange(c, dtype=torch.int64)], 1) for j, c in enumerate(counts)])
    parts = shard.reduce_scatter_map(dist, rs, counts)
    own = torch.cat(parts)
    want_own = sorted((1000 * r + rank, k) for r in range(world) for k in range(r + rank + 1))
    ok &= sorted(map(tuple, own.tolist())) == want_own
    res = [None] * world
    dist.all_gather_object(res, (bool(ok), got, plan["n_recv"]))
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_three_ranks_exchange_tables_and_reduce_the_map():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert sum(r[2] for r in res) > 0                     # the plan really crosses ranks
    want = sorted((k + 100 * r, r) for r in range(world) for k in range(r + 2))
    assert res[0][1] == want
