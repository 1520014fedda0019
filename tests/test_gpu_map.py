"""World-frame static map (scvod_map_*, csrc/scvod_map.hip): the device hash grid against a numpy restatement of its
definition, and the property the multi-GPU reduce relies on: shards that accumulate independently and merge their
record lists give the bit-identical map of a single shard."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracked_batch(scvod, P, kind, seq, first, count, chain=True):
    import synth
    pts, offs, poses, _ = synth.make_batch(seq, first, count, kind, device="cuda")  # (ray casting on the GPU)
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    ctx.set_track_mode(chain=chain)
    d = pts
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    T = np.zeros((count, 12), np.float32)
    for s in range(count - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    return ctx, d, pts.cpu().numpy(), offs, poses


def _numpy_map(scvod, ctx, x, offs, poses, leaf, use_dyn=True, ground=True, rejected=True):
    """definition of the map: per static point the cell key and the packed in-cell offset, per cell the smallest"""
    keys, vals = [], []
    inv = np.float32(1.0) / np.float32(leaf)
    for s in range(len(offs) - 1):
        r = ctx.batch_fetch(s)
        t = ctx.batch_fetch_track(s)
        keep = t["pt_dyn"] != 1 if use_dyn else np.ones(r["n_apri"], bool)
        src = [r["apri_src"][keep]]
        if ground:
            src.append(r["ground_idx"])
        if rejected:
            src.append(r["rejected_src"])
        p = x[offs[s]:offs[s + 1]][np.concatenate(src)]
        T = scvod.pose_matrix(poses[s])
        w = [((T[4 * i] * p[:, 0] + T[4 * i + 1] * p[:, 1]) + T[4 * i + 2] * p[:, 2]) + T[4 * i + 3] for i in range(3)]
        f = [c * inv for c in w]
        c = [np.floor(v) for v in f]
        u = [(ci.astype(np.int64) + (1 << 20)).astype(np.uint64) for ci in c]
        key = (u[0] << np.uint64(42)) | (u[1] << np.uint64(21)) | u[2]
        q = [np.clip(((fi - ci) * np.float32(65536.0)).astype(np.int64), 0, 65535).astype(np.uint64) for fi, ci in zip(f, c)]
        qi = np.clip(p[:, 3] * np.float32(256.0), 0, 65535).astype(np.int64).astype(np.uint64)
        keys.append(key)
        vals.append((q[0] << np.uint64(48)) | (q[1] << np.uint64(32)) | (q[2] << np.uint64(16)) | qi)
    key, val = np.concatenate(keys), np.concatenate(vals)
    order = np.lexsort((val, key))
    key, val = key[order], val[order]
    first = np.ones(len(key), bool)
    first[1:] = key[1:] != key[:-1]
    return key[first], val[first]


def _sorted_records(m):
    rec = m.export().cpu().numpy().view(np.uint64)
    o = np.argsort(rec[:, 0])
    return rec[o, 0], rec[o, 1]


@pytest.mark.parametrize("kind,preset", [("K64", "semantickitti"), ("PARK", "parkinglot")])
def test_map_matches_its_definition(scvod, kind, preset):
    P = scvod.make_params(preset)
    count = 5
    ctx, d, x, offs, poses = _tracked_batch(scvod, P, kind, 5, 640, count)
    m = scvod.StaticMap(1 << 21, leaf=0.2)
    m.accumulate(ctx, poses)
    k, v = _sorted_records(m)
    ek, ev = _numpy_map(scvod, ctx, x, offs, poses, 0.2)
    assert np.array_equal(k, ek) and np.array_equal(v, ev)
    assert m.count() == len(ek)
    # the points handed out sit inside their cell, in the record order
    xyzi, rec = m.points()
    xyzi, rec = xyzi.cpu().numpy(), rec.cpu().numpy().view(np.uint64)
    cx = (rec[:, 0] >> np.uint64(42)).astype(np.int64) - (1 << 20)
    assert (np.floor(xyzi[:, 0] / np.float32(0.2) + 1e-3) >= cx - 1).all() and (np.abs(xyzi[:, 0] - (cx + 0.5) * 0.2) <= 0.1001).all()
    # accumulating the same batch again changes nothing (idempotent union); the raw map is a superset; flags drop lists
    m.accumulate(ctx, poses)
    k2, v2 = _sorted_records(m)
    assert np.array_equal(k, k2) and np.array_equal(v, v2)
    raw = scvod.StaticMap(1 << 21, leaf=0.2)
    raw.accumulate(ctx, poses, flags=scvod.MAP_IGNORE_DYNAMIC)
    rk, _ = _sorted_records(raw)
    assert len(rk) >= len(k) and np.isin(k, rk).all()
    raw.clear()
    raw.accumulate(ctx, poses, flags=scvod.MAP_NO_GROUND | scvod.MAP_NO_REJECTED)
    nk, nv = _sorted_records(raw)
    ek2, ev2 = _numpy_map(scvod, ctx, x, offs, poses, 0.2, ground=False, rejected=False)
    assert np.array_equal(nk, ek2) and np.array_equal(nv, ev2)
    raw.close()
    m.close()
    ctx.close()


def test_shards_merge_into_the_single_shard_map(scvod):
    """two shards (the second one tracked on its own, its first scan exported to the first as the boundary table)
    accumulate their scans into their own maps; merging the exported record lists reproduces the unsplit map bit for bit"""
    import torch
    P = scvod.make_params("semantickitti")
    count, cut = 6, 3
    ctx, d, x, offs, poses = _tracked_batch(scvod, P, "K64", 5, 2000, count, chain=False)  # (a cut ends a chain)
    whole = scvod.StaticMap(1 << 21)
    whole.accumulate(ctx, poses)
    wk, wv = _sorted_records(whole)
    ctx.close()
    T = np.zeros((count, 12), np.float32)
    oa, ob = np.asarray(offs[:cut + 1], np.int32), (np.asarray(offs[cut:], np.int64) - offs[cut]).astype(np.int32)
    ca = scvod.Ctx(P, max_points_total=int(oa[-1]) + 64, max_scans=cut)
    cb = scvod.Ctx(P, max_points_total=int(ob[-1]) + 64, max_scans=count - cut)
    for s in range(count - 1):
        T[s] = ca.pose_delta(poses[s], poses[s + 1])
    da, db = d[:offs[cut]].contiguous(), d[offs[cut]:].contiguous()
    for c, dd, oo in ((ca, da, oa), (cb, db, ob)):
        c.set_track_mode(chain=False)
        c.batch_process(dd, oo)
        c.batch_cluster()
        c.batch_cluster_types()
    cb.batch_track_tables()       # the order of a sharded step: tables -> export -> exchange -> track
    msg = torch.zeros((1 << 16, 4), dtype=torch.int32, device="cuda")
    cb.batch_export_table(0, msg)
    cb.batch_track(T[cut:])
    ca.batch_track(T[:cut], next_scan=np.array([1, 2, -2], np.int32), ext_tables=[msg])
    ma, mb = scvod.StaticMap(1 << 20), scvod.StaticMap(1 << 20)
    ma.accumulate(ca, poses[:cut])
    mb.accumulate(cb, poses[cut:])
    ra, rb = ma.export(), mb.export()
    # what the RCCL all_gather moves: equally sized, padded lists (key ~0 = padding)
    cap = max(ra.shape[0], rb.shape[0]) + 7
    pad = torch.full((2, cap, 2), -1, dtype=torch.int64, device="cuda")
    pad[0, :ra.shape[0]] = ra
    pad[1, :rb.shape[0]] = rb
    merged = scvod.StaticMap(1 << 21)
    merged.merge(pad.reshape(-1, 2))
    mk, mv = _sorted_records(merged)
    assert np.array_equal(mk, wk) and np.array_equal(mv, wv)
    # the reduce-scatter form: records grouped by owner; the groups partition the map and every shard computes the same owner
    ga, na = ma.export_parts(3)
    gb, nb = mb.export_parts(3)
    assert sum(na) == ra.shape[0] and sum(nb) == rb.shape[0]
    owner = {}
    for g, c in ((ga, na), (gb, nb)):
        k = g.cpu().numpy().view(np.uint64)[:, 0]
        o = np.repeat(np.arange(3), c)
        for kk, oo in zip(k.tolist(), o.tolist()):
            assert owner.setdefault(kk, oo) == oo
    parts = [scvod.StaticMap(1 << 20) for _ in range(3)]
    oa, ob2 = np.concatenate([[0], np.cumsum(na)]), np.concatenate([[0], np.cumsum(nb)])
    for p in range(3):
        parts[p].merge(ga[oa[p]:oa[p + 1]])
        parts[p].merge(gb[ob2[p]:ob2[p + 1]])
    allk = np.concatenate([_sorted_records(q)[0] for q in parts])
    allv = np.concatenate([_sorted_records(q)[1] for q in parts])
    o = np.argsort(allk)
    assert np.array_equal(allk[o], wk) and np.array_equal(allv[o], wv)
    for q in parts:
        q.close()
    # a map that is too small reports it instead of silently dropping cells
    tiny = scvod.StaticMap(1024)
    tiny.merge(ra)
    with pytest.raises(scvod.ScvodError):
        tiny.count()
    for o in (ma, mb, merged, whole, tiny):
        o.close()
    ca.close()
    cb.close()


def test_map_needs_a_tracked_batch(scvod):
    import torch
    P = scvod.make_params("semantickitti")
    ctx = scvod.Ctx(P, max_points_total=4096, max_scans=1)
    m = scvod.StaticMap(4096)
    pose = np.zeros((1, 6), np.float32)
    import ctypes as C
    assert m.lib.scvod_batch_map_accumulate(ctx.h, m.h, pose.ctypes.data_as(C.c_void_p), 0, None) == -5
    d = torch.zeros((100, 4), device="cuda")
    ctx.batch_process(d, [0, 100])
    assert m.lib.scvod_batch_map_accumulate(ctx.h, m.h, pose.ctypes.data_as(C.c_void_p), 0, None) == -5   # no tracking result
    assert m.lib.scvod_batch_map_accumulate(ctx.h, m.h, pose.ctypes.data_as(C.c_void_p), 4, None) == 0    # raw map is fine
    assert m.count() == 0  # 100 points at the origin: r <= 2.7 m, dropped by Patchwork, in neither cloud
    m.close()
    ctx.close()


def test_map_in_two_parts_equals_the_map_in_one(scvod):
    """SCVOD_MAP_PART_UNTRACKED (everything but the members of car clusters: final once the box rules ran, accumulated on a
    second stream while the batch is tracked) + SCVOD_MAP_PART_TRACKED (the car points tracking left static) must give the
    records of one unflagged call, bit for bit; the untracked part must not need a tracking result."""
    import torch
    P = scvod.make_params("semantickitti")
    count = 6
    import synth
    pts, offs, poses, _ = synth.make_batch(5, 2100, count, "K64")
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    d = pts.cuda()
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    two = scvod.StaticMap(1 << 21)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    two.accumulate(ctx, poses, flags=8, stream=side.cuda_stream)  # before any tracking call
    T = np.zeros((count, 12), np.float32)
    for s in range(count - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    torch.cuda.current_stream().wait_stream(side)
    n1 = two.count()
    two.accumulate(ctx, poses, flags=16)
    one = scvod.StaticMap(1 << 21)
    one.accumulate(ctx, poses)
    k1, v1 = _sorted_records(one)
    k2, v2 = _sorted_records(two)
    assert 0 < n1 < len(k1)
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
    n_dyn = sum(ctx.batch_fetch_track(s)["n_dynamic_points"] for s in range(count))
    assert n_dyn > 0
    raw = scvod.StaticMap(1 << 21)
    raw.accumulate(ctx, poses, flags=4)
    assert raw.count() > len(k1)  # the dynamic points opened cells of their own
    import ctypes as C
    bad = scvod.StaticMap(1 << 16)
    p = np.ascontiguousarray(poses, np.float32)
    assert bad.lib.scvod_batch_map_accumulate(ctx.h, bad.h, p.ctypes.data_as(C.c_void_p), 8 | 16, None) == -1
    for m in (one, two, raw, bad):
        m.close()
    ctx.close()


@pytest.mark.parametrize("kind,preset,skip", [("K64", "semantickitti", 5), ("PARK", "parkinglot", 1)])
def test_map_cells_equal_the_cells_of_the_reference_accumulation(scvod, oracle, kind, preset, skip):
    """The reference's map is `*instance_map += *rgb_ptr` over the clusters that are not dynamic (SSC::saveSegCloud mode 3,
    ssc.cpp:477-554) plus the ground clouds and the range / FOV rejects of the evaluation block (ssc.cpp:1460-1480), every scan
    moved to the world by its pose: a CONCATENATION of clouds.  The device keeps one representative per occupied 0.2 m cell (its
    own design, checked above against its definition); what must agree with the reference's semantics is WHICH cells are
    occupied.  Here the clouds come from the ORACLE alone (oracle_time_sequence: Patchwork, binning, clustering, box rules,
    the sequential tracking chain -> per input point static / dynamic / dropped), not from anything the device computed."""
    import synth
    import torch
    P = scvod.make_params(preset)
    count = 12
    scans = [synth.make_scan(5, 200 + k * skip, kind, device="cuda") for k in range(count)]
    x = np.concatenate([sc[0].cpu().numpy() for sc in scans])
    offs = np.concatenate([[0], np.cumsum([len(sc[0]) for sc in scans])]).astype(np.int32)
    poses = np.asarray([sc[2] for sc in scans], np.float32)
    _, lab, _ = oracle.time_sequence(P, x, offs, poses)   # 0 static, 1 dynamic, 2 in no cluster (kept), 3 dropped by Patchwork
    keep = (lab != 1) & (lab != 3)
    cells = []
    inv = np.float32(1.0) / np.float32(0.2)
    for s in range(count):
        p = x[offs[s]:offs[s + 1]][keep[offs[s]:offs[s + 1]]]
        T = scvod.pose_matrix(poses[s])
        w = [((T[4 * i] * p[:, 0] + T[4 * i + 1] * p[:, 1]) + T[4 * i + 2] * p[:, 2]) + T[4 * i + 3] for i in range(3)]
        u = [(np.floor(c * inv).astype(np.int64) + (1 << 20)).astype(np.uint64) for c in w]
        cells.append((u[0] << np.uint64(42)) | (u[1] << np.uint64(21)) | u[2])
    want = np.unique(np.concatenate(cells))
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    d = torch.from_numpy(x).cuda()
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    T = np.zeros((count, 12), np.float32)
    for s in range(count - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    m = scvod.StaticMap(1 << 21, leaf=0.2)
    m.accumulate(ctx, poses)
    k, _ = _sorted_records(m)
    assert int((lab == 1).sum()) > 0
    assert np.array_equal(k, want)
    m.close()
    ctx.close()
