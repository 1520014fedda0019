"""Size-independent properties of the HIP path at BASELINE.json's full scan sizes (where the oracle
would take too long to run on every scan): conservation, ordering, idempotence, checksums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _invariants(r, x, grid):
    R, S, A, bins = grid
    n = r["n_points"]
    assert n == x.shape[0] == r["n_ground"] + r["n_nonground"] + r["n_dropped"]
    assert r["n_apri"] + r["n_rejected"] == r["n_nonground"]
    cls = r["cls"]
    assert (cls[r["ground_idx"]] == 0).all() and (cls[r["nonground_idx"]] == 1).all()
    assert int((cls == 0).sum()) == r["n_ground"] and int((cls == 1).sum()) == r["n_nonground"]
    # index lists are permutations of disjoint subsets
    assert len(np.unique(r["ground_idx"])) == r["n_ground"] and len(np.unique(r["nonground_idx"])) == r["n_nonground"]
    # apri_vec is the non-ground cloud filtered IN ORDER
    pos = np.full(n, -1, np.int64)
    pos[r["nonground_idx"]] = np.arange(r["n_nonground"])
    pa, pr = pos[r["apri_src"]], pos[r["rejected_src"]]
    assert (pa >= 0).all() and (pr >= 0).all() and (np.diff(pa) > 0).all() and (np.diff(pr) > 0).all()
    assert np.array_equal(np.sort(np.concatenate([pa, pr])), np.arange(r["n_nonground"]))
    a = r["apri"]
    assert np.array_equal(a["x"].view(np.uint32), x[r["apri_src"], 0].view(np.uint32))
    assert np.array_equal(a["voxel_idx"], a["azimuth_idx"] * R * S + a["range_idx"] * S + a["sector_idx"])
    assert (a["voxel_idx"] < bins).all()
    # hash cloud: keys strictly ascending, CSR covers every apri point once, ptIdx ascending per voxel
    k, b, p = r["vox_key"], r["vox_pt_begin"], r["vox_pts"]
    assert (np.diff(k) > 0).all() and b[0] == 0 and b[-1] == r["n_apri"] and (np.diff(b) > 0).all()
    assert np.array_equal(np.sort(p), np.arange(r["n_apri"]))
    vox_of = np.repeat(np.arange(len(k)), np.diff(b))
    assert np.array_equal(a["voxel_idx"][p], k[vox_of])
    same = vox_of[1:] == vox_of[:-1]
    assert (np.diff(p)[same] > 0).all()
    # descriptor floats against a float64 recomputation (1e-4, north_star tolerance)
    inten = a["intensity"][p].astype(np.float64)
    cnt = np.diff(b)
    mean = np.add.reduceat(inten, b[:-1]) / cnt
    var = np.add.reduceat((inten - mean[vox_of]) ** 2, b[:-1]) / cnt
    assert np.allclose(r["vox_av"], mean, rtol=1e-4, atol=1e-3)
    assert np.allclose(r["vox_cov"], var, rtol=1e-3, atol=1e-2)
    # Patchwork planes: unit normals, singular values descending, populations add up
    pl = r["planes"]
    live = pl["status"] > 0
    assert np.allclose(np.linalg.norm(pl["normal"][live], axis=1), 1.0, atol=1e-4)
    assert (pl["sv"][live][:, 0] >= pl["sv"][live][:, 1]).all() and (pl["sv"][live][:, 1] >= pl["sv"][live][:, 2]).all()
    assert pl["n_pts"][live].sum() == r["n_ground"] + r["n_nonground"]
    assert (pl["n_ground"][pl["status"] == 1]).sum() == r["n_ground"]
    return (int(r["n_ground"]), int(r["n_apri"]), int(r["n_voxels"]), int(k.astype(np.int64).sum() % (1 << 31)),
            int(a["voxel_idx"].astype(np.int64).sum() % (1 << 31)))


@pytest.mark.parametrize("kind,preset,count", [("K64", "semantickitti", 48), ("OS128", "os128_fine", 6), ("PARK", "parkinglot", 24)])
def test_full_size_batches(scvod, kind, preset, count):
    import torch
    import synth
    P = scvod.make_params(preset)
    grid = scvod.grid_dims(P)
    pts, offs, poses, _ = synth.make_batch(5, 1000, count, kind, device="cuda")
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    ctx.batch_process(pts, offs)
    c1 = ctx.batch_counts().copy()
    x = pts.cpu().numpy()
    sums = []
    for s in (0, count // 2, count - 1):
        sums.append(_invariants(ctx.batch_fetch(s), x[offs[s]:offs[s + 1]], grid))
    # idempotence + order independence: the same scans in reverse batch order give the same per-scan results
    rev_pts = torch.cat([pts[offs[s]:offs[s + 1]] for s in reversed(range(count))]).contiguous()
    rev_offs = np.concatenate([[0], np.cumsum([offs[s + 1] - offs[s] for s in reversed(range(count))])]).astype(np.int32)
    ctx.batch_process(rev_pts, rev_offs)
    c2 = ctx.batch_counts()
    assert np.array_equal(c1, c2[::-1])
    xr = rev_pts.cpu().numpy()
    s = count - 1
    assert _invariants(ctx.batch_fetch(0), xr[rev_offs[0]:rev_offs[1]], grid) == sums[2]
    # scan-vs-next-scan differencing over the batch: per cluster the unique labelled voxels are bounded by its size and by
    # the successor's table, the pair counts add up to them, dynamic points are car points; a batch tracked against
    # ITSELF (identity transforms, next_scan[s] = s) finds every own voxel: a cluster can only come out dynamic when its
    # voxels carry another label (voxels shared through the -1 index aliasing with a cluster the refine erased)
    ctx.batch_process(pts, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    cnt = ctx.batch_counts()
    T = np.zeros((count, 12), np.float32)
    for s in range(count - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    seen = 0
    for s in (0, count // 2, count - 2):
        t = ctx.batch_fetch_track(s)
        assert (t["n_unique"] <= t["cluster_size"]).all() and (t["n_unique"] <= cnt[s + 1, 6]).all()
        sums = np.array([t["pair_count"][t["pair_begin"][k]:t["pair_begin"][k + 1]].sum() for k in range(t["n_clusters"])], np.int64)
        assert np.array_equal(sums, t["n_unique"])
        assert t["cluster_size"].sum() == t["n_car_points"] and int((t["pt_dyn"] == 1).sum()) == t["n_dynamic_points"]
        assert set(np.unique(t["cluster_state"])) <= {-1, 0, 1}
        seen += t["n_clusters"]
    assert seen > 0
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (count, 1))
    import ctypes as C
    nself = np.arange(count, dtype=np.int32)
    assert ctx.lib.scvod_batch_track(ctx.h, ident.ctypes.data_as(C.c_void_p), nself.ctypes.data_as(C.c_void_p), None, 0, None, 1) == -1, \
        "a scan cannot be its own successor in the sequential chain"
    ctx.set_track_mode(chain=False)
    ctx.batch_track(ident, next_scan=nself)
    for s in (0, count - 1):
        t = ctx.batch_fetch_track(s)
        assert t["n_dynamic_clusters"] <= max(1, t["n_clusters"] // 10) and (t["cluster_state"] == 0).sum() >= 0.8 * t["n_clusters"]
    ctx.close()


def test_self_probe_hits_every_voxel(scvod):
    import synth
    P = scvod.make_params("semantickitti")
    pts, _, _ = synth.make_scan(5, 77, "K64")
    x = pts.numpy()
    ctx = scvod.Ctx(P, max_points_total=x.shape[0] + 64, max_scans=1)
    r = ctx.process_scan(x)
    a = r["apri"]
    xyzi = np.stack([a["x"], a["y"], a["z"], a["intensity"]], 1)
    T = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
    hit, uq, ub = ctx.track_probe(xyzi, [0, len(a)], T, r["vox_key"], None)
    assert (hit >= 0).all()                       # identity transform: every point finds its own voxel
    assert np.array_equal(r["vox_key"][hit], a["voxel_idx"])
    assert np.array_equal(uq, np.arange(r["n_voxels"]))
    ctx.close()


@pytest.mark.gpu
def test_repeated_batches_are_bitwise_identical(scvod):
    """Scatters and unions use atomics (non-deterministic intermediate order), every exported array is canonical:
    five runs of the same 24-scan batch must agree bit for bit in everything a caller can fetch, including clusters,
    types and the tracking counts."""
    import hashlib
    import torch
    import synth
    P = scvod.make_params("semantickitti")
    count = 24
    pts, offs, poses, _ = synth.make_batch(5, 1500, count, "K64", device="cuda")
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    T = np.stack([ctx.pose_delta(poses[s], poses[s + 1]) for s in range(count - 1)] + [np.zeros(12, np.float32)])

    def digest():
        h = hashlib.sha256()
        ctx.batch_process(pts, offs)
        cnt = ctx.batch_counts().copy()
        h.update(cnt.tobytes())
        for s in (0, 7, count - 1):
            r = ctx.batch_fetch(s)
            for k in ("cls", "ground_idx", "nonground_idx", "planes", "apri", "apri_src", "rejected_src", "vox_key",
                      "vox_pt_begin", "vox_pts", "vox_av", "vox_cov"):
                h.update(np.ascontiguousarray(r[k]).tobytes())
        ctx.batch_cluster()
        ctx.batch_cluster_types()
        for s in range(count):
            h.update(ctx.batch_fetch_clusters(s, int(cnt[s, 4])).tobytes())
            h.update(ctx.batch_fetch_cluster_types(s, int(cnt[s, 4])).tobytes())
        ctx.batch_track(T)
        for s in (0, 7, count - 1):
            t = ctx.batch_fetch_track(s)
            for k in ("cluster_root", "cluster_size", "cluster_state", "n_unique", "pair_begin", "pair_label", "pair_count", "pt_dyn"):
                h.update(np.ascontiguousarray(t[k]).tobytes())
        return h.hexdigest()

    first = digest()
    for _ in range(4):
        assert digest() == first
    ctx.close()
