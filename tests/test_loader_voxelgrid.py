"""label filter + pcl::VoxelGrid (loader step in front of the hot path, ssc.cpp:1063-1076, 1103-1106).
CPU part: hand-derived known answers for the oracle's PCL 1.8.1 restatement.  GPU part: the HIP path through the C-ABI,
bit-exact against the oracle (canonical order inside a cell), per scan and as a device-resident batch."""
import numpy as np
import pytest


def _grid_ref(x, leaf):
    """independent numpy statement of the cell index (float32 arithmetic like PCL)"""
    inv = (np.float32(1.0) / np.asarray(leaf, np.float32)).astype(np.float32)
    mn, mx = x[:, :3].min(0), x[:, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    div = np.floor(mx * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(x[:, :3] * inv).astype(np.float32) - min_b.astype(np.float32)).astype(np.int64)
    return ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]


def test_oracle_voxelgrid_known_answers(oracle):
    leaf = (0.5, 0.5, 0.5)
    # three points in one cell, one alone, one on a cell edge (floor puts 1.0 into the next cell)
    x = np.array([[0.10, 0.10, 0.10, 10], [0.20, 0.30, 0.40, 20], [0.45, 0.05, 0.25, 60],
                  [2.10, 0.10, 0.10, 5], [1.00, 0.20, 0.20, 7]], np.float32)
    out, rc = oracle.voxelgrid(x, leaf)
    assert rc == 0 and out.shape == (3, 4)
    c0 = (x[0] + x[1] + x[2]) / np.float32(3)           # fp32 running sums in input order, then / 3
    assert np.array_equal(out[0], c0)
    assert np.array_equal(out[1], x[4]) and np.array_equal(out[2], x[3])   # ascending cell index: x = 1.0 before x = 2.1
    # label filter + intensity scaling: labels 0 / 1 (low 16 bits) are dropped, the rest scaled
    lab = np.array([0, 40, 0x00010001, 50, 0x00050030], np.uint32)
    out, rc = oracle.voxelgrid(x, leaf, labels=lab, max_intensity=255.0)
    keep = x[[1, 3, 4]].copy()
    keep[:, 3] *= np.float32(255.0)
    order = np.argsort(_grid_ref(keep, leaf), kind="stable")
    assert np.array_equal(out, keep[order])
    # empty after filtering
    out, rc = oracle.voxelgrid(x, leaf, labels=np.zeros(5, np.uint32))
    assert out.shape[0] == 0
    # PCL's "leaf size is too small" branch returns the input cloud unchanged
    far = np.array([[-4000, -4000, -40, 1], [4000, 4000, 40, 2], [0, 0, 0, 3]], np.float32)
    out, rc = oracle.voxelgrid(far, (0.08, 0.08, 0.08))
    assert rc == 1 and np.array_equal(out, far)


def test_oracle_voxelgrid_cell_assignment_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-30, 30, (20000, 2)), rng.uniform(-3, 2, (20000, 1)), rng.uniform(0, 1, (20000, 1))], 1).astype(np.float32)
    leaf = (0.08, 0.08, 0.08)
    out, rc = oracle.voxelgrid(x, leaf)
    idx = _grid_ref(x, leaf)
    assert rc == 0 and out.shape[0] == len(np.unique(idx))
    # single-point cells come back bit-identical, in ascending cell order
    u, first, cnt = np.unique(idx, return_index=True, return_counts=True)
    single = cnt == 1
    assert np.array_equal(out[single], x[first[single]])
    # canonical vs std::sort order inside a cell: only sums of >= 3 points may differ, and only in the last bits
    out0, _ = oracle.voxelgrid(x, leaf, sort_mode=0)
    assert out0.shape == out.shape and np.array_equal(out0[cnt <= 2], out[cnt <= 2])
    assert np.allclose(out0, out, rtol=0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,leaf", [("K64", (0.08, 0.08, 0.08)), ("PARK", (0.1, 0.1, 0.1)), ("K64", (0.5, 0.25, 1.0))])
def test_voxelgrid_scan_parity(scvod, oracle, kind, leaf):
    import synth
    pts, _, _ = synth.make_scan(5, 123, kind)
    x = pts.numpy()
    ctx = scvod.Ctx(scvod.make_params("semantickitti"), max_points_total=x.shape[0] + 64, max_scans=1)
    got = ctx.voxelgrid(x, leaf)
    ref, rc = oracle.voxelgrid(x, leaf)
    assert rc == 0 and got.shape == ref.shape and got.shape[0] < x.shape[0]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # with the loader's label filter and intensity scaling
    rng = np.random.default_rng(8)
    lab = rng.choice(np.array([0, 1, 40, 48, 50, 0x00070000 | 252, 0x00010000], np.uint32), x.shape[0])
    got = ctx.voxelgrid(x, leaf, labels=lab, max_intensity=255.0)
    ref, rc = oracle.voxelgrid(x, leaf, labels=lab, max_intensity=255.0)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and got.shape[0] > 0
    # everything filtered, and the empty scan
    assert ctx.voxelgrid(x, leaf, labels=np.ones(x.shape[0], np.uint32)).shape[0] == 0
    assert ctx.voxelgrid(np.zeros((0, 4), np.float32), leaf).shape[0] == 0
    # PCL's overflow branch: output = (filtered) input
    far = np.concatenate([x[:1000], np.array([[-4000, -4000, -40, 1], [4000, 4000, 40, 2]], np.float32)])
    got = ctx.voxelgrid(far, (0.08, 0.08, 0.08))
    assert np.array_equal(got, far)
    ctx.close()


@pytest.mark.gpu
def test_voxelgrid_batch_feeds_the_hot_path(scvod, oracle):
    """device-resident loader: filter + downsample a batch, then run the hot path on the result without leaving HBM"""
    import torch
    import synth
    count = 4
    pts, offs, _, _ = synth.make_batch(5, 900, count, "K64")
    x = pts.numpy()
    rng = np.random.default_rng(5)
    lab = rng.choice(np.array([0, 1, 40, 44, 48, 50, 70, 252], np.uint32), x.shape[0], p=[.03, .02, .3, .1, .2, .2, .1, .05])
    P = scvod.make_params("semantickitti")
    ctx = scvod.Ctx(P, max_points_total=x.shape[0] + 64, max_scans=count)
    d_in, d_lab = pts.cuda(), torch.from_numpy(lab.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    out_off = ctx.batch_voxelgrid(d_in, offs, d_out, d_labels=d_lab, max_intensity=255.0)
    got = d_out.cpu().numpy()
    for s in range(count):
        ref, _ = oracle.voxelgrid(x[offs[s]:offs[s + 1]], labels=lab[offs[s]:offs[s + 1]], max_intensity=255.0)
        g = got[out_off[s]:out_off[s + 1]]
        assert g.shape == ref.shape and np.array_equal(g.view(np.uint32), ref.view(np.uint32)), f"scan {s}"
    assert out_off[-1] < x.shape[0]
    # the downsampled batch is a valid input of the hot path (still resident)
    ctx.batch_process(d_out, out_off)
    cnt = ctx.batch_counts()
    assert (cnt[:, 0] == np.diff(out_off)).all() and (cnt[:, 6] > 0).all()
    r = ctx.batch_fetch(1)
    o = oracle.patchwork(P, got[out_off[1]:out_off[2]], 1)
    assert np.array_equal(r["ground_idx"], o["ground_idx"])
    ctx.close()


@pytest.mark.gpu
def test_voxelgrid_error_conventions(scvod):
    """status codes, never a silent truncation: output buffer too small, bad leaf, too many scans; an earlier batch
    result is invalidated because the arena is reused"""
    import ctypes as C
    import torch
    import synth
    pts, offs, _, _ = synth.make_batch(3, 10, 2, "PARK")
    P = scvod.make_params("parkinglot")
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=2)
    d_in = pts.cuda()
    full = torch.empty_like(d_in)
    out_off = ctx.batch_voxelgrid(d_in, offs, full)
    need = int(out_off[-1])
    small = torch.empty((need - 5, 4), device="cuda")
    with pytest.raises(scvod.ScvodError):
        ctx.batch_voxelgrid(d_in, offs, small)
    assert b"output buffer too small" in ctx.lib.scvod_last_error(ctx.h)
    with pytest.raises(scvod.ScvodError):
        ctx.batch_voxelgrid(d_in, offs, full, leaf=(0.08, 0.0, 0.08))
    three = np.array([0, 10, 20, 30], np.int32)
    with pytest.raises(scvod.ScvodError):
        ctx.batch_voxelgrid(d_in, three, full)                     # 3 scans > max_scans 2
    # a processed batch is invalidated by a VoxelGrid run on the same ctx
    ctx.batch_process(d_in, offs)
    assert ctx.batch_counts()[0, 0] == offs[1]
    ctx.batch_voxelgrid(d_in, offs, full)
    r = scvod.ScanResult()
    assert ctx.lib.scvod_batch_fetch(ctx.h, 0, C.byref(r)) == -5    # SCVOD_ERR_STATE
    ctx.close()


@pytest.mark.gpu
def test_voxelgrid_degenerate_clouds(scvod, oracle):
    """everything in ONE cell (a single key bucket far beyond the LDS tiers: global-memory sort, one 30 000-term fp32
    sum), a cloud on a single line, and two far-apart clusters (almost every histogram bin empty)"""
    rng = np.random.default_rng(9)
    ctx = scvod.Ctx(scvod.make_params("semantickitti"), max_points_total=70000, max_scans=1)
    one = np.tile(np.array([[1.01, 2.02, -0.53, 7.0]], np.float32), (30000, 1))
    one[:, :3] += rng.uniform(0, 0.01, (30000, 3)).astype(np.float32)
    one = np.concatenate([one, np.array([[5, 5, 0, 1], [-5, 5, 0, 2]], np.float32)])
    line = np.zeros((5000, 4), np.float32)
    line[:, 0] = np.linspace(-30, 30, 5000)
    line[:, 3] = np.arange(5000)
    far = np.concatenate([rng.normal([-70, -70, -2, 50], 0.3, (20000, 4)), rng.normal([70, 70, 2, 50], 0.3, (20000, 4))]).astype(np.float32)
    for name, x in (("one cell", one), ("line", line), ("two clusters", far)):
        got = ctx.voxelgrid(x, (0.08, 0.08, 0.08))
        ref, rc = oracle.voxelgrid(x, (0.08, 0.08, 0.08))
        assert rc == 0 and got.shape == ref.shape, name
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), name
    ctx.close()
