"""Known-answer cases that pin the oracle, derived by hand from the reference's formulas
(SURVEY.md 8c "golden vectors to commit").  Not gpu."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_grid_constants(oracle, scvod):
    # src/ssc.cpp:36-39 with config/semantickitti.yaml, config/parkinglot.yaml and the 2x finer grid
    assert oracle.grid_dims(scvod.make_params("semantickitti")) == (72, 300, 60, 1296000)
    assert oracle.grid_dims(scvod.make_params("parkinglot")) == (98, 300, 45, 1323000)
    assert oracle.grid_dims(scvod.make_params("os128_fine")) == (143, 600, 120, 10296000)
    # nh.param<> defaults (utility.h:283-292): (50-0)/0.2 = 250, 360/1.2 = 300, 90/2 = 45
    assert oracle.grid_dims(oracle.params_default()) == (250, 300, 45, 3375000)


def _pt(x, y, z, i=7.0):
    return np.array([[x, y, z, i]], np.float32)


def test_bin_edges(oracle, scvod):
    P = scvod.make_params("semantickitti")  # min_dis 1.5, max_dis 30, res 0.4 / 1.2 / 2.0, azimuth -40..80
    R, S, A, _ = oracle.grid_dims(P)
    a = oracle.bin(P, _pt(1.5, 0.0, 0.0), True)["apri"][0]   # dis == min_dis -> range_idx -1; y == +0 -> angle 0 -> sector -1
    assert (a["range_idx"], a["sector_idx"]) == (-1, -1)
    assert a["azimuth_idx"] == 19                                # ceil((0+40)/2)-1
    assert a["voxel_idx"] == 19 * R * S + (-1) * S + (-1)
    a = oracle.bin(P, _pt(30.0, 1e-3, 0.0), True)["apri"][0]  # dis ~ max_dis -> R-1
    assert a["range_idx"] == R - 1 and a["sector_idx"] == 0
    a = oracle.bin(P, _pt(10.0, -1e-4, 0.0), True)["apri"][0]  # just below 360 deg -> S-1
    assert a["sector_idx"] == S - 1
    assert len(oracle.bin(P, _pt(30.01, 0.5, 0.0), True)["apri"]) == 0    # range reject
    assert len(oracle.bin(P, _pt(1.0, 1.0, 0.0), True)["apri"]) == 0      # dis 1.41 < 1.5
    assert len(oracle.bin(P, _pt(5.0, 0.0, -5.0), True)["apri"]) == 0     # azimuth -45 < -40
    assert len(oracle.bin(P, _pt(5.0, 0.0, -5.0), False)["apri"]) == 1    # tracking re-bin keeps it, unclamped
    a = oracle.bin(P, _pt(0.0, 0.0, 1.0), False)["apri"][0]              # x == y == 0 -> angle 0
    assert a["angle"] == 0.0 and a["azimuth"] == 90.0
    a = oracle.bin(P, _pt(-3.0, 0.0, 0.0), True)["apri"][0]              # y == +0, x < 0 -> 180 deg
    assert a["angle"] == 180.0 and a["sector_idx"] == 149               # ceil(180/1.2) - 1


def test_voxel_mean_variance(oracle, scvod):
    P = scvod.make_params("semantickitti")
    pts = np.array([[5.0, 5.0, 0.1, 10.0], [5.01, 5.0, 0.1, 20.0], [5.0, 5.01, 0.1, 60.0], [-9.0, 2.0, 0.5, 100.0]], np.float32)
    b = oracle.bin(P, pts, True)
    v = oracle.voxelize(P, b["apri"])
    assert len(v["vox_key"]) == 2
    k0 = b["apri"]["voxel_idx"][0]
    i = int(np.nonzero(v["vox_key"] == k0)[0][0])
    assert list(v["vox_pts"][v["vox_pt_begin"][i]:v["vox_pt_begin"][i + 1]]) == [0, 1, 2]
    assert v["vox_av"][i] == np.float32(30.0)
    # population variance ((20^2 + 10^2 + 30^2) / 3)
    assert abs(v["vox_cov"][i] - 1400.0 / 3.0) < 1e-4
    j = 1 - i
    assert v["vox_av"][j] == np.float32(100.0) and v["vox_cov"][j] == 0.0
    # "center" is the lower corner because (2i+1)/2 is integer division (ssc.cpp:271-273)
    a = b["apri"][3]
    rc = a["range_idx"] * np.float32(0.4) + np.float32(1.5)
    assert abs(np.hypot(v["center"][j][0], v["center"][j][1]) - rc) < 1e-4
    assert v["center"][j][3] == np.float32(a["voxel_idx"])
    assert list(v["idx3"][j]) == [a["range_idx"], a["sector_idx"], a["azimuth_idx"]]


def _plane_patch(rng, n, r0, r1, t0, t1, fn):
    r = rng.uniform(r0, r1, n)
    t = rng.uniform(t0, t1, n)
    x, y = r * np.cos(t), r * np.sin(t)
    return np.stack([x, y, fn(x, y), np.full(n, 50.0)], 1).astype(np.float32)


def test_patchwork_zone_edges_and_planes(oracle, scvod):
    P = scvod.make_params("semantickitti")  # sensor_height 1.73
    rng = np.random.default_rng(5)
    # radii exactly at the model boundaries (patchwork.h:83-85, 436): (2.7, 80] is binned
    # r = 2.6999 (dropped), 2.7001 (binned), exactly 80.0 (binned: r <= max_range), 80.001 (dropped)
    edge = np.array([[2.6999, 0.0, -1.73, 1], [2.7001, 0.001, -1.73, 1], [80.0, 0.0, -1.73, 1], [80.001, 0.0, -1.73, 1]], np.float32)
    flat = _plane_patch(rng, 400, 3.0, 7.0, 0.02, 0.37, lambda x, y: np.full_like(x, -1.73))
    cloud = np.concatenate([edge, flat])
    o = oracle.patchwork(P, cloud, 1)
    pl = o["planes"]
    assert pl.shape[0] == 504
    assert pl["n_pts"].sum() == 402                         # 400 + the two in-range edge points
    assert o["cls"][0] == 2 and o["cls"][3] == 2            # out of (2.7, 80]: silently dropped
    assert o["cls"][2] == 2 and pl["n_pts"][32 + 128 + 216 + 3 * 32] == 1   # zone 3 last ring sector 0, size <= 10
    assert pl["n_pts"][0] == 401                            # zone 0, ring 0, sector 0 (incl. the r = 2.7001 point)
    assert pl["status"][0] == 1
    assert abs(abs(pl["normal"][0][2]) - 1.0) < 1e-6        # perfect plane -> normal (0,0,+-1)
    # exact plane, but PCL's single-pass fp32 covariance cancels catastrophically: ~1e-5, not 0
    assert pl["sv"][0][2] / pl["sv"][0].sum() < 5e-5
    assert (o["cls"][4:] == 0).all()                        # every point of the flat patch is ground
    # erased prefix: z < -1.8 * 1.73 = -3.114 (patchwork.h:304)
    low = flat.copy()
    low[:5, 2] = -3.2
    o2 = oracle.patchwork(P, low, 1)
    assert (o2["cls"][:5] == 2).all() and o2["planes"]["n_pts"][0] == 395
    # uprightness gate |n_z| >= 0.707 (patchwork.h:346): a 35-degree slope (n_z = 0.82) stays ground,
    # a 55-degree slope (n_z = 0.57) sends the whole patch to non-ground
    for deg, ok in ((35.0, True), (55.0, False)):
        s = np.tan(np.radians(deg))
        tilt = _plane_patch(rng, 600, 26.5, 27.5, 0.01, 0.1, lambda x, y: -1.0 + s * (x - 27.0))
        o3 = oracle.patchwork(P, tilt, 1)
        st = o3["planes"]["status"]
        live = st > 0
        assert live.any()
        assert ((st[live] == 1).all()) == ok, (deg, st[live], o3["planes"]["normal"][live])
        if not ok:
            assert (o3["cls"] != 0).all()                   # nothing of a rejected patch is ground


def test_patchwork_tie_order_only_affects_ties(oracle, scvod):
    """std::sort (reference) and the canonical (z, idx) order agree exactly when no two points of a
    patch share z; with ties the SETS still agree up to the fp32 summation order."""
    import synth
    P = scvod.make_params("semantickitti")
    pts, _, _ = synth.make_scan(5, 3, "PARK")
    x = pts.numpy().copy()
    # make z unique by a deterministic sub-ulp-free perturbation: sort-rank based
    order = np.argsort(x[:, 2], kind="stable")
    z = x[order, 2].copy()
    for i in range(1, len(z)):
        if z[i] <= z[i - 1]:
            z[i] = np.nextafter(z[i - 1], np.float32(np.inf))
    x[order, 2] = z
    a, b = oracle.patchwork(P, x, 0), oracle.patchwork(P, x, 1)
    for k in ("cls", "ground_idx", "nonground_idx"):
        assert np.array_equal(a[k], b[k])


def test_track_probe_ratio_and_pose_delta(oracle, scvod):
    P = scvod.make_params("semantickitti")
    # identical poses -> identity transform
    T = oracle.pose_delta([3, 1, 0.2, 0.01, -0.02, 0.3], [3, 1, 0.2, 0.01, -0.02, 0.3])
    assert np.allclose(T.reshape(3, 4), np.hstack([np.eye(3), np.zeros((3, 1))]), atol=2e-6)
    # pure forward motion of 1 m: a static point moves 1 m backwards in the next frame
    T = oracle.pose_delta([0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0])
    assert np.allclose(T.reshape(3, 4)[:, 3], [-1, 0, 0])
    nxt = np.array([[9.0, 2.0, 0.0, 1], [9.05, 2.0, 0.0, 1], [9.0, 2.3, 0.3, 1], [15.0, -4.0, 0.5, 1]], np.float32)
    b = oracle.bin(P, nxt, True)
    v = oracle.voxelize(P, b["apri"])
    labels = np.arange(len(v["vox_key"]), dtype=np.int32) + 5
    pre = nxt.copy()
    pre[:, 0] += 1.0                                  # the same static points seen one metre earlier
    hit, uq, ub = oracle.track_probe(P, pre, [0, 3, 4], T, v["vox_key"], labels)
    assert (hit >= 0).all() and list(ub) == [0, len(np.unique(hit[:3])), len(np.unique(hit[:3])) + 1]
    labels2 = labels.copy()
    labels2[hit[3]] = -1                              # unlabeled voxel (refined away) is not a hit
    hit2, _, ub2 = oracle.track_probe(P, pre, [0, 3, 4], T, v["vox_key"], labels2)
    assert hit2[3] == -1 and ub2[2] - ub2[1] == 0
    # occupancy ratio = unique hit voxels / voxels owned (ssc.cpp:1336): here 2 of 2 or 3 of 3 >= occupancy_
    assert (ub[1] - ub[0]) / float(len(np.unique(b["apri"]["voxel_idx"][:3]))) >= P.occupancy


def test_metric_matches_reference_golden(oracle):
    """tests/golden/metric_golden.json was produced by the reference's own tool/analysis.py."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from metric_cases import make_case
    import metric
    gold = json.load(open(os.path.join(HERE, "golden", "metric_golden.json")))
    for g in gold:
        xyz, lab, exyz, elab = make_case(**g["case"])
        m = metric.preservation_rejection(xyz, lab, exyz, elab, oracle.nn_search, voxelsize=0.2)
        for k in ("num_gt_static", "num_gt_dynamic", "num_est_static", "num_est_dynamic", "num_preserved",
                  "num_static_preserved", "num_dynamic_preserved"):
            assert m[k] == g[k], (g["case"], k)
        assert abs(m["PR"] - g["PR"]) < 1e-9 and abs(m["RR"] - g["RR"]) < 1e-9 and abs(m["F1"] - g["F1"]) < 1e-9


def test_map_point_classes_of_the_evaluation_viewer(oracle):
    """evaluate.cpp:79-145 by hand: one point per branch, radii 0.15 / 0.10 with strict comparisons"""
    sys_path_hack = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dr-using-scv-od_amd", "pyshim")
    if sys_path_hack not in sys.path:
        sys.path.insert(0, sys_path_hack)
    import metric
    static = np.array([[0, 0, 0], [10, 0, 0]], np.float32)
    dynamic = np.array([[5, 0, 0], [20, 0, 0]], np.float32)
    orig = np.array([[0.10, 0, 0],     # predicted static, static within 0.15          -> TP
                     [5.05, 0, 0],     # predicted static, no static, dynamic within 0.10 -> FN (orange)
                     [5.12, 0, 0],     # predicted static, dynamic only within 0.15 (not 0.10) -> unmatched
                     [5.12, 0, 0],     # predicted dynamic, dynamic within 0.15        -> TN
                     [10.08, 0, 0],    # predicted dynamic, no dynamic, static within 0.10 -> FN (pink)
                     [10.12, 0, 0],    # predicted dynamic, static only within 0.15   -> unmatched
                     [50, 0, 0]], np.float32)
    pred_static = np.array([1, 1, 1, 0, 0, 0, 1], bool)
    got = metric.classify_map_points(orig, pred_static, static, dynamic, oracle.nn_search)
    assert got.tolist() == [metric.TP_STATIC, metric.FN_STATIC, metric.UNMATCHED, metric.TN_DYNAMIC, metric.FN_DYNAMIC,
                            metric.UNMATCHED, metric.UNMATCHED]
    assert (metric.classify_map_points(orig, pred_static, static, np.zeros((0, 3), np.float32), oracle.nn_search)[[1, 3]] == 0).all()
