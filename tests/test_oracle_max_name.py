"""Frame::max_name as the reference stores and re-uses it, pinned on hand-built cases.  Not gpu.

ssc.cpp:354  `frame_ssc.max_name = cluster_name ++;`   -> max_name is the LAST USED running number K (post-increment)
ssc.cpp:1357 / :1401  `cluster_new.name = frame_next_.max_name ++;`  -> the first cluster SSC::tracking makes in a frame is K again
ssc.cpp:1372 / :1419  `cluster_set.insert(...)` is a no-op while a cluster K is alive: the new cluster is lost
ssc.cpp:329, :413-419 `mergeClusters(clusterIdxs, oc, nc)`: the VISITING point's cluster takes the NEIGHBOUR's name, so
                      whether K is still alive when the loop ends depends on the visiting order."""
import numpy as np


def _cell_point(P, r, s, a, jitter=0.0):
    """a point in the middle of grid cell (range r, sector s, azimuth a) of the curved-voxel grid (ssc.cpp:185-188)"""
    dis = P.min_dis + (r + 0.5) * P.range_res + jitter
    ang = np.deg2rad(P.min_angle + (s + 0.5) * P.sector_res)
    azi = np.deg2rad(P.min_azimuth + (a + 0.5) * P.azimuth_res)
    return [dis * np.cos(ang), dis * np.sin(ang), dis * np.tan(azi), 5.0]


def _apri(oracle, P, cells):
    pts = np.asarray([_cell_point(P, *c[:3], jitter=(c[3] if len(c) > 3 else 0.0)) for c in cells], np.float32)
    b = oracle.bin(P, pts, True)["apri"]
    assert len(b) == len(cells)
    for k, c in enumerate(cells):
        assert (b["range_idx"][k], b["sector_idx"][k], b["azimuth_idx"][k]) == tuple(c[:3])
    return b


def test_which_cluster_keeps_the_last_running_number(oracle, scvod):
    P = scvod.make_params("semantickitti")
    # two blobs far apart: point 0 opens cluster 5, point 1 opens cluster 6 = K: alive, its canonical name is point 1
    name, info = oracle.cluster_last_name(P, _apri(oracle, P, [(10, 50, 20), (40, 200, 20)]))
    assert name == 1 and info[0] == 6 and info[1] == 1
    # four cells in a row.  Point 0 (range 10) opens 5 and labels what it lists (the point in range 9); point 1 (range 7, not a
    # neighbour) opens 6 = K and labels the point in range 8; that point (label 6) then visits and meets the point in range 9
    # (label 5): mergeClusters(6, 5) -- the VISITOR's cluster takes the neighbour's name, K is renamed away.  No cluster carries K.
    a = _apri(oracle, P, [(10, 50, 20), (7, 50, 20), (8, 50, 20), (9, 50, 20)])
    cl, n_cl, mx = oracle.cluster(P, a)
    assert n_cl == 1 and mx == 6 and set(cl) == {5}
    name, info = oracle.cluster_last_name(P, a)
    assert name == -1 and info[4] == 1
    # mirrored (ranges 7, 10, 8, 9): the visitor in range 8 now carries 5 and meets 6: mergeClusters(5, 6), everything is K
    a = _apri(oracle, P, [(7, 50, 20), (10, 50, 20), (8, 50, 20), (9, 50, 20)])
    cl, n_cl, mx = oracle.cluster(P, a)
    assert n_cl == 1 and mx == 6 and set(cl) == {6}
    name, info = oracle.cluster_last_name(P, a)
    assert name == 0


def _three_frames(oracle, P):
    """frame 0: a car cluster over cells A and B.  frame 1: car cluster X in A, car cluster Y in B, a third cluster Z far
    away.  frame 2: nothing near A or B.  Identity poses.  tracking(0, 1) fuses X and Y (two labels hit, both car clusters
    covered at 1 / 1 >= occupancy: ssc.cpp:1396-1419); tracking(1, 2) would find the fused cluster without a successor."""
    A, B, Z, FAR = (20, 50, 20), (30, 50, 20), (40, 200, 20), (50, 120, 20)
    f0 = _apri(oracle, P, [A, A, B, B])
    f1 = _apri(oracle, P, [A, A, B, B, Z])
    f2 = _apri(oracle, P, [FAR])
    apri = np.concatenate([f0, f1, f2])
    offs = np.asarray([0, 4, 9, 10], np.int32)
    car, other = 2, 1
    names = np.asarray([0, 0, 0, 0, 0, 0, 2, 2, 4, 0], np.int32)
    types = np.asarray([car] * 4 + [car, car, car, car, other] + [other], np.int32)
    poses = np.zeros((3, 6), np.float32)
    return apri, offs, names, types, poses


def test_the_first_new_cluster_of_a_frame_reuses_max_name(oracle, scvod):
    P = scvod.make_params("semantickitti")
    apri, offs, names, types, poses = _three_frames(oracle, P)
    X = slice(4, 8)  # the points of X and Y in frame 1

    def run(collide_1):
        dyn, nd, st = oracle.sequence_tracking_literal(P, apri, offs, names, types, [-1, collide_1, -1], poses, chain=3)
        return dyn, nd, st

    # K was merged away in frame 1 (no cluster carries it): the fused cluster gets the free name, is walked by the next call,
    # finds nothing in frame 2 -> dynamic (ssc.cpp:1323-1326), its points are marked
    dyn, nd, st = run(-1)
    assert list(dyn[X]) == [1, 1, 1, 1] and nd == 1 and list(st[:2]) == [0, 0]
    fresh, fresh_nd = oracle.sequence_tracking(P, apri, offs, names, types, poses, chain=3)
    assert np.array_equal(dyn, fresh) and nd == fresh_nd
    # cluster Z (name 4) still carries K: the fused cluster is called 4, X and Y are erased (ssc.cpp:1411), their voxels
    # re-labelled 4 (:1417) and `insert` (:1419) does nothing -- the points of X and Y sit in no cluster any more, nothing
    # walks them, nothing marks them
    dyn, nd, st = run(4)
    assert list(dyn[X]) == [0, 0, 0, 0] and nd == 0 and list(st) == [0, 1, 2, 4]
    # X itself carries K: the erase frees the name before the insert, which succeeds -- same outcome as a fresh number
    dyn, nd, st = run(0)
    assert list(dyn[X]) == [1, 1, 1, 1] and nd == 1 and list(st[:2]) == [0, 0]
    assert (dyn[:4] == 0).all() and dyn[8] == 0 and dyn[9] == 0


def test_a_split_off_cluster_called_max_name_hands_its_voxels_to_that_cluster(oracle, scvod):
    """ssc.cpp:1351-1372 with the name of a live cluster: the hit voxel is re-labelled K (:1366), the source loses it (:1364),
    the insert (:1372) does nothing.  A later cluster of the same call that hits this voxel is then judged against cluster K's
    size and type, not against a one-voxel cluster of the source's type."""
    P = scvod.make_params("semantickitti")  # occupancy 0.4
    car, other = 2, 1
    A = (20, 50, 20)
    W = [A, (20, 52, 20), (20, 54, 20), (20, 56, 20), (20, 58, 20)]          # frame 1: a non-car cluster over five cells (apart: names given)
    K = [(40, 200, 20), (40, 202, 20), (40, 204, 20), (40, 206, 20)]          # frame 1: a car cluster over four cells
    f0 = _apri(oracle, P, [A, A + (0.05,)])                                   # frame 0: two car clusters, one point each, both in cell A
    f1 = _apri(oracle, P, W + K)
    f2 = _apri(oracle, P, [(50, 120, 20)])
    apri = np.concatenate([f0, f1, f2])
    offs = np.asarray([0, 2, 11, 12], np.int32)
    names = np.asarray([0, 1] + [0] * 5 + [5] * 4 + [0], np.int32)
    types = np.asarray([car, car] + [other] * 5 + [car] * 4 + [other], np.int32)
    poses = np.zeros((3, 6), np.float32)
    # fresh number: cluster 0 of frame 0 splits cell A off the five-voxel cluster (1 / 5 < 0.4, not a car); cluster 1 then hits a
    # one-voxel non-car cluster at 1 / 1: nothing is decided, its state stays -1
    dyn, nd = oracle.sequence_tracking(P, apri, offs, names, types, poses, chain=3)
    assert list(dyn[:2]) == [0, 0] and nd == 1  # (the one: car cluster 5 of frame 1, which finds nothing in frame 2)
    # the car cluster 5 carries K: cell A is re-labelled 5, and cluster 1 hits ONE of "its" four voxels: 1 / 4 < 0.4 against a car
    # cluster -> dynamic (ssc.cpp:1337-1349)
    dyn, nd, st = oracle.sequence_tracking_literal(P, apri, offs, names, types, [-1, 5, -1], poses, chain=3)
    assert list(dyn[:2]) == [0, 1] and list(st[:2]) == [1, 0]
    assert nd == 2
