"""Call sequences on ONE ctx that used to corrupt state: streaming ingest after a batch run and a longer
ingest after a shorter one (pinned staging ring freed under its users), the one-shot probe between two batch tracks
(cached transform upload), a batch of empty scans."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ingest(scvod, ctx, host, offs, chunk):
    lib = scvod.load_lib()
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)
    lib.scvod_sequence_ingest.restype = C.c_int
    lib.scvod_sequence_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, CB, C.c_void_p]
    seen = []

    def consumers(user, h, first, n, stream):
        if lib.scvod_batch_cluster(h, stream, 0) or lib.scvod_batch_cluster_types(h, stream, 0):
            return -3
        T = np.zeros((n, 12), np.float32)
        T[:, 0] = T[:, 5] = T[:, 10] = 1.0
        rc = lib.scvod_batch_track(h, T.ctypes.data_as(C.c_void_p), None, None, 0, stream, 1)
        cnt = np.zeros((n, 8), np.int32)
        lib.scvod_batch_counts(h, cnt.ctypes.data_as(C.c_void_p))
        seen.append((first, cnt[:, 0].copy()))
        return rc
    o = np.ascontiguousarray(offs, np.int32)
    rc = lib.scvod_sequence_ingest(ctx.h, C.c_void_p(host.data_ptr()), o.ctypes.data_as(C.c_void_p), len(o) - 1, chunk, 0, CB(consumers), None)
    assert rc == 0, lib.scvod_last_error(ctx.h)
    return seen


def test_batch_track_then_ingest_then_longer_ingest_on_one_ctx(scvod):
    import synth
    import torch
    P = scvod.make_params("parkinglot")
    pts, offs, poses, _ = synth.make_batch(3, 0, 12, "PARK")
    host = pts.contiguous().pin_memory()
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=12)
    d = pts.cuda()
    o4 = np.asarray(offs[:5], np.int32)
    ctx.batch_process(d, o4)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    T = np.zeros((4, 12), np.float32)
    for s in range(3):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)  # uses the pinned staging ring
    ref = ctx.batch_fetch_track(0)["pt_dyn"].copy()
    a = _ingest(scvod, ctx, host, offs[:7], 3)          # first ingest of the ctx: grows the pinned offset table
    b = _ingest(scvod, ctx, host, offs, 4)              # longer: grows it again while the ring is alive
    assert [f for f, _ in a] == [0, 3] and [f for f, _ in b] == [0, 4, 8]
    n = np.diff(offs)
    for first, cnt in b:
        assert np.array_equal(cnt, n[first:first + len(cnt)])
    ctx.batch_process(d, o4)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    ctx.batch_track(T)
    assert np.array_equal(ctx.batch_fetch_track(0)["pt_dyn"], ref)
    ctx.close()  # (no double free of the ring)


def test_probe_between_two_batch_tracks_does_not_leak_its_transform(scvod):
    import synth
    P = scvod.make_params("semantickitti")
    pts, offs, poses, _ = synth.make_batch(5, 640, 3, "K64")
    ctx = scvod.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=3)
    d = pts.cuda()
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    T = np.zeros((3, 12), np.float32)
    for s in range(2):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    ref = [ctx.batch_fetch_track(s) for s in range(3)]
    r1 = ctx.batch_fetch(1)
    a = r1["apri"][:500]
    xyzi = np.stack([a["x"], a["y"], a["z"], a["intensity"]], 1)
    bogus = np.array([0, 1, 0, 5, -1, 0, 0, 7, 0, 0, 1, 0.3], np.float32)
    ctx.track_probe(xyzi, [0, len(a)], bogus, r1["vox_key"], None)
    lib = ctx.lib
    from scvod_py import TrackResult
    assert lib.scvod_batch_fetch_track(ctx.h, 0, C.byref(TrackResult())) == -5, "the probe overwrote the batch's tracking scratch"
    ctx.batch_track(T)  # same host T as before: must be uploaded again
    for s in range(3):
        t = ctx.batch_fetch_track(s)
        for k in ("cluster_state", "n_unique", "pair_label", "pair_count", "pt_dyn"):
            assert np.array_equal(t[k], ref[s][k]), (s, k)
    ctx.close()


def test_batch_of_empty_scans(scvod):
    import torch
    P = scvod.make_params("semantickitti")
    ctx = scvod.Ctx(P, max_points_total=1024, max_scans=3)
    d = torch.zeros((4, 4), dtype=torch.float32, device="cuda")
    offs = np.zeros(4, np.int32)
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    ctx.batch_track(np.zeros((3, 12), np.float32))
    for s in range(3):
        t = ctx.batch_fetch_track(s)
        assert t["n_clusters"] == 0 and t["n_apri"] == 0 and t["n_dynamic_points"] == 0
    ctx.close()
