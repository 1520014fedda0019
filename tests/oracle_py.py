"""ctypes loader of the ORACLE (oracle/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
from scvod_py import APRI_DTYPE, PLANE_DTYPE, Params, PwParams  # POD layouts of include/scvod.h

_lib = None


def load():
    global _lib
    if _lib is not None:
        return Oracle(_lib)
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("patchwork_oracle.cpp", "ssc_oracle.cpp", "tracking_oracle.cpp", "loader_oracle.cpp", "oracle.h")]
    if (not os.path.exists(so)) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    _lib = C.CDLL(so)
    _lib.oracle_libm_atan2f.restype = C.c_float
    _lib.oracle_libm_atan2f.argtypes = [C.c_float, C.c_float]
    _lib.oracle_libm_atan2.restype = C.c_double
    _lib.oracle_libm_atan2.argtypes = [C.c_double, C.c_double]
    return Oracle(_lib)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def params_default(self):
        p = Params()
        self.lib.oracle_params_default(C.byref(p))
        return p

    def grid_dims(self, p):
        o = [C.c_int32() for _ in range(4)]
        self.lib.oracle_grid_dims(C.byref(p), *[C.byref(x) for x in o])
        return tuple(x.value for x in o)

    def patchwork(self, params, xyzi, sort_mode=1, pw=None):
        a = np.ascontiguousarray(xyzi, np.float32)
        n = a.shape[0]
        cls = np.zeros(max(n, 1), np.uint8)
        g = np.zeros(max(n, 1), np.int32)
        ng = np.zeros(max(n, 1), np.int32)
        planes = np.zeros(1024, PLANE_DTYPE)
        n_g, n_ng, n_p = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.oracle_patchwork(C.byref(params), C.byref(pw) if pw is not None else None, _p(a), n, sort_mode, _p(cls), _p(g), C.byref(n_g), _p(ng),
                                  C.byref(n_ng), _p(planes), C.byref(n_p))
        return dict(cls=cls[:n], ground_idx=g[:n_g.value], nonground_idx=ng[:n_ng.value], planes=planes[:n_p.value])

    def bin(self, params, xyzi, apply_filter=True):
        a = np.ascontiguousarray(xyzi, np.float32)
        n = a.shape[0]
        apri = np.zeros(max(n, 1), APRI_DTYPE)
        src = np.zeros(max(n, 1), np.int32)
        rej = np.zeros(max(n, 1), np.int32)
        nk, nr = C.c_int32(), C.c_int32()
        self.lib.oracle_bin(C.byref(params), _p(a), n, int(apply_filter), _p(apri), _p(src), C.byref(nk), _p(rej),
                            C.byref(nr))
        return dict(apri=apri[:nk.value], src=src[:nk.value], rejected=rej[:nr.value])

    def voxelize(self, params, apri):
        a = np.ascontiguousarray(apri)
        n = a.shape[0]
        key = np.zeros(max(n, 1), np.int32)
        beg = np.zeros(n + 1, np.int32)
        pts = np.zeros(max(n, 1), np.int32)
        av = np.zeros(max(n, 1), np.float32)
        cov = np.zeros(max(n, 1), np.float32)
        idx3 = np.zeros((max(n, 1), 3), np.int32)
        cen = np.zeros((max(n, 1), 4), np.float32)
        nv = C.c_int32()
        self.lib.oracle_voxelize(C.byref(params), _p(a), n, _p(key), _p(beg), _p(pts), _p(av), _p(cov), _p(idx3),
                                 _p(cen), C.byref(nv))
        v = nv.value
        return dict(vox_key=key[:v], vox_pt_begin=beg[:v + 1], vox_pts=pts[:n], vox_av=av[:v], vox_cov=cov[:v],
                    idx3=idx3[:v], center=cen[:v])

    def pose_delta(self, pre, nxt):
        a = np.ascontiguousarray(pre, np.float32)
        b = np.ascontiguousarray(nxt, np.float32)
        T = np.zeros(12, np.float32)
        self.lib.oracle_pose_delta(_p(a), _p(b), _p(T))
        return T

    def track_probe(self, params, xyzi, offsets, T, next_keys, next_labels):
        a = np.ascontiguousarray(xyzi, np.float32)
        o = np.ascontiguousarray(offsets, np.int32)
        t = np.ascontiguousarray(T, np.float32)
        k = np.ascontiguousarray(next_keys, np.int32)
        l = np.ascontiguousarray(next_labels, np.int32)
        n_pts = int(o[-1])
        hit = np.zeros(max(n_pts, 1), np.int32)
        uq = np.zeros(max(n_pts, 1), np.int32)
        ub = np.zeros(o.shape[0], np.int32)
        self.lib.oracle_track_probe(C.byref(params), _p(a), _p(o), o.shape[0] - 1, _p(t), _p(k), _p(l), k.shape[0],
                                    _p(hit), _p(uq), _p(ub))
        return hit[:n_pts], uq[:ub[-1]], ub

    def toy_tracking(self, params, apri_a, apri_b, pose_a, pose_b, car=2, tree=1):
        a = np.ascontiguousarray(apri_a)
        b = np.ascontiguousarray(apri_b)
        pa = np.ascontiguousarray(pose_a, np.float32)
        pb = np.ascontiguousarray(pose_b, np.float32)
        states = np.zeros((max(len(a), 1), 3), np.int32)
        labels = np.zeros(max(len(b), 1), np.int32)
        ns, nv, dyn, nc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.oracle_toy_tracking(C.byref(params), _p(a), len(a), _p(b), len(b), _p(pa), _p(pb), car, tree, _p(states),
                                     C.byref(ns), _p(labels), C.byref(nv), C.byref(dyn), C.byref(nc))
        return states[:ns.value], labels[:nv.value], dyn.value, nc.value

    def cluster(self, params, apri):
        a = np.ascontiguousarray(apri)
        n = a.shape[0]
        out = np.zeros(max(n, 1), np.int32)
        mx = C.c_int32()
        nc = self.lib.oracle_cluster(C.byref(params), _p(a), n, _p(out), C.byref(mx))
        return out[:n], nc, mx.value

    def cluster_last_name(self, params, apri):
        """canonical name (smallest point index) of the cluster that carries Frame::max_name as ssc.cpp:354 stores it, or -1;
        info = {K, opener of K, final number of the opener's cluster, openers in it, renames of K, merges}"""
        a = np.ascontiguousarray(apri)
        info = np.zeros(6, np.int64)
        return int(self.lib.oracle_cluster_last_name(C.byref(params), _p(a), a.shape[0], _p(info))), info

    def atan2_overload_flips(self, params, xyzi):
        """{kept points, sector_idx flips, azimuth_idx flips, filter-verdict flips} between atan2f and float(atan2(double, double))"""
        a = np.ascontiguousarray(xyzi, np.float32)
        out = np.zeros(4, np.int64)
        self.lib.oracle_atan2_overload_flips(C.byref(params), _p(a), a.shape[0], _p(out))
        return out

    def cluster_types(self, params, apri, pt_cluster, car_label=2, other_label=1):
        a = np.ascontiguousarray(apri)
        c = np.ascontiguousarray(pt_cluster, np.int32)
        out = np.zeros(max(len(a), 1), np.int32)
        self.lib.oracle_cluster_types(C.byref(params), _p(a), len(a), _p(c), car_label, other_label, _p(out))
        return out[:len(a)]

    def nn_search(self, map_xyz, query_xyz, radius):
        m = np.ascontiguousarray(map_xyz, np.float32)
        q = np.ascontiguousarray(query_xyz, np.float32)
        nq = q.shape[0]
        idx = np.zeros(max(nq, 1), np.int32)
        sq = np.zeros(max(nq, 1), np.float32)
        w = np.zeros(max(nq, 1), np.uint8)
        self.lib.oracle_nn_search(_p(m), m.shape[0], _p(q), nq, C.c_float(radius), _p(idx), _p(sq), _p(w))
        return idx[:nq], sq[:nq], w[:nq]

    def voxelgrid(self, xyzi, leaf=(0.08, 0.08, 0.08), labels=None, max_intensity=1.0, sort_mode=1):
        a = np.ascontiguousarray(xyzi, np.float32)
        n = a.shape[0]
        lf = np.asarray(leaf, np.float32)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint32)
        out = np.zeros((max(n, 1), 4), np.float32)
        n_out = C.c_int32(0)
        rc = self.lib.oracle_voxelgrid(_p(a), None if lab is None else _p(lab), n, C.c_float(max_intensity), _p(lf), sort_mode,
                                       _p(out), C.byref(n_out))
        return out[:n_out.value], rc

    def svd3(self, cov):
        c = np.ascontiguousarray(cov, np.float32).reshape(9)
        sv = np.zeros(3, np.float32)
        U = np.zeros(9, np.float32)
        self.lib.oracle_svd3(_p(c), _p(sv), _p(U))
        return sv, U.reshape(3, 3)

    def time_process(self, params, xyzi, offsets):
        a = np.ascontiguousarray(xyzi, np.float32)
        o = np.ascontiguousarray(offsets, np.int32)
        st = (C.c_double * 3)()
        cs = C.c_int64()
        self.lib.oracle_time_process(C.byref(params), _p(a), _p(o), o.shape[0] - 1, st, C.byref(cs))
        return [st[0], st[1], st[2]], cs.value

    def track_decide(self, params, apri_a, cl_a, ty_a, apri_b, cl_b, ty_b, T, car=2):
        a, b = np.ascontiguousarray(apri_a), np.ascontiguousarray(apri_b)
        ca, ta = np.ascontiguousarray(cl_a, np.int32), np.ascontiguousarray(ty_a, np.int32)
        cb, tb = np.ascontiguousarray(cl_b, np.int32), np.ascontiguousarray(ty_b, np.int32)
        t = np.ascontiguousarray(T, np.float32)
        out = np.zeros((max(len(a), 1), 4), np.int32)
        pb = np.zeros(len(a) + 1, np.int32)
        pairs = np.zeros((max(len(a), 1), 2), np.int32)
        n = C.c_int32()
        self.lib.oracle_track_decide(C.byref(params), _p(a), len(a), _p(ca), _p(ta), _p(b), len(b), _p(cb), _p(tb), _p(t), car,
                                     _p(out), C.byref(n), _p(pb), _p(pairs))
        k = n.value
        return dict(clusters=out[:k], pair_begin=pb[:k + 1], pairs=pairs[:pb[k]])

    def sequence_tracking(self, params, apri, offs, pt_cluster, pt_type, poses, car=2, chain=1):
        a = np.ascontiguousarray(apri)
        o = np.ascontiguousarray(offs, np.int32)
        cl, ty = np.ascontiguousarray(pt_cluster, np.int32), np.ascontiguousarray(pt_type, np.int32)
        ps = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        dyn = np.zeros(max(len(a), 1), np.uint8)
        nd = C.c_int32()
        self.lib.oracle_sequence_tracking(C.byref(params), _p(a), _p(o), len(o) - 1, _p(cl), _p(ty), _p(ps), car, int(chain),
                                          _p(dyn), C.byref(nd))
        return dyn[:len(a)], nd.value

    def sequence_tracking_literal(self, params, apri, offs, pt_cluster, pt_type, collide, poses, car=2, chain=3):
        """SSC::segDF's loop with the reference's literal max_name (ssc.cpp:354, 1357, 1401); collide[s] from cluster_last_name"""
        a = np.ascontiguousarray(apri)
        o = np.ascontiguousarray(offs, np.int32)
        cl, ty = np.ascontiguousarray(pt_cluster, np.int32), np.ascontiguousarray(pt_type, np.int32)
        co = np.ascontiguousarray(collide, np.int32)
        assert len(co) == len(o) - 1
        ps = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        dyn = np.zeros(max(len(a), 1), np.uint8)
        nd = C.c_int32()
        st = np.zeros(4, np.int64)
        self.lib.oracle_sequence_tracking_literal(C.byref(params), _p(a), _p(o), len(o) - 1, _p(cl), _p(ty), _p(co), _p(ps), car,
                                                  int(chain), _p(dyn), C.byref(nd), _p(st))
        return dyn[:len(a)], nd.value, st

    def reference_chain(self, params, res, names, types, poses, unknown=None, chain=3):
        """SSC::segDF's tracking loop as the reference runs it, Frame::max_name included (ssc.cpp:354): per scan the cluster
        that still carries the last running number (the literal visiting loop), then the literal chain.  res / names / types:
        per-scan dicts / arrays of a segmented batch; unknown: scans whose name the device reported as undetermined (it then
        hands out a fresh number there: the comparison follows)."""
        collide = np.asarray([self.cluster_last_name(params, r["apri"])[0] for r in res], np.int32)
        if unknown is not None:
            collide[np.asarray(unknown, bool)] = -1
        apri = np.concatenate([r["apri"] for r in res])
        ao = np.concatenate([[0], np.cumsum([r["n_apri"] for r in res])]).astype(np.int32)
        dyn, nd, _ = self.sequence_tracking_literal(params, apri, ao, np.concatenate(names), np.concatenate(types), collide,
                                                    np.asarray(poses, np.float32), chain=chain)
        return dyn, nd

    def time_sequence(self, params, xyzi, offsets, poses, car=2, other=1, want_labels=True):
        a = np.ascontiguousarray(xyzi, np.float32)
        o = np.ascontiguousarray(offsets, np.int32)
        ps = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        st = (C.c_double * 6)()
        cs = C.c_int64()
        lab = np.zeros(max(len(a), 1), np.uint8) if want_labels else None
        self.lib.oracle_time_sequence(C.byref(params), _p(a), _p(o), len(o) - 1, _p(ps), car, other, st,
                                      _p(lab) if want_labels else None, C.byref(cs))
        return list(st), (lab[:len(a)] if want_labels else None), cs.value
