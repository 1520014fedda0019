"""ctypes plumbing over the C-ABI of libscvod.so (include/scvod.h) for tests/ and bench.py.

This is NOT the product's host side (that is the C++ facade in ../host/, mirroring the
reference's SSC / PatchWork classes); it only lets Python drive the same extern "C" entry
points with numpy arrays and torch device tensors.  There is no CPU fallback: loading fails
loudly when libscvod.so is missing and scvod_create fails when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libscvod.so")

MAX_PATCHES = 1024


class Params(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "sensor_height", "min_dis", "max_dis", "min_angle", "max_angle", "min_azimuth", "max_azimuth",
        "range_res", "sector_res", "azimuth_res", "occupancy", "max_z", "min_z", "car_square")] + \
               [("toBeClass", C.c_int32), ("reserved", C.c_int32)]


class PwParams(C.Structure):
    _fields_ = [("num_iter", C.c_int32), ("num_lpr", C.c_int32), ("num_min_pts", C.c_int32),
                ("num_rings_of_interest", C.c_int32), ("num_sectors_each_zone", C.c_int32 * 4),
                ("num_rings_each_zone", C.c_int32 * 4), ("th_seeds", C.c_double), ("th_dist", C.c_double),
                ("max_range", C.c_double), ("min_range", C.c_double), ("uprightness_thr", C.c_double),
                ("adaptive_seed_selection_margin", C.c_double), ("elevation_thr", C.c_double * 4),
                ("flatness_thr", C.c_double * 4)]


APRI_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("range", "f4"), ("angle", "f4"), ("azimuth", "f4"),
                       ("intensity", "f4"), ("range_idx", "i4"), ("sector_idx", "i4"), ("azimuth_idx", "i4"),
                       ("voxel_idx", "i4")])
PLANE_DTYPE = np.dtype([("normal", "f4", 3), ("mean", "f4", 3), ("sv", "f4", 3), ("n_pts", "i4"), ("n_ground", "i4"),
                        ("status", "i4")])
assert APRI_DTYPE.itemsize == 44 and PLANE_DTYPE.itemsize == 48


class ScanResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_points", "n_ground", "n_nonground", "n_dropped", "n_apri", "n_rejected",
                                          "n_voxels", "n_patches")] + \
               [(n, C.c_void_p) for n in ("cls", "ground_idx", "nonground_idx", "planes", "apri", "apri_src",
                                          "rejected_src", "vox_key", "vox_pt_begin", "vox_pts", "vox_av", "vox_cov")]


class TrackResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_apri", "n_clusters", "n_car_points", "n_dynamic_clusters", "n_dynamic_points",
                                          "reserved")] + \
               [(n, C.c_void_p) for n in ("cluster_root", "cluster_size", "cluster_state", "n_unique", "pair_begin",
                                          "pair_label", "pair_count", "pt_dyn")]


# ssc/ keys of config/*.yaml -> Params fields (include/utility.h:283-310)
YAML_KEYS = {"sensor_height_": "sensor_height", "min_dis_": "min_dis", "max_dis_": "max_dis", "min_angle_": "min_angle",
             "max_angle_": "max_angle", "min_azimuth_": "min_azimuth", "max_azimuth_": "max_azimuth",
             "range_res_": "range_res", "sector_res_": "sector_res", "azimuth_res_": "azimuth_res",
             "occupancy_": "occupancy", "max_z_": "max_z", "min_z_": "min_z", "car_square_": "car_square",
             "toBeClass_": "toBeClass"}

# values of the two YAML files shipped by the reference (config/semantickitti.yaml:24-55, config/parkinglot.yaml:23-48)
PRESETS = {
    "semantickitti": dict(sensor_height=1.73, min_dis=1.5, max_dis=30.0, min_angle=0.0, max_angle=360.0,
                          min_azimuth=-40.0, max_azimuth=80.0, range_res=0.4, sector_res=1.2, azimuth_res=2.0,
                          occupancy=0.4, max_z=0.8, min_z=-1.2, car_square=30.0, toBeClass=10),
    "parkinglot": dict(sensor_height=1.83, min_dis=0.8, max_dis=40.0, min_angle=0.0, max_angle=360.0,
                       min_azimuth=-30.0, max_azimuth=60.0, range_res=0.4, sector_res=1.2, azimuth_res=2.0,
                       occupancy=0.8, max_z=1.0, min_z=-1.0, car_square=2.0, toBeClass=6),
    # the parameters behind the reference's published seq-05 row (doc/note.txt:36: skip 5, max_z 4.0, min_z 1.0, car_square 50; its
    # refine_height / car_height / car_angle feed code that is commented out or out of scope): the committed YAML with those three
    "semantickitti_seq05": dict(sensor_height=1.73, min_dis=1.5, max_dis=30.0, min_angle=0.0, max_angle=360.0,
                                min_azimuth=-40.0, max_azimuth=80.0, range_res=0.4, sector_res=1.2, azimuth_res=2.0,
                                occupancy=0.4, max_z=4.0, min_z=1.0, car_square=50.0, toBeClass=10),
    # BASELINE.json configs[4]: OS1-128 stream with a 2x finer voxel grid
    "os128_fine": dict(sensor_height=1.73, min_dis=1.5, max_dis=30.0, min_angle=0.0, max_angle=360.0,
                       min_azimuth=-40.0, max_azimuth=80.0, range_res=0.2, sector_res=0.6, azimuth_res=1.0,
                       occupancy=0.4, max_z=0.8, min_z=-1.2, car_square=30.0, toBeClass=10),
}

_lib = None


def load_lib():
    """Load libscvod.so (after torch, so both share torch's HIP runtime).  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (must come first: one libamdhip64 per process)
    except Exception:
        pass
    path = os.path.abspath(os.environ.get("SCVOD_LIB", LIB_PATH))  # (SCVOD_LIB: a development build, e.g. libscvod_prof.so)
    if not os.path.exists(path):
        # the library is a build product (git-ignored): compile it in-tree when a hipcc is around
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.check_call(["make", "-C", os.path.dirname(path)])
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the SCV-OD hot path has no CPU fallback)")
    lib = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "scvod_params_default": (None, [C.POINTER(Params)]),
        "scvod_pw_params_default": (None, [C.POINTER(PwParams)]),
        "scvod_grid_dims": (None, [C.POINTER(Params)] + [C.POINTER(i32)] * 4),
        "scvod_create": (C.c_int, [C.POINTER(Params), C.POINTER(PwParams), C.c_int, i64, i32, C.POINTER(vp)]),
        "scvod_destroy": (None, [vp]),
        "scvod_last_error": (C.c_char_p, [vp]),
        "scvod_arena_bytes": (i64, [vp]),
        "scvod_process_scan": (C.c_int, [vp, vp, i32, C.POINTER(ScanResult)]),
        "scvod_patchwork": (C.c_int, [vp, vp, i32, C.POINTER(ScanResult)]),
        "scvod_bin_scan": (C.c_int, [vp, vp, i32, i32, i32, C.POINTER(ScanResult)]),
        "scvod_voxelize": (C.c_int, [vp, vp, i32, C.POINTER(ScanResult)]),
        "scvod_pose_delta": (None, [vp, vp, vp]),
        "scvod_track_probe": (C.c_int, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
        "scvod_batch_process": (C.c_int, [vp, vp, vp, i32, vp, i32]),
        "scvod_batch_counts": (C.c_int, [vp, vp]),
        "scvod_batch_fetch": (C.c_int, [vp, i32, C.POINTER(ScanResult)]),
        "scvod_batch_cluster": (C.c_int, [vp, vp, i32]),
        "scvod_batch_fetch_clusters": (C.c_int, [vp, i32, vp, i32]),
        "scvod_cluster": (C.c_int, [vp, vp, i32, vp]),
        "scvod_batch_cluster_types": (C.c_int, [vp, vp, i32]),
        "scvod_batch_fetch_cluster_types": (C.c_int, [vp, i32, i32, i32, vp, i32]),
        "scvod_batch_track": (C.c_int, [vp, vp, vp, vp, i32, vp, i32]),
        "scvod_batch_fetch_track": (C.c_int, [vp, i32, C.POINTER(TrackResult)]),
        "scvod_batch_export_table": (C.c_int, [vp, i32, vp, i64, vp]),
        "scvod_set_track_mode": (C.c_int, [vp, i32, i32, i32]),
        "scvod_set_cluster_exact": (C.c_int, [vp, i32]),
        "scvod_batch_cluster_stats": (C.c_int, [vp, vp]),
        "scvod_batch_cluster_rule_stats": (C.c_int, [vp, vp]),
        "scvod_batch_cluster_help_stats": (C.c_int, [vp, vp]),
        "scvod_set_max_name_literal": (C.c_int, [vp, i32]),
        "scvod_batch_cluster_last_name": (C.c_int, [vp, vp, i32, vp]),
        "scvod_set_chain_capacity": (C.c_int, [vp, i64]),
        "scvod_chain_workspace_bytes": (i64, [vp]),
        "scvod_get_params": (C.c_int, [vp, vp]),
        "scvod_set_track_owned": (C.c_int, [vp, i32]),
        "scvod_set_track_halo": (C.c_int, [vp, vp, i32]),
        "scvod_batch_track_chains": (C.c_int, [vp, vp, i32]),
        "scvod_chain_state_bytes": (i64, [vp]),
        "scvod_chain_export_state": (C.c_int, [vp, i32, i32, vp, i64, vp]),
        "scvod_batch_track_resume": (C.c_int, [vp, vp, i32, vp, i32]),
        "scvod_batch_track_compare": (C.c_int, [vp, vp, i32, vp, vp]),
        "scvod_batch_track_compare_device": (C.c_int, [vp, vp, i32, vp, vp]),
        "scvod_batch_map_accumulate_range": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "scvod_batch_track_stats": (C.c_int, [vp, vp]),
        "scvod_batch_track_tables": (C.c_int, [vp, vp]),
        "scvod_map_create": (C.c_int, [C.c_int, i64, f32, C.POINTER(vp)]),
        "scvod_map_destroy": (None, [vp]),
        "scvod_map_last_error": (C.c_char_p, [vp]),
        "scvod_map_capacity": (i64, [vp]),
        "scvod_map_clear": (C.c_int, [vp, vp]),
        "scvod_pose_matrix": (None, [vp, vp]),
        "scvod_batch_map_accumulate": (C.c_int, [vp, vp, vp, i32, vp]),
        "scvod_map_export": (C.c_int, [vp, vp, i64, C.POINTER(i64), vp]),
        "scvod_map_merge": (C.c_int, [vp, vp, i64, vp]),
        "scvod_map_export_parts": (C.c_int, [vp, i32, vp, i64, vp, vp]),
        "scvod_map_export_parts_padded": (C.c_int, [vp, i32, vp, i64, vp, vp]),
        "scvod_map_points": (C.c_int, [vp, vp, vp, i64, C.POINTER(i64), vp]),
        "scvod_batch_timings": (C.c_int, [vp, vp, vp, i32]),
        "scvod_set_timing": (C.c_int, [vp, i32]),
        "scvod_nn_search": (C.c_int, [vp, vp, i32, vp, i32, f32, vp, vp, vp]),
        "scvod_nn_radius_search": (C.c_int, [vp, vp, i32, vp, i32, f32, vp, vp]),
        "scvod_nn_search_device": (C.c_int, [vp, vp, i32, vp, i32, f32, vp, vp, vp, vp]),
        "scvod_batch_voxelgrid": (C.c_int, [vp, vp, vp, vp, i32, vp, f32, vp, i64, vp, vp]),
        "scvod_voxelgrid": (C.c_int, [vp, vp, vp, i32, vp, f32, vp, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # raises AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["scvod_params_default", "scvod_pw_params_default", "scvod_grid_dims", "scvod_create",
                    "scvod_destroy", "scvod_last_error", "scvod_arena_bytes", "scvod_process_scan", "scvod_patchwork",
                    "scvod_bin_scan", "scvod_voxelize", "scvod_pose_delta", "scvod_track_probe", "scvod_batch_process",
                    "scvod_batch_counts", "scvod_batch_fetch", "scvod_batch_cluster", "scvod_batch_fetch_clusters", "scvod_cluster",
                    "scvod_batch_cluster_types", "scvod_batch_fetch_cluster_types",
                    "scvod_batch_track", "scvod_batch_fetch_track", "scvod_set_track_mode", "scvod_set_cluster_exact", "scvod_batch_cluster_stats", "scvod_batch_cluster_rule_stats", "scvod_batch_cluster_help_stats", "scvod_set_max_name_literal", "scvod_batch_cluster_last_name", "scvod_set_chain_capacity", "scvod_chain_workspace_bytes", "scvod_get_params", "scvod_set_track_owned", "scvod_set_track_halo", "scvod_batch_track_chains", "scvod_chain_state_bytes", "scvod_chain_export_state", "scvod_batch_track_resume", "scvod_batch_track_compare", "scvod_batch_track_compare_device", "scvod_batch_map_accumulate_range", "scvod_batch_track_stats", "scvod_batch_export_table", "scvod_batch_track_tables", "scvod_sequence_ingest",
                    "scvod_map_create", "scvod_map_destroy", "scvod_map_last_error", "scvod_map_capacity", "scvod_map_clear",
                    "scvod_pose_matrix", "scvod_batch_map_accumulate", "scvod_map_export", "scvod_map_export_parts", "scvod_map_export_parts_padded", "scvod_map_merge", "scvod_map_points",
                    "scvod_batch_timings", "scvod_set_timing", "scvod_nn_search", "scvod_nn_radius_search", "scvod_nn_search_device", "scvod_batch_voxelgrid", "scvod_voxelgrid"]


def make_params(preset=None, **kw):
    p = Params()
    load_lib().scvod_params_default(C.byref(p))
    vals = dict(PRESETS[preset]) if preset else {}
    vals.update(kw)
    for k, v in vals.items():
        setattr(p, k, v)
    return p


def params_from_yaml(path):
    """Reads the `ssc:` block of a reference YAML file (config/*.yaml) into Params; keys that the
    hot path does not use are ignored, missing keys keep the nh.param<> defaults."""
    import yaml
    with open(path) as f:
        doc = yaml.safe_load(f)
    p = make_params()
    for k, v in (doc.get("ssc") or {}).items():
        if k in YAML_KEYS:
            setattr(p, YAML_KEYS[k], int(v) if YAML_KEYS[k] == "toBeClass" else float(v))
    return p


def grid_dims(p):
    out = [C.c_int32() for _ in range(4)]
    load_lib().scvod_grid_dims(C.byref(p), *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def _unpack(r):
    d = {k: getattr(r, k) for k in ("n_points", "n_ground", "n_nonground", "n_dropped", "n_apri", "n_rejected",
                                    "n_voxels", "n_patches")}
    d["cls"] = _arr(r.cls, r.n_points, np.uint8)
    d["ground_idx"] = _arr(r.ground_idx, r.n_ground, np.int32)
    d["nonground_idx"] = _arr(r.nonground_idx, r.n_nonground, np.int32)
    d["planes"] = _arr(r.planes, r.n_patches, PLANE_DTYPE)
    d["apri"] = _arr(r.apri, r.n_apri, APRI_DTYPE)
    d["apri_src"] = _arr(r.apri_src, r.n_apri, np.int32)
    d["rejected_src"] = _arr(r.rejected_src, r.n_rejected, np.int32)
    d["vox_key"] = _arr(r.vox_key, r.n_voxels, np.int32)
    d["vox_pt_begin"] = _arr(r.vox_pt_begin, r.n_voxels + 1, np.int32)
    d["vox_pts"] = _arr(r.vox_pts, r.n_apri if r.n_voxels else 0, np.int32)
    d["vox_av"] = _arr(r.vox_av, r.n_voxels, np.float32)
    d["vox_cov"] = _arr(r.vox_cov, r.n_voxels, np.float32)
    return d


class ScvodError(RuntimeError):
    pass


class Ctx:
    def __init__(self, params, max_points_total, max_scans=1, device=0, pw=None):
        self.lib = load_lib()
        self.params = params
        self.device = int(device)
        self.h = C.c_void_p()
        rc = self.lib.scvod_create(C.byref(params), C.byref(pw) if pw is not None else None, device,
                                   int(max_points_total), int(max_scans), C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise ScvodError(f"scvod_create failed with status {rc} "
                             "(-2 = no HIP device: the SCV-OD path is GPU-only, there is no CPU fallback)")

    def close(self):
        if self.h:
            self.lib.scvod_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise ScvodError(f"status {rc}: {self.lib.scvod_last_error(self.h).decode()}")

    @staticmethod
    def _f32(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return a, a.ctypes.data_as(C.c_void_p)

    def process_scan(self, xyzi):
        a, p = self._f32(xyzi)
        r = ScanResult()
        self._chk(self.lib.scvod_process_scan(self.h, p, a.shape[0], C.byref(r)))
        return _unpack(r)

    def patchwork(self, xyzi):
        a, p = self._f32(xyzi)
        r = ScanResult()
        self._chk(self.lib.scvod_patchwork(self.h, p, a.shape[0], C.byref(r)))
        return _unpack(r)

    def bin_scan(self, xyzi, apply_filter=True, with_voxels=True):
        a, p = self._f32(xyzi)
        r = ScanResult()
        self._chk(self.lib.scvod_bin_scan(self.h, p, a.shape[0], int(apply_filter), int(with_voxels), C.byref(r)))
        return _unpack(r)

    def voxelize(self, apri):
        a = np.ascontiguousarray(apri)
        r = ScanResult()
        self._chk(self.lib.scvod_voxelize(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], C.byref(r)))
        return _unpack(r)

    def pose_delta(self, pose_pre, pose_next):
        a, pa = self._f32(pose_pre)
        b, pb = self._f32(pose_next)
        T = np.zeros(12, np.float32)
        self.lib.scvod_pose_delta(pa, pb, T.ctypes.data_as(C.c_void_p))
        return T

    def track_probe(self, xyzi, offsets, T, next_keys, next_labels):
        a, pa = self._f32(xyzi)
        o, po = self._i32(offsets)
        t, pt = self._f32(T)
        k, pk = self._i32(next_keys)
        n_c = o.shape[0] - 1
        n_pts = int(o[-1])
        if next_labels is not None:
            l, pl = self._i32(next_labels)
        else:
            l, pl = None, None
        hit = np.zeros(max(n_pts, 1), np.int32)
        uq = np.zeros(max(n_pts, 1), np.int32)
        ub = np.zeros(n_c + 1, np.int32)
        self._chk(self.lib.scvod_track_probe(self.h, pa, po, n_c, pt, pk, pl, k.shape[0], hit.ctypes.data_as(C.c_void_p),
                                             uq.ctypes.data_as(C.c_void_p), ub.ctypes.data_as(C.c_void_p)))
        return hit[:n_pts], uq[:ub[-1]], ub

    # ---- device-resident batch API (torch tensors) ----
    def batch_process(self, d_xyzi, scan_offsets, stream=None, sync=True):
        off, po = self._i32(scan_offsets)
        self._n_scans = off.shape[0] - 1
        ptr = C.c_void_p(d_xyzi.data_ptr())
        self._chk(self.lib.scvod_batch_process(self.h, ptr, po, self._n_scans, C.c_void_p(stream or 0), int(sync)))

    def batch_counts(self):
        out = np.zeros((self._n_scans, 8), np.int32)
        self._chk(self.lib.scvod_batch_counts(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def batch_fetch(self, s):
        r = ScanResult()
        self._chk(self.lib.scvod_batch_fetch(self.h, int(s), C.byref(r)))
        return _unpack(r)

    def batch_cluster(self, stream=None, sync=True):
        self._chk(self.lib.scvod_batch_cluster(self.h, C.c_void_p(stream or 0), int(sync)))

    def batch_fetch_clusters(self, s, cap):
        out = np.zeros(max(cap, 1), np.int32)
        n = self.lib.scvod_batch_fetch_clusters(self.h, int(s), out.ctypes.data_as(C.c_void_p), int(cap))
        if n < 0:
            self._chk(n)
        return out[:n]

    def batch_cluster_types(self, stream=None, sync=True):
        self._chk(self.lib.scvod_batch_cluster_types(self.h, C.c_void_p(stream or 0), int(sync)))

    def batch_fetch_cluster_types(self, s, cap, car_label=2, other_label=1):
        out = np.zeros(max(cap, 1), np.int32)
        n = self.lib.scvod_batch_fetch_cluster_types(self.h, int(s), car_label, other_label, out.ctypes.data_as(C.c_void_p), int(cap))
        if n < 0:
            self._chk(n)
        return out[:n]

    def cluster(self, apri):
        a = np.ascontiguousarray(apri)
        out = np.zeros(max(a.shape[0], 1), np.int32)
        self._chk(self.lib.scvod_cluster(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out[:a.shape[0]]

    def batch_track(self, T, next_scan=None, ext_tables=None, stream=None, sync=True):
        """T [n_scans, 12] (row s: delta of scan s to its successor); next_scan [n_scans] or None (s + 1);
        ext_tables: list of torch device tensors written by batch_export_table on another shard."""
        t, pt = self._f32(T)
        assert t.size == 12 * self._n_scans, "one 3x4 transform per scan"
        pn = None
        if next_scan is not None:
            nx, pn = self._i32(next_scan)
            assert nx.shape[0] == self._n_scans
        n_ext = len(ext_tables) if ext_tables else 0
        ptrs = (C.c_void_p * max(n_ext, 1))(*[C.c_void_p(e.data_ptr()) for e in (ext_tables or [])])
        self._ext_keep = ext_tables  # the device buffers must outlive the asynchronous launch
        self._chk(self.lib.scvod_batch_track(self.h, pt, pn, ptrs if n_ext else None, n_ext, C.c_void_p(stream or 0), int(sync)))

    # ---- one sequence over several shards (include/scvod.h: scvod_set_track_owned ...) ----
    def set_track_owned(self, first_owned_scan):
        self._chk(self.lib.scvod_set_track_owned(self.h, int(first_owned_scan)))

    def set_track_halo(self, is_halo):
        m = np.ascontiguousarray(is_halo, np.uint8)
        self._chk(self.lib.scvod_set_track_halo(self.h, m.ctypes.data_as(C.c_void_p), len(m)))

    def batch_track_chains(self):
        """first scan of every chain of the last batch_track"""
        out = np.zeros(max(self._n_scans, 1), np.int32)
        n = self.lib.scvod_batch_track_chains(self.h, out.ctypes.data_as(C.c_void_p), len(out))
        if n < 0:
            self._chk(n)
        return out[:n].copy()

    def chain_export_state(self, chain, which, stream=None):
        """the state chain `chain` ended in (which=1) / assumed at its first own step (which=0): a torch uint8 device tensor"""
        import torch
        nb = int(self.lib.scvod_chain_state_bytes(self.h))
        buf = torch.zeros(nb, dtype=torch.uint8, device=f"cuda:{self.device}")
        self._chk(self.lib.scvod_chain_export_state(self.h, int(chain), int(which), C.c_void_p(buf.data_ptr()), nb, C.c_void_p(stream or 0)))
        torch.cuda.synchronize(self.device)
        hdr = buf[:16].view(torch.int32).cpu().numpy()
        used = 16 + 32 * int(hdr[0]) + ((4 * int(hdr[2]) + 15) // 16) * 16 + 16 * int(hdr[1])
        return buf[:max(used, 16)].clone()

    def chain_export_states(self, chains, which, stream=None):
        """chain_export_state for several chains: the export kernels back to back, ONE synchronisation, one copy of the headers"""
        import torch
        chains = [int(c) for c in chains]
        if not chains:
            return []
        nb = int(self.lib.scvod_chain_state_bytes(self.h))
        big = torch.empty((len(chains), nb), dtype=torch.uint8, device=f"cuda:{self.device}")
        for k, c in enumerate(chains):
            self._chk(self.lib.scvod_chain_export_state(self.h, c, int(which), C.c_void_p(big[k].data_ptr()), nb, C.c_void_p(stream or 0)))
        hdr = big[:, :16].contiguous().view(torch.int32).cpu().numpy().reshape(len(chains), 4)  # (synchronises)
        out = []
        for k in range(len(chains)):
            used = 16 + 32 * int(hdr[k, 0]) + ((4 * int(hdr[k, 2]) + 15) // 16) * 16 + 16 * int(hdr[k, 1])
            out.append(big[k, :max(used, 16)])
        return out

    def batch_track_resume(self, states, stream=None):
        """states[k]: the record the shard before this one exported (which=1) for chain k, or None"""
        self._resume_keep = states
        ptrs = (C.c_void_p * max(len(states), 1))(*[C.c_void_p(t.data_ptr() if t is not None else 0) for t in states])
        self._chk(self.lib.scvod_batch_track_resume(self.h, ptrs, len(states), C.c_void_p(stream or 0), 1))

    def batch_track_compare(self, states, stream=None):
        """how many chains would be walked again if `states` (as for batch_track_resume) were resumed from; changes nothing"""
        self._resume_keep = states
        ptrs = (C.c_void_p * max(len(states), 1))(*[C.c_void_p(t.data_ptr() if t is not None else 0) for t in states])
        out = C.c_int32(0)
        self._chk(self.lib.scvod_batch_track_compare(self.h, ptrs, len(states), C.byref(out), C.c_void_p(stream or 0)))
        return int(out.value)

    def chain_export_state_into(self, chain, which, buf, stream=None):
        """the record of chain_export_state written into `buf` (a torch uint8 device tensor of fixed size) without a word read on the
        host: a state that needs more room leaves header word 3 = 2 (compare_device then counts the chain as differing)"""
        self._chk(self.lib.scvod_chain_export_state(self.h, int(chain), int(which), C.c_void_p(buf.data_ptr()), int(buf.numel()), C.c_void_p(stream or 0)))

    def batch_track_compare_device(self, states, d_differ, stream=None):
        """batch_track_compare with the verdict ADDED to d_differ (a torch int32 device tensor the caller cleared): asynchronous"""
        self._resume_keep = states
        ptrs = (C.c_void_p * max(len(states), 1))(*[C.c_void_p(t.data_ptr() if t is not None else 0) for t in states])
        self._chk(self.lib.scvod_batch_track_compare_device(self.h, ptrs, len(states), C.c_void_p(d_differ.data_ptr()), C.c_void_p(stream or 0)))

    def batch_fetch_track(self, s):
        r = TrackResult()
        self._chk(self.lib.scvod_batch_fetch_track(self.h, int(s), C.byref(r)))

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32 if dt == np.int32 else C.c_uint8)), shape=(n,)).copy()
        ncl = r.n_clusters
        pb = arr(r.pair_begin, ncl + 1, np.int32)
        npair = int(pb[-1]) if ncl else 0
        return dict(n_apri=r.n_apri, n_clusters=ncl, n_car_points=r.n_car_points, n_dynamic_clusters=r.n_dynamic_clusters,
                    n_dynamic_points=r.n_dynamic_points, cluster_root=arr(r.cluster_root, ncl, np.int32),
                    cluster_size=arr(r.cluster_size, ncl, np.int32), cluster_state=arr(r.cluster_state, ncl, np.int32),
                    n_unique=arr(r.n_unique, ncl, np.int32), pair_begin=pb, pair_label=arr(r.pair_label, npair, np.int32),
                    pair_count=arr(r.pair_count, npair, np.int32), pt_dyn=arr(r.pt_dyn, r.n_apri, np.uint8))

    def set_track_mode(self, chain=True, segment_steps=0, warmup_steps=-1, generic_step=False):
        """chain=True: the reference's sequential tracking chain (default); False: first-order decisions.
        segment_steps 0: chosen per job; warmup_steps -1: keep the current value."""
        self._chk(self.lib.scvod_set_track_mode(self.h, (3 if generic_step else 1) if chain else 0, int(segment_steps), int(warmup_steps)))

    def set_cluster_exact(self, on=True):
        """True / 1 (default of a context since round 6): the components the local rule does not settle are clustered again in visiting
        order whatever their size (k_cc_exact, passes shared with helper blocks); 0: only while they have <= 4096 nodes together (larger
        ones keep "everything found is joined" and are counted: the default of rounds 3-5); 2: exact without the rule; 3: like 1, every
        scan's workgroup on its own"""
        self._chk(self.lib.scvod_set_cluster_exact(self.h, int(on)))

    def batch_cluster_stats(self):
        out = np.zeros(4, np.int32)
        self._chk(self.lib.scvod_batch_cluster_stats(self.h, out.ctypes.data_as(C.c_void_p)))
        r2 = np.zeros(2, np.int32)
        self._chk(self.lib.scvod_batch_cluster_rule_stats(self.h, r2.ctypes.data_as(C.c_void_p)))
        h2 = np.zeros(2, np.int32)
        self._chk(self.lib.scvod_batch_cluster_help_stats(self.h, h2.ctypes.data_as(C.c_void_p)))
        return dict(scans_approximated=int(out[0]), nodes_concerned=int(out[1]), exact=bool(out[2]), scans_on_hbm_forest=int(out[3]),
                    runs_settled_by_rule=int(r2[0]), runs_clustered_again=int(r2[1]), scans_that_shared_their_rounds=int(h2[0]), chunks_taken_by_helpers=int(h2[1]))

    def set_max_name_literal(self, literal=True):
        """ssc.cpp:354 keeps the LAST USED running number in Frame::max_name; False = fresh numbers (rounds 1-3)"""
        self._chk(self.lib.scvod_set_max_name_literal(self.h, 1 if literal else 0))

    def batch_cluster_last_name(self, n_scans):
        """per scan {name of the cluster that still carries Frame::max_name or -1, voxel slot, status, events}, and the batch's counters"""
        out = np.zeros((max(n_scans, 1), 4), np.int32)
        st = np.zeros(4, np.int32)
        rc = self.lib.scvod_batch_cluster_last_name(self.h, out.ctypes.data_as(C.c_void_p), n_scans, st.ctypes.data_as(C.c_void_p))
        if rc < 0:
            self._chk(rc)
        return out[:n_scans], dict(unknown_too_large=int(st[0]), unknown_irregular=int(st[1]))

    def set_chain_capacity(self, pool_points):
        self._chk(self.lib.scvod_set_chain_capacity(self.h, int(pool_points)))

    def batch_track_stats(self):
        out = np.zeros(8, np.int32)
        self._chk(self.lib.scvod_batch_track_stats(self.h, out.ctypes.data_as(C.c_void_p)))
        return dict(chain=bool(out[0]), segments=int(out[1]), verified=int(out[2]), rewalked=int(out[3]), error_bits=int(out[4]),
                    segment_steps=int(out[5]), warmup_steps=int(out[6]), max_name_undetermined=int(out[7]))

    def batch_track_tables(self, stream=None):
        self._chk(self.lib.scvod_batch_track_tables(self.h, C.c_void_p(stream or 0)))

    def batch_export_table(self, s, d_out, stream=None):
        """d_out: torch int32 device tensor [cap_records, 4]"""
        self._chk(self.lib.scvod_batch_export_table(self.h, int(s), C.c_void_p(d_out.data_ptr()), int(d_out.shape[0]),
                                                    C.c_void_p(stream or 0)))

    def set_timing(self, on):
        self._chk(self.lib.scvod_set_timing(self.h, int(bool(on))))

    def timings(self, cap=64):
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = self.lib.scvod_batch_timings(self.h, names, ms, cap)
        return [(names[i].decode(), float(ms[i])) for i in range(max(n, 0))]

    def arena_bytes(self):
        return int(self.lib.scvod_arena_bytes(self.h))

    def chain_workspace_bytes(self):
        """the tracking chain's walker workspace of the last job (not part of arena_bytes)"""
        return int(self.lib.scvod_chain_workspace_bytes(self.h))

    def nn_search_device(self, d_map_xyz, d_query_xyz, radius, stream=None):
        """torch CUDA tensors [n, 3] float32 in, (idx int32, sqdist float32, within uint8) CUDA tensors out."""
        import torch
        nq = int(d_query_xyz.shape[0])
        idx = torch.empty(max(nq, 1), dtype=torch.int32, device=d_query_xyz.device)
        sq = torch.empty(max(nq, 1), dtype=torch.float32, device=d_query_xyz.device)
        w = torch.empty(max(nq, 1), dtype=torch.uint8, device=d_query_xyz.device)
        self._chk(self.lib.scvod_nn_search_device(self.h, C.c_void_p(d_map_xyz.data_ptr()), int(d_map_xyz.shape[0]),
                                                  C.c_void_p(d_query_xyz.data_ptr()), nq, float(radius), C.c_void_p(idx.data_ptr()),
                                                  C.c_void_p(sq.data_ptr()), C.c_void_p(w.data_ptr()),
                                                  C.c_void_p(stream) if stream else None))
        return idx[:nq], sq[:nq], w[:nq]

    def voxelgrid(self, xyzi, leaf=(0.08, 0.08, 0.08), labels=None, max_intensity=1.0):
        """SSC::getCloud label filter + pcl::VoxelGrid of one host scan (ssc.cpp:1063-1076, 1103-1106)."""
        x, px = self._f32(xyzi)
        n = x.shape[0]
        lf = np.asarray(leaf, np.float32)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint32)
        out = np.zeros((max(n, 1), 4), np.float32)
        n_out = C.c_int32(0)
        self._chk(self.lib.scvod_voxelgrid(self.h, px, None if lab is None else lab.ctypes.data_as(C.c_void_p), n,
                                           lf.ctypes.data_as(C.c_void_p), float(max_intensity), out.ctypes.data_as(C.c_void_p),
                                           out.shape[0], C.byref(n_out)))
        return out[:n_out.value]

    def batch_voxelgrid(self, d_xyzi, offsets, d_out, leaf=(0.08, 0.08, 0.08), d_labels=None, max_intensity=1.0, stream=None):
        """Device-resident form: d_xyzi / d_out / d_labels are torch CUDA tensors; returns the output offsets."""
        offs = np.ascontiguousarray(offsets, np.int32)
        lf = np.asarray(leaf, np.float32)
        out_off = np.zeros(len(offs), np.int32)
        self._chk(self.lib.scvod_batch_voxelgrid(self.h, C.c_void_p(d_xyzi.data_ptr()),
                                                 None if d_labels is None else C.c_void_p(d_labels.data_ptr()),
                                                 offs.ctypes.data_as(C.c_void_p), len(offs) - 1, lf.ctypes.data_as(C.c_void_p),
                                                 float(max_intensity), C.c_void_p(d_out.data_ptr()), int(d_out.shape[0]),
                                                 out_off.ctypes.data_as(C.c_void_p), C.c_void_p(stream) if stream else None))
        return out_off

    def nn_search(self, map_xyz, query_xyz, radius):
        m, pm = self._f32(map_xyz)
        q, pq = self._f32(query_xyz)
        nq = q.shape[0]
        idx = np.zeros(max(nq, 1), np.int32)
        sq = np.zeros(max(nq, 1), np.float32)
        w = np.zeros(max(nq, 1), np.uint8)
        self._chk(self.lib.scvod_nn_search(self.h, pm, m.shape[0], pq, nq, float(radius), idx.ctypes.data_as(C.c_void_p),
                                           sq.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)))
        return idx[:nq], sq[:nq], w[:nq]


    def nn_radius_search(self, map_xyz, query_xyz, radius):
        """nearest map point strictly inside `radius` per query: (idx or -1, squared distance or +inf, found)"""
        m, pm = self._f32(map_xyz)
        q, pq = self._f32(query_xyz)
        nq = q.shape[0]
        idx = np.zeros(max(nq, 1), np.int32)
        sq = np.zeros(max(nq, 1), np.float32)
        self._chk(self.lib.scvod_nn_radius_search(self.h, pm, m.shape[0], pq, nq, float(radius), idx.ctypes.data_as(C.c_void_p),
                                                  sq.ctypes.data_as(C.c_void_p)))
        return idx[:nq], sq[:nq], (idx[:nq] >= 0).astype(np.uint8)


MAP_NO_GROUND, MAP_NO_REJECTED, MAP_IGNORE_DYNAMIC = 1, 2, 4


class StaticMap:
    """World-frame static map (include/scvod.h, scvod_map_*): device-resident set of occupied cells, mergeable across shards."""

    def __init__(self, capacity_cells, leaf=0.2, device=0):
        self.lib = load_lib()
        h = C.c_void_p()
        rc = self.lib.scvod_map_create(int(device), int(capacity_cells), float(leaf), C.byref(h))
        if rc != 0:
            raise ScvodError(f"scvod_map_create failed with status {rc}")
        self.h = h
        self.leaf = float(leaf)
        self.device = int(device)

    def _chk(self, rc):
        if rc != 0:
            raise ScvodError(f"status {rc}: {self.lib.scvod_map_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.scvod_map_destroy(self.h)
            self.h = None

    def clear(self, stream=None):
        self._chk(self.lib.scvod_map_clear(self.h, C.c_void_p(stream or 0)))

    def accumulate(self, ctx, poses, flags=0, stream=None):
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        assert p.shape[0] == ctx._n_scans
        self._chk(self.lib.scvod_batch_map_accumulate(ctx.h, self.h, p.ctypes.data_as(C.c_void_p), int(flags), C.c_void_p(stream or 0)))

    def accumulate_range(self, ctx, poses, first, count, flags=0, stream=None):
        """scans [first, first + count) of the batch only: a shard's own block without its halo"""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        assert p.shape[0] == ctx._n_scans
        self._chk(self.lib.scvod_batch_map_accumulate_range(ctx.h, self.h, p.ctypes.data_as(C.c_void_p), int(flags), int(first), int(count), C.c_void_p(stream or 0)))

    def count(self, stream=None):
        n = C.c_int64()
        self._chk(self.lib.scvod_map_export(self.h, None, 0, C.byref(n), C.c_void_p(stream or 0)))
        return int(n.value)

    def export(self, d_records=None, stream=None):
        """records as a torch int64 device tensor [n, 2] (cell key, packed point); unspecified order"""
        import torch
        if d_records is None:
            d_records = torch.empty((max(self.count(stream), 1), 2), dtype=torch.int64, device=torch.device("cuda", self.device))
        n = C.c_int64()
        self._chk(self.lib.scvod_map_export(self.h, C.c_void_p(d_records.data_ptr()), int(d_records.shape[0]), C.byref(n), C.c_void_p(stream or 0)))
        return d_records[:int(n.value)]

    def export_parts(self, n_parts, stream=None):
        """(records [n, 2] int64 grouped by owner shard, counts per shard): the send side of the map's reduce-scatter"""
        import torch
        rec = torch.empty((max(self.count(stream), 1), 2), dtype=torch.int64, device=torch.device("cuda", self.device))
        cnt = (C.c_int64 * int(n_parts))()
        self._chk(self.lib.scvod_map_export_parts(self.h, int(n_parts), C.c_void_p(rec.data_ptr()), int(rec.shape[0]), cnt, C.c_void_p(stream or 0)))
        counts = [int(v) for v in cnt]
        return rec[:sum(counts)], counts

    def export_parts_padded(self, n_parts, d_records, d_counts=None, stream=None):
        """d_records: torch int64 device tensor [n_parts, cap, 2], filled group by group, padded with key -1; d_counts: int64
        device tensor [n_parts] or None.  No host synchronisation (the timed multi-GPU step)."""
        assert d_records.dim() == 3 and d_records.shape[0] == n_parts and d_records.shape[2] == 2 and d_records.is_contiguous()
        self._chk(self.lib.scvod_map_export_parts_padded(self.h, int(n_parts), C.c_void_p(d_records.data_ptr()), int(d_records.shape[1]),
                                                         C.c_void_p(d_counts.data_ptr()) if d_counts is not None else None, C.c_void_p(stream or 0)))

    def merge(self, d_records, stream=None):
        self._chk(self.lib.scvod_map_merge(self.h, C.c_void_p(d_records.data_ptr()), int(d_records.numel() // 2), C.c_void_p(stream or 0)))

    def points(self, stream=None):
        """(xyzi float32 [n, 4], records int64 [n, 2]) device tensors, same (unspecified) order"""
        import torch
        n0 = max(self.count(stream), 1)
        dev = torch.device("cuda", self.device)
        xyzi = torch.empty((n0, 4), dtype=torch.float32, device=dev)
        rec = torch.empty((n0, 2), dtype=torch.int64, device=dev)
        n = C.c_int64()
        self._chk(self.lib.scvod_map_points(self.h, C.c_void_p(xyzi.data_ptr()), C.c_void_p(rec.data_ptr()), n0, C.byref(n), C.c_void_p(stream or 0)))
        return xyzi[:int(n.value)], rec[:int(n.value)]


def pose_matrix(pose):
    p = np.ascontiguousarray(pose, np.float32)
    t = np.zeros(12, np.float32)
    load_lib().scvod_pose_matrix(p.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    return t
