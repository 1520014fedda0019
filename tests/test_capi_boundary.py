"""The C-ABI library loads and exports every symbol include/scvod.h declares; host-only entry points
agree with the oracle; and the product refuses to run without a GPU (no CPU fallback).  Not gpu."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(scvod):
    lib = scvod.load_lib()
    hdr = open(os.path.join(ROOT, "include", "scvod.h")).read()
    declared = sorted(set(re.findall(r"\b(scvod_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/scvod.h but not exported by libscvod.so"
    assert sorted(scvod.EXPORTED_SYMBOLS) == declared


def test_struct_layouts(scvod):
    assert C.sizeof(scvod.Params) == 64
    assert scvod.APRI_DTYPE.itemsize == 44       # struct PointAPRI, utility.h:96-106
    assert scvod.PLANE_DTYPE.itemsize == 48
    assert C.sizeof(scvod.ScanResult) == 8 * 4 + 12 * 8


def test_defaults_and_grid_match_oracle(scvod, oracle):
    p = scvod.make_params()
    q = oracle.params_default()
    assert bytes(p) == bytes(q)
    pw, qw = scvod.PwParams(), scvod.PwParams()
    scvod.load_lib().scvod_pw_params_default(C.byref(pw))
    oracle.lib.oracle_pw_params_default(C.byref(qw))
    assert bytes(pw) == bytes(qw)
    assert list(pw.num_sectors_each_zone) == [16, 32, 54, 32] and list(pw.num_rings_each_zone) == [2, 4, 4, 4]
    for preset in scvod.PRESETS:
        P = scvod.make_params(preset)
        assert scvod.grid_dims(P) == oracle.grid_dims(P)


def test_pose_delta_host_matches_oracle(scvod, oracle):
    rng = np.random.default_rng(0)
    lib = scvod.load_lib()
    for _ in range(200):
        a = rng.uniform(-1, 1, 6).astype(np.float32) * np.array([50, 50, 2, 0.1, 0.1, 3.1], np.float32)
        b = a + rng.uniform(-1, 1, 6).astype(np.float32) * np.array([2, 2, 0.1, 0.02, 0.02, 0.1], np.float32)
        T = np.zeros(12, np.float32)
        lib.scvod_pose_delta(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
        assert np.array_equal(T.view(np.uint32), oracle.pose_delta(a, b).view(np.uint32))


def test_yaml_surface(scvod, tmp_path):
    y = tmp_path / "cfg.yaml"
    y.write_text("common:\n  skip_: 5\nssc:\n  sensor_height_: 1.73\n  min_dis_:  1.5  # /m\n  max_dis_: 30.0\n"
                 "  min_azimuth_: -40.0\n  max_azimuth_: 80.0\n  range_res_: 0.4\n  occupancy_: 0.4\n  toBeClass_: 10\n")
    P = scvod.params_from_yaml(str(y))
    assert (P.sensor_height, P.min_dis, P.max_dis) == (np.float32(1.73), 1.5, 30.0)
    assert P.sector_res == np.float32(1.2) and P.azimuth_res == 2.0   # defaults kept
    assert scvod.grid_dims(P) == (72, 300, 60, 1296000)


def test_no_cpu_fallback(scvod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the refusal path is exercised on CPU-only hosts")
    with pytest.raises(scvod.ScvodError) as e:
        scvod.Ctx(scvod.make_params("semantickitti"), max_points_total=1000)
    assert "-2" in str(e.value)


def test_product_does_not_reference_the_oracle():
    bad = []
    for base in ("dr-using-scv-od_amd", "include", "tools"):  # (development tools that need the checker live under tests/devtools)
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".so", ".pyc", ".o")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle[/_.]|liboracle|oracle_py", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
