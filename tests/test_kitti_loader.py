"""KITTI-format sequence input of the facade (host/ssc.cpp getPose / getCloud / segDF; reference src/ssc.cpp:930-1125,
1428-1452): the pose maths on a hand-built 3-pose file against known answers, the oracle's restatement and a float64
evaluation (not gpu: `scvod_sequence --poses-only` needs no device); the whole driver on a KITTI-layout directory (gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dr-using-scv-od_amd", "host")

# ssc/tr_ of config/semantickitti.yaml:59-62 (velodyne -> camera of KITTI odometry)
TR = [4.276802385584e-04, -9.999672484946e-01, -8.084491683471e-03, -1.198459927713e-02,
      -7.210626507497e-03, 8.081198471645e-03, -9.999413164504e-01, -5.403984729748e-02,
      9.999738645903e-01, 4.859485810390e-04, -7.206933692422e-03, -2.921968648686e-01,
      0, 0, 0, 1]


def _yaml(path, tr, poses, start=0, end=3, skip=1):
    path.write_text(f"""common:
  skip_: {skip}
session:
  pose_path_: "{poses}"
  start_: {start}
  end_: {end}
ssc:
  tr_: [{", ".join(repr(float(v)) for v in tr[:4])},
        {", ".join(repr(float(v)) for v in tr[4:8])},
        {", ".join(repr(float(v)) for v in tr[8:12])},
        {", ".join(repr(float(v)) for v in tr[12:])}]
""")


def _exe():
    exe = os.path.join(HOST, "scvod_sequence")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST])
    return exe


def _rot(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, sy * sr + cy * sp * cr],
                     [sy * cp, cy * cr + sy * sp * sr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _euler(R):  # utility.h:488-505 in float64
    sy = np.hypot(R[0, 0], R[1, 0])
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])


def _run(tmp_path, tr, cams, **kw):
    poses = tmp_path / "poses.txt"
    poses.write_text("\n".join(" ".join(repr(float(v)) for v in c) for c in cams) + "\n")
    cfg = tmp_path / "cfg.yaml"
    _yaml(cfg, tr, poses, **kw)
    r = subprocess.run([_exe(), "--poses-only", str(cfg)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return np.array([[float(v) for v in l.split()] for l in r.stdout.strip().splitlines()])


def test_three_hand_built_poses(tmp_path, oracle):
    ident = np.eye(4).ravel().tolist()
    # identity extrinsic: the camera pose IS the pose -- identity, a pure translation, 90 deg yaw + translation
    cams = [[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0],
            [1, 0, 0, 1.5, 0, 1, 0, -2.0, 0, 0, 1, 0.25],
            [0, -1, 0, 3.0, 1, 0, 0, 4.0, 0, 0, 1, 5.0]]
    got = _run(tmp_path, ident, cams)
    want = np.array([[0, 0, 0, 0, 0, 0], [1.5, -2.0, 0.25, 0, 0, 0], [3.0, 4.0, 5.0, 0, 0, np.pi / 2]])
    assert np.allclose(got, want, atol=1e-6)
    # the KITTI extrinsic: velo_to_cam = Tr^-1 cam Tr in float64 as the reference formula, angles by utility.h:488-505
    Tr = np.array(TR, np.float64).reshape(4, 4)
    rng = np.random.default_rng(3)
    cams, want = [], []
    for k in range(3):
        M = np.eye(4)
        M[:3, :3] = _rot(*rng.uniform(-0.3, 0.3, 3))
        M[:3, 3] = rng.uniform(-50, 50, 3)
        V = np.linalg.inv(Tr) @ M @ Tr
        cams.append(M[:3].ravel().tolist())
        want.append(np.concatenate([V[:3, 3], _euler(V[:3, :3])]))
    got = _run(tmp_path, TR, cams)
    assert np.allclose(got, np.array(want), rtol=1e-5, atol=2e-5)
    # the oracle's own restatement agrees to float rounding
    lib = oracle.lib
    for cam, g in zip(cams, got):
        tr32, cam32 = np.array(TR, np.float32), np.array(cam, np.float32)
        pose, v2c = np.zeros(6, np.float32), np.zeros(16, np.float32)
        assert lib.oracle_kitti_pose(tr32.ctypes.data_as(C.c_void_p), cam32.ctypes.data_as(C.c_void_p), pose.ctypes.data_as(C.c_void_p),
                                     v2c.ctypes.data_as(C.c_void_p)) == 0
        assert np.allclose(pose, g, rtol=1e-5, atol=2e-5)


def test_start_end_skip_select_the_lines_the_reference_selects(tmp_path):
    """ssc.cpp:943-951: line `count` is used when count >= start, (count - start) % skip == 0 and count < end"""
    ident = np.eye(4).ravel().tolist()
    cams = [[1, 0, 0, float(i), 0, 1, 0, 0, 0, 0, 1, 0] for i in range(12)]
    got = _run(tmp_path, ident, cams, start=2, end=11, skip=4)
    assert np.array_equal(got[:, 0], [2.0, 6.0, 10.0])


@pytest.mark.gpu
def test_segdf_on_a_kitti_layout_directory(scvod):
    """SSC::segDF on velodyne/*.bin + labels/*.label + poses.txt: frames load in numeric file order with the label
    filter and the 0.08 m VoxelGrid applied, every frame is processed, the tracking chain runs"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sequence_demo", os.path.join(ROOT, "tools", "sequence_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    r = demo.run(seq=3, first=0, count=8, kind="PARK", preset="parkinglot", skip=2, verbose=False)
    lines = [l for l in r["log"].splitlines() if l.startswith("frame ")]
    assert [int(l.split()[1]) for l in lines] == [0, 2, 4, 6]
    assert all(int(l.split()[3]) > 5000 for l in lines)          # points after the VoxelGrid
    assert "frames 4 dynamic_total" in r["log"]
    assert r["PR"] > 98.0 and r["n_dynamic"] > 0
