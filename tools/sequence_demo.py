#!/usr/bin/env python3
"""End-to-end demo on a labelled synthetic sequence written in KITTI layout (velodyne/*.bin, labels/*.label, poses.txt):
the C++ facade's SSC::segDF (dr-using-scv-od_amd/host/scvod_sequence: getPose / getCloud -> per frame process -> GPU
clustering + bounding-box rules -> tracking chain) removes the clusters it found dynamic; preservation / rejection rates are then computed with the
definition of the reference's tool/analysis.py (PR = kept static / all static, RR = removed dynamic / all
dynamic; the estimate is an exact subset of the ground-truth cloud, so no NN search is needed).

Not a parity claim against the reference's published PR/RR: segmentGpu() has no intensity merge and no
region growing, and the scene is synthetic.  It shows the hot path + the "next" rows working as a pipeline."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))

YAML = """common:
  skip_: {skip}
  is_pcd_: false
session:
  data_path_: "{data}"
  label_path_: "{labels}"
  pose_path_: "{poses}"
  start_: 0
  end_: {count}
ssc:
  sensor_height_: {sensor_height}
  min_dis_: {min_dis}
  max_dis_: {max_dis}
  min_angle_: 0.0
  max_angle_: 360.0
  min_azimuth_: {min_azimuth}
  max_azimuth_: {max_azimuth}
  range_res_: {range_res}
  sector_res_: {sector_res}
  azimuth_res_: {azimuth_res}
  occupancy_: {occupancy}
  max_z_: {max_z}
  min_z_: {min_z}
  car_square_: {car_square}
  toBeClass_: {toBeClass}
  max_intensity_: 255.0
  building_: 0
  tree_: 1
  car_: 2
  tr_: [1, 0, 0, 0,
        0, 1, 0, 0,
        0, 0, 1, 0,
        0, 0, 0, 1]
"""


def write_kitti_sequence(d, seq, first, count, kind):
    """velodyne/%06d.bin (x y z intensity in [0, 1]), labels/%06d.label (uint32), poses.txt (3x4 camera poses; with an
    identity Tr the camera frame IS the velodyne frame) -- the layout SSC::getCloud / getPose read (ssc.cpp:930-1125)"""
    import scvod_py
    import synth
    os.makedirs(os.path.join(d, "velodyne"))
    os.makedirs(os.path.join(d, "labels"))
    scans, labels = [], []
    with open(os.path.join(d, "poses.txt"), "w") as pf:
        for k in range(count):
            pts, lab, pose = synth.make_scan(seq, first + k, kind)
            x = pts.numpy().copy()
            scans.append(x.copy())
            labels.append(lab.numpy())
            x[:, 3] /= np.float32(255.0)
            x.tofile(os.path.join(d, "velodyne", f"{k:06d}.bin"))
            lab.numpy().astype(np.uint32).tofile(os.path.join(d, "labels", f"{k:06d}.label"))
            T = scvod_py.pose_matrix(pose)
            pf.write(" ".join(repr(float(v)) for v in T) + "\n")
    return scans, labels


def run(seq=5, first=0, count=12, kind="K64", preset="semantickitti", skip=1, verbose=True):
    import scvod_py
    from scipy.spatial import cKDTree
    exe = os.path.join(ROOT, "dr-using-scv-od_amd", "host", "scvod_sequence")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    with tempfile.TemporaryDirectory() as d:
        scans, labels = write_kitti_sequence(d, seq, first, count, kind)
        cfg = os.path.join(d, "cfg.yaml")
        open(cfg, "w").write(YAML.format(skip=skip, count=count, data=os.path.join(d, "velodyne"), labels=os.path.join(d, "labels"),
                                         poses=os.path.join(d, "poses.txt"), **scvod_py.PRESETS[preset]))
        out = os.path.join(d, "out")
        os.makedirs(out)
        res = subprocess.run([exe, cfg, out], capture_output=True, text=True, timeout=900)
        if res.returncode != 0:
            raise RuntimeError(res.stderr)
        n_static = n_dynamic = kept_static = kept_dynamic = 0
        frames = list(range(0, count, skip))
        for k in frames[:-1]:  # the last frame is never the `pre` of a tracking call
            # the frame the facade processed: the scan after the label filter + VoxelGrid 0.08 m (one centroid per cell);
            # a centroid takes the label of the nearest raw point
            cloud = np.fromfile(os.path.join(out, f"{k}_cloud.f32"), np.float32).reshape(-1, 4)
            dyn = np.fromfile(os.path.join(out, f"{k}_dynamic.f32"), np.float32).reshape(-1, 4)
            lab = labels[k][cKDTree(scans[k][:, :3]).query(cloud[:, :3])[1]]
            key = lambda a: a[:, :3].copy().view([("", np.float32)] * 3).ravel()
            removed = np.isin(key(cloud), key(dyn))
            is_dyn = lab >= 252
            n_static += int((~is_dyn).sum())
            n_dynamic += int(is_dyn.sum())
            kept_static += int((~is_dyn & ~removed).sum())
            kept_dynamic += int((is_dyn & ~removed).sum())
        pr = 100.0 * kept_static / max(n_static, 1)
        rr = 100.0 * (n_dynamic - kept_dynamic) / max(n_dynamic, 1)
        f1 = 2 * (pr / 100) * (rr / 100) / max((pr / 100) + (rr / 100), 1e-12)
        if verbose:
            print(res.stdout.strip().splitlines()[-1])
            print(f"frames {len(frames)}  static pts {n_static}  dynamic pts {n_dynamic}  PR {pr:.2f} %  RR {rr:.2f} %  F1 {f1:.4f}")
        return dict(PR=pr, RR=rr, F1=f1, n_static=n_static, n_dynamic=n_dynamic, log=res.stdout)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=5)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=12)
    ap.add_argument("--kind", default="K64")
    ap.add_argument("--preset", default="semantickitti")
    ap.add_argument("--skip", type=int, default=1)
    a = ap.parse_args()
    run(a.seq, a.first, a.count, a.kind, a.preset, a.skip)
