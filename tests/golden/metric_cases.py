"""Seeded gt / estimate clouds of the metric golden vectors (data generator shared by
make_metric_golden.py, which runs the reference's tool/analysis.py on them, and by the tests)."""
import numpy as np

CASES = [dict(seed=1, n_gt=4000, frac_dyn=0.1, keep_static=0.97, keep_dyn=0.05, jitter=0.01),
         dict(seed=2, n_gt=6000, frac_dyn=0.2, keep_static=0.9, keep_dyn=0.3, jitter=0.05),
         dict(seed=3, n_gt=3000, frac_dyn=0.05, keep_static=1.0, keep_dyn=0.0, jitter=0.0)]


def make_case(seed, n_gt, frac_dyn, keep_static, keep_dyn, jitter):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-30, 30, (n_gt, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-2, 3, n_gt)
    lab = np.where(rng.random(n_gt) < frac_dyn, rng.choice([252, 253, 255, 259], n_gt), rng.choice([40, 50, 70, 10], n_gt)).astype(np.uint32)
    inst = rng.integers(0, 30, n_gt).astype(np.uint32)
    lab_full = lab | (inst << 16) * (lab >= 252)
    is_dyn = lab >= 252
    keep = np.where(is_dyn, rng.random(n_gt) < keep_dyn, rng.random(n_gt) < keep_static)
    est_xyz = (xyz[keep] + rng.normal(0, jitter, (int(keep.sum()), 3))).astype(np.float32)
    est_lab = lab_full[keep]
    return xyz, lab_full, est_xyz, est_lab
