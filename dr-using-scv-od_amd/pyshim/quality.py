"""Dynamic-removal quality of the device path next to the oracle chain on the same labelled sample (bench.py `quality`,
tests/test_gpu_facade.py): preservation rate PR and rejection rate RR as the reference's tool/analysis.py defines them
(analysis.py:186-187; 1-NN inlier radius voxelsize * sqrt(3) / 2, analysis.py:133), ERASOR protocol: the ground-truth
map is every point of the sample in the world frame with its label, an estimate is the subset of those points a method
keeps (everything it did not mark dynamic and that Patchwork did not drop), carrying the same labels."""
import numpy as np

import metric

DROPPED = 3  # per input point: 0 static, 1 dynamic, 2 in no cluster (kept), 3 dropped by Patchwork (in neither cloud)


def device_point_labels(ctx, s, n_points):
    """per INPUT point of scan s of the ctx's last tracked batch: the oracle's label convention"""
    r = ctx.batch_fetch(s)
    t = ctx.batch_fetch_track(s)
    lab = np.zeros(n_points, np.uint8)
    lab[r["cls"] == 2] = DROPPED
    lab[r["apri_src"]] = t["pt_dyn"]
    return lab


def world_points(scvod_py, x, offs, poses):
    out = np.empty((int(offs[-1]), 3), np.float32)
    for s in range(len(offs) - 1):
        T = scvod_py.pose_matrix(poses[s])
        p = x[offs[s]:offs[s + 1]]
        for i in range(3):
            out[offs[s]:offs[s + 1], i] = ((T[4 * i] * p[:, 0] + T[4 * i + 1] * p[:, 1]) + T[4 * i + 2] * p[:, 2]) + T[4 * i + 3]
    return out


def compare(scvod_py, ctx, x, offs, poses, gt_label, ref_label, dev_label=None, voxelsize=0.2):
    """x [n, 4] points of the first len(offs) - 1 scans of the ctx's last tracked batch, gt_label their ground-truth
    semantic labels, ref_label the oracle chain's per-point labels.  The sample's last scan has no successor in the oracle
    run, so the comparison covers the scans before it."""
    ns = len(offs) - 2
    if ns < 1:
        return None
    n = int(offs[ns])
    if dev_label is None:
        dev_label = np.concatenate([device_point_labels(ctx, s, int(offs[s + 1] - offs[s])) for s in range(ns)])
    dev_label = np.asarray(dev_label)[:n]
    w = world_points(scvod_py, x[:n], offs[: ns + 1], poses[:ns])
    gt = np.asarray(gt_label[:n])
    ref = np.asarray(ref_label[:n])
    out = {"scans": ns, "points": n, "voxelsize": voxelsize, "definition": "tool/analysis.py:186-187 (PR = preserved static / gt static, RR = 1 - preserved dynamic / gt dynamic)"}
    res = {}
    for name, lab in (("device", dev_label), ("reference_chain", ref)):
        keep = (lab != 1) & (lab != DROPPED)
        m = metric.preservation_rejection(w, gt, w[keep], gt[keep], ctx.nn_radius_search, voxelsize)  # inlier radius 0.866 voxelsize < voxelsize
        res[name] = m
        out[name] = {"PR": m["PR"], "RR": m["RR"], "F1": m["F1"], "kept_points": int(keep.sum()), "marked_dynamic": int((lab == 1).sum())}
    out["delta_PR"] = res["device"]["PR"] - res["reference_chain"]["PR"]
    out["delta_RR"] = res["device"]["RR"] - res["reference_chain"]["RR"]
    out["labels_equal_fraction"] = float((dev_label == ref).mean())
    out["num_gt_dynamic"] = res["device"]["num_gt_dynamic"]
    return out
