"""Static-map quality metric of the reference's tool/analysis.py (ERASOR-style), host side.

PR = preserved static points / gt static points, RR = 1 - preserved dynamic points / gt dynamic
points (analysis.py:186-187), where a gt point is "preserved" when its nearest estimate point lies
within voxelsize*sqrt(3)/2 (analysis.py:133) and both carry a static (resp. dynamic) label
(analysis.py:147-152).  The 1-NN look-up is supplied by the caller: the GPU correspondence kernel
(Ctx.nn_search) in the product, the brute-force oracle in the CPU tests."""
import numpy as np

DYNAMIC_CLASSES = (252, 253, 254, 255, 256, 257, 258, 259)  # analysis.py:6, config ssc/dynamic_label_


def _sem(label):
    return label.astype(np.uint32) & 0xFFFF  # analysis.py:8-12


def preservation_rejection(gt_xyz, gt_label, est_xyz, est_label, nn_fn, voxelsize=0.2):
    gt_sem, est_sem = _sem(np.asarray(gt_label)), _sem(np.asarray(est_label))
    gt_dyn = np.isin(gt_sem, DYNAMIC_CLASSES)
    est_dyn = np.isin(est_sem, DYNAMIC_CLASSES)
    idx, sqd, _ = nn_fn(np.asarray(est_xyz, np.float32), np.asarray(gt_xyz, np.float32), voxelsize)
    dist = np.sqrt(sqd.astype(np.float64))
    inl = dist < voxelsize * np.sqrt(3) / 2
    est_dyn_at = est_dyn[np.maximum(idx, 0)]  # idx -1 (a radius-bounded search found nothing) is never an inlier
    num_static_preserved = int(np.count_nonzero(inl & ~gt_dyn & ~est_dyn_at))
    num_dynamic_preserved = int(np.count_nonzero(inl & gt_dyn & est_dyn_at))
    n_static, n_dynamic = int((~gt_dyn).sum()), int(gt_dyn.sum())
    pr = 100.0 * num_static_preserved / n_static
    rr = 100.0 * (n_dynamic - num_dynamic_preserved) / n_dynamic
    f1 = 2 * (pr / 100) * (rr / 100) / ((pr / 100) + (rr / 100)) if pr + rr > 0 else 0.0
    return dict(num_gt_static=n_static, num_gt_dynamic=n_dynamic, num_est_static=int((~est_dyn).sum()),
                num_est_dynamic=int(est_dyn.sum()), num_preserved=int(np.count_nonzero(inl)),
                num_static_preserved=num_static_preserved, num_dynamic_preserved=num_dynamic_preserved, PR=pr, RR=rr,
                F1=f1)


# ---- evaluate() of src/evaluate.cpp:79-145: the colour classes of the reference's map viewer --------------------
TP_STATIC, FN_STATIC, TN_DYNAMIC, FN_DYNAMIC, UNMATCHED = 1, 2, 3, 4, 0


def classify_map_points(original_xyz, predicted_static, static_xyz, dynamic_xyz, nn_fn):
    """Per point of the original map: predicted static (g != 0 in the reference's RGB encoding) -> TP_STATIC when a
    ground-truth static point lies within 0.15 m, else FN_STATIC (orange) when a ground-truth dynamic point lies within
    0.10 m; predicted dynamic -> TN_DYNAMIC when a dynamic point lies within 0.15 m, else FN_DYNAMIC (pink) when a
    static point lies within 0.10 m; UNMATCHED otherwise (not drawn).  pcl::KdTreeFLANN::radiusSearch is non-empty iff
    the nearest neighbour's squared distance is below radius^2 (FLANN's RadiusResultSet keeps dist < r^2; parity
    unpinned), so four 1-NN searches of the A7 kernel answer it."""
    o = np.asarray(original_xyz, np.float32).reshape(-1, 3)
    ps = np.asarray(predicted_static, bool)
    out = np.zeros(len(o), np.int32)

    def near(cloud, radius):
        c = np.asarray(cloud, np.float32).reshape(-1, 3)
        if len(c) == 0:
            return np.zeros(len(o), bool)
        _, sq, _ = nn_fn(c, o, radius)
        return sq < np.float32(radius) * np.float32(radius)

    s15, s10, d15, d10 = near(static_xyz, 0.15), near(static_xyz, 0.1), near(dynamic_xyz, 0.15), near(dynamic_xyz, 0.1)
    out[ps & s15] = TP_STATIC
    out[ps & ~s15 & d10] = FN_STATIC
    out[~ps & d15] = TN_DYNAMIC
    out[~ps & ~d15 & s10] = FN_DYNAMIC
    return out
