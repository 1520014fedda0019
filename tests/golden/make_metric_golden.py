"""Generates tests/golden/metric_golden.json by importing the REFERENCE's tool/analysis.py (only
possible in the build container: /root/reference does not exist on the GPU box) on seeded synthetic
gt / estimate clouds.  Only data is committed: the seeded generator parameters and the numbers the
reference's functions returned.  `pypcd` is absent, so a stub exposing `.pc_data` stands in for it."""
import json
import os
import sys
import types

import numpy as np

sys.modules["pypcd"] = types.ModuleType("pypcd")
sys.path.insert(0, "/root/reference/tool")
import analysis  # noqa: E402  (the reference's metric script)
from sklearn.neighbors import NearestNeighbors  # noqa: E402


class PC:
    def __init__(self, xyz, label):
        self.pc_data = np.zeros(len(xyz), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("intensity", "f4")])
        self.pc_data["x"], self.pc_data["y"], self.pc_data["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        self.pc_data["intensity"] = label.astype(np.float32)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from metric_cases import CASES, make_case  # noqa: E402

out = []
for c in CASES:
    xyz, lab, exyz, elab = make_case(**c)
    gt, est = PC(xyz, lab), PC(exyz, elab)
    nbrs = NearestNeighbors(n_neighbors=1, algorithm="kd_tree").fit(analysis.data2xyz_np(est))
    dists, indices = nbrs.kneighbors(analysis.data2xyz_np(gt))
    num_gt = analysis.count_static_and_dynamic(gt.pc_data["intensity"])
    num_est = analysis.count_static_and_dynamic(est.pc_data["intensity"])
    npres, nstat, ndyn = analysis.calc_naive_preservation(gt, est, dists.reshape(-1), indices.reshape(-1), 0.2)
    pr = float(nstat) / float(num_gt["static"]) * 100   # analysis.py:186
    rr = float(num_gt["dynamic"] - ndyn) / float(num_gt["dynamic"]) * 100  # analysis.py:187
    out.append(dict(case=c, num_gt_static=int(num_gt["static"]), num_gt_dynamic=int(num_gt["dynamic"]),
                    num_est_static=int(num_est["static"]), num_est_dynamic=int(num_est["dynamic"]),
                    num_preserved=int(npres), num_static_preserved=int(nstat), num_dynamic_preserved=int(ndyn),
                    PR=pr, RR=rr, F1=2 * (pr / 100) * (rr / 100) / ((pr / 100) + (rr / 100))))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metric_golden.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
