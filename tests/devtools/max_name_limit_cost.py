#!/usr/bin/env python3
"""What the one counted limit costs: the 20 scans of the 1000-scan OS128 bench job whose Frame::max_name stays undetermined (the chain hands out
a fresh number there) -- per-point dynamic bytes of the device chain against the oracle's literal chain that KNOWS which cluster carries the
number, over a window of consecutive scans around every group of them.  usage (GPU box): python tests/devtools/max_name_limit_cost.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
import scvod_py
import synth
import torch

orc = oracle_py.load()
P = scvod_py.make_params("os128_fine")
total = differ_all = und_all = 0
for lo, hi in ((115, 131), (216, 236), (290, 304), (489, 503), (671, 685), (726, 742), (855, 869), (945, 965)):
    count = hi - lo
    scans = [synth.make_scan(5, lo + k, "OS128", device="cuda") for k in range(count)]
    d = torch.cat([sc[0] for sc in scans]).contiguous()
    offs = np.concatenate([[0], np.cumsum([len(sc[0]) for sc in scans])]).astype(np.int32)
    poses = np.asarray([sc[2] for sc in scans], np.float32)
    ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    res = [ctx.batch_fetch(s) for s in range(count)]
    names = [ctx.batch_fetch_clusters(s, res[s]["n_apri"]) for s in range(count)]
    types = [ctx.batch_fetch_cluster_types(s, res[s]["n_apri"], car_label=2, other_label=1) for s in range(count)]
    ln, st = ctx.batch_cluster_last_name(count)
    unknown = ln[:, 2] != 0
    T = np.zeros((count, 12), np.float32)
    for s in range(count - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    got = np.concatenate([ctx.batch_fetch_track(s)["pt_dyn"] for s in range(count)])
    fresh, _ = orc.reference_chain(P, res, names, types, poses, unknown=unknown)
    literal, _ = orc.reference_chain(P, res, names, types, poses)
    assert np.array_equal(got, fresh)
    n_d = int((got != literal).sum())
    print(f"scans {lo}..{hi - 1}: undetermined {[lo + int(s) for s in np.nonzero(unknown)[0]]} (sets of {[int(ln[s, 3]) for s in np.nonzero(unknown)[0]]} voxels): "
          f"{n_d} of {len(got)} per-point bytes differ from the literal chain; dynamic points {int(got.sum())}", flush=True)
    total += len(got)
    differ_all += n_d
    und_all += int(unknown.sum())
    ctx.close()
print(f"all windows: {und_all} undetermined scans, {differ_all} of {total} per-point bytes differ from the reference's own reading")
