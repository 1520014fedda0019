"""latency of k_cc_lastname per scan (each scan as a batch of its own); run on the GPU box"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scvod_py, synth, torch
kind, preset, first, stride, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
P = scvod_py.make_params(preset)
rows = []
for k in range(count):
    pts, _, _ = synth.make_scan(5, first + k * stride, kind, device="cuda")
    offs = np.asarray([0, len(pts)], np.int32)
    ctx = scvod_py.Ctx(P, max_points_total=len(pts) + 64, max_scans=1)
    ctx.set_timing(True)
    ctx.batch_process(pts.contiguous(), offs)
    for rep in range(3):
        ctx.batch_cluster(); torch.cuda.synchronize()
    t = dict(ctx.timings())
    ln, st = ctx.batch_cluster_last_name(1)
    rows.append((round(t["cc_lastname"], 3), round(t["cc_scan"], 3), ln[0].tolist()))
    ctx.close()
rows.sort()
for r in rows: print(r)
