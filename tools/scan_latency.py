"""Latency of the per-scan host API (scvod_process_scan: host buffers in and out), the call a 10 Hz ROS node would make."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dr-using-scv-od_amd", "pyshim"))
import torch  # noqa: F401  (loads the HIP runtime the library links against)
import scvod_py as scvod
import synth

for kind, preset in (("K64", "semantickitti"), ("PARK", "parkinglot")):
    x = synth.make_scan(5, 40, kind)[0].numpy()
    ctx = scvod.Ctx(scvod.make_params(preset), max_points_total=x.shape[0] + 64, max_scans=1)
    for _ in range(3):
        ctx.process_scan(x)
    ts = []
    for _ in range(30):
        t = time.perf_counter()
        r = ctx.process_scan(x)
        ts.append(1e3 * (time.perf_counter() - t))
    ts = np.sort(ts)
    print(f"{kind}: {x.shape[0]} points -> {r['n_apri']} apri, {r['n_voxels']} voxels; process_scan median {ts[len(ts)//2]:.3f} ms, min {ts[0]:.3f} ms")
    ctx.close()

# where the time goes at B = 1: hipEvent time of every stage of one scan
x = synth.make_scan(5, 40, "K64")[0].numpy()
ctx = scvod.Ctx(scvod.make_params("semantickitti"), max_points_total=x.shape[0] + 64, max_scans=1)
ctx.process_scan(x)
ctx.set_timing(True)
ctx.process_scan(x)
tm = ctx.timings()
print("device time per stage (us):", {k: round(1e3 * v, 1) for k, v in tm}, "sum", round(1e3 * sum(v for _, v in tm), 1))
