import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dr-using-scv-od_amd', 'pyshim'))
import scvod_py as scvod, synth
pts, offs, _, _ = synth.make_batch(5, 0, 256, "K64", device="cuda")
ctx = scvod.Ctx(scvod.make_params("semantickitti"), max_points_total=int(offs[-1]) + 64, max_scans=256)
out = torch.empty_like(pts)
for i in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    oo = ctx.batch_voxelgrid(pts, offs, out)
    torch.cuda.synchronize(); print("ms", 1e3*(time.perf_counter()-t), oo[-1]/offs[-1])
