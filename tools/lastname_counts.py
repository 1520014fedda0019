"""development: how many scans of a test sample end with an undetermined max_name (run on the GPU box):
python tools/lastname_counts.py OS128 os128_fine 700 50 5"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
import scvod_py, synth, torch
kind, preset, first, count, stride = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
P = scvod_py.make_params(preset)
sc = [synth.make_scan(5, first + k * stride, kind, device="cuda")[0] for k in range(count)]
d = torch.cat(sc).contiguous()
offs = np.concatenate([[0], np.cumsum([len(s) for s in sc])]).astype(np.int32)
ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
ctx.batch_process(d, offs)
ctx.batch_cluster()
ln, st = ctx.batch_cluster_last_name(count)
print(kind, first, count, stride, st, "status histogram", np.bincount(ln[:, 2], minlength=4).tolist())
ctx.close()
bad = np.nonzero(ln[:, 2] != 0)[0]
print("undetermined scans (index in the sample: nodes of the set that would have to be followed):", {int(first + b * stride): int(ln[b, 3]) for b in bad})
