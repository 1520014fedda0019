"""per-scan outcome of csrc/scvod_lastname.hip next to the oracle's literal loop (run on the GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (under tests/: the oracle is test infrastructure)
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py, scvod_py, synth, torch
kind, preset, first, stride, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
orc = oracle_py.load(); P = scvod_py.make_params(preset)
scans = [synth.make_scan(5, first + k * stride, kind, device="cuda") for k in range(count)]
d = torch.cat([sc[0] for sc in scans]).contiguous()
offs = np.concatenate([[0], np.cumsum([len(sc[0]) for sc in scans])]).astype(np.int32)
ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=count)
ctx.set_timing(True) if hasattr(ctx, "set_timing") else None
ctx.batch_process(d, offs)
for rep in range(2):
    ctx.batch_cluster()
    torch.cuda.synchronize()
print([(k, round(v, 3)) for k, v in ctx.timings()])
ln, st = ctx.batch_cluster_last_name(count)
print(st, "events mean", ln[:, 3].mean(), "max", ln[:, 3].max())
bad = 0
for s in range(count):
    if ln[s, 2] == 0 and "-v" not in sys.argv: continue
    r = ctx.batch_fetch(s); names = ctx.batch_fetch_clusters(s, r["n_apri"])
    want, info = orc.cluster_last_name(P, r["apri"])
    comp = names[info[1]]
    cv = len(np.unique(r["apri"]["voxel_idx"][names == comp]))
    L = names.max()
    print(s, "status", ln[s], "want", want, "last opener", info[1], "of n", r["n_apri"], "its cluster", comp, "voxels", cv, "openers in it", info[3], "latest-born cluster", L,
          "voxels", len(np.unique(r["apri"]["voxel_idx"][names == L])), "nv", r["n_voxels"])
