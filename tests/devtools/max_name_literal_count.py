"""How many per-point labels does the reference's literal `max_name` (ssc.cpp:354 stores the LAST USED running number, so the
first split-off / fused cluster of every frame re-uses it: ssc.cpp:1357, 1401, no-op insert at 1372 / 1419) move, against a
chain that hands out fresh numbers?  Device segmentation, oracle chains.  Run on the GPU box:
    python tests/devtools/max_name_literal_count.py [K64|PARK|OS128 ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (under tests/: the oracle is test infrastructure)
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
import scvod_py
import synth
import torch

CASES = {"K64": ("semantickitti", 5, 120, 300), "PARK": ("parkinglot", 1, 200, 30), "OS128": ("os128_fine", 5, 50, 700)}


def run(kind):
    preset, skip, count, first = CASES[kind]
    orc = oracle_py.load()
    P = scvod_py.make_params(preset)
    scans = [synth.make_scan(5, first + k * skip, kind, device="cuda") for k in range(count)]
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
    ctx.close()
    collide, infos = [], []
    for s in range(count):
        c, info = orc.cluster_last_name(P, res[s]["apri"])
        if c >= 0:
            assert names[s][c] == c, "the oracle's partition and the device's disagree"
        collide.append(c)
        comp = int((names[s] == names[s][info[1]]).sum()) if info[1] >= 0 else 0
        infos.append(list(map(int, info)) + [comp, int(c >= 0 and types[s][c] != -1), int(c >= 0 and types[s][c] == 2)])
    infos = np.asarray(infos)
    apri = np.concatenate([r["apri"] for r in res])
    ao = np.concatenate([[0], np.cumsum([r["n_apri"] for r in res])]).astype(np.int32)
    nm, ty = np.concatenate(names), np.concatenate(types)
    dyn3, nd3 = orc.sequence_tracking(P, apri, ao, nm, ty, poses, chain=3)
    dynL, ndL, st = orc.sequence_tracking_literal(P, apri, ao, nm, ty, collide, poses, chain=3)
    dynL1, ndL1, _ = orc.sequence_tracking_literal(P, apri, ao, nm, ty, collide, poses, chain=1)
    out = dict(kind=kind, frames=count, points=int(len(dyn3)),
               K_alive_frames=int((np.asarray(collide) >= 0).sum()),
               K_alive_after_box_refine=int(infos[:, 7].sum()), K_is_car=int(infos[:, 8].sum()),
               openers_in_K_component=dict(zip(*map(lambda a: a.tolist(), np.unique(np.minimum(infos[:, 3], 9), return_counts=True)))),
               K_component_points=dict(median=int(np.median(infos[:, 6])), p90=int(np.percentile(infos[:, 6], 90)), max=int(infos[:, 6].max())),
               renames_of_K=int((infos[:, 4] > 0).sum()),
               label_bytes_differ=int((dyn3 != dynL).sum()), dynamic_points=[int((dyn3 == 1).sum()), int((dynL == 1).sum())],
               dynamic_clusters=[nd3, ndL], splits_dropped=int(st[0]), fuses_dropped=int(st[1]), car_clusters_lost=int(st[2]),
               points_lost=int(st[3]), literal_container_order_differs=int((dynL1 != dynL).sum()))
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    kinds = sys.argv[1:] or ["K64", "PARK", "OS128"]
    rows = [run(k) for k in kinds]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "max_name_literal_count.json"), "w"), indent=1)
