#!/usr/bin/env python3
"""The shared rounds of the exact re-clustering (k_cc_scan's helper blocks, round 6) against the same rounds run by every scan's
workgroup alone: same partition point for point, and what a batch costs in each mode.
usage: python tools/cluster_help_check.py [--scans 1000] [--first 0] [--stride 1] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="OS128")
    ap.add_argument("--preset", default="os128_fine")
    ap.add_argument("--scans", type=int, default=1000)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--modes", default="0,3,1")
    a = ap.parse_args()
    import torch
    import scvod_py
    import synth
    P = scvod_py.make_params(a.preset)
    parts, offs = [], [0]
    for k in range(a.scans):
        p, _, _ = synth.make_scan(5, a.first + k * a.stride, a.kind, device="cuda")
        parts.append(p)
        offs.append(offs[-1] + p.shape[0])
    pts = torch.cat(parts).contiguous()
    offs = np.asarray(offs, np.int32)
    ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=a.scans)
    ctx.batch_process(pts, offs)
    cnt = ctx.batch_counts()
    names = {}
    for mode in [int(m) for m in a.modes.split(",")]:
        ctx.set_cluster_exact(mode)
        ts = []
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.batch_cluster()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        st = ctx.batch_cluster_stats()
        names[mode] = [ctx.batch_fetch_clusters(s, int(cnt[s, 4])) for s in range(a.scans)]
        print(f"mode {mode}: batch_cluster ms {[round(t, 2) for t in ts]}  {st}", flush=True)
    ref = 3 if 3 in names else max(names)
    for mode in names:
        if mode == ref:
            continue
        d = [int((names[mode][s] != names[ref][s]).sum()) for s in range(a.scans)]
        print(f"mode {mode} vs mode {ref}: {sum(d)} points differ in {sum(1 for x in d if x)} scans; scans: {[s for s, x in enumerate(d) if x][:40]}")


if __name__ == "__main__":
    main()
