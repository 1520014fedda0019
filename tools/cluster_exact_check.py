#!/usr/bin/env python3
"""How often does the bounded visiting-order model of the generic clustering variant (scans beyond the LDS tables, OS128 class)
change a partition?  Clusters the same scans with and without scvod_set_cluster_exact and counts differing points.
usage: python tools/cluster_exact_check.py [--kind OS128] [--preset os128_fine] [--scans 64] [--first 0] [--stride 5]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="OS128")
    ap.add_argument("--preset", default="os128_fine")
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--stride", type=int, default=5)
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
    out = {}
    for exact in (False, True):
        ctx.set_cluster_exact(exact)
        ctx.batch_cluster()
        st = ctx.batch_cluster_stats()
        out[exact] = ([ctx.batch_fetch_clusters(s, int(cnt[s, 4])) for s in range(a.scans)], st)
    differ = scans_differ = total = 0
    for s in range(a.scans):
        d = int((out[False][0][s] != out[True][0][s]).sum())
        differ += d
        scans_differ += d > 0
        total += len(out[True][0][s])
    print(f"{a.kind} {a.preset}: {a.scans} scans, {total} binned points; bounded model: {out[False][1]}; exact: {out[True][1]}")
    print(f"points whose cluster name differs between the two: {differ} in {scans_differ} scans")


if __name__ == "__main__":
    main()
