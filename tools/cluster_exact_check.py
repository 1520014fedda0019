#!/usr/bin/env python3
"""How often does the bounded visiting-order model of the generic clustering variant (scans beyond the LDS tables, OS128 class)
change a partition, and does the local rule (cc_run_is_plain) ever?  Clusters the same scans in the three modes of
scvod_set_cluster_exact -- 0: rule + components up to 4096 nodes (default), 1: rule + components of any size, 2: no rule, every
component with an irregular run (the visiting-order model itself, checked against the oracle by the tests) -- and counts the
points whose cluster differs from mode 2's.
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
    for mode in (0, 1, 2):
        ctx.set_cluster_exact(mode)
        ctx.batch_cluster()
        st = ctx.batch_cluster_stats()
        out[mode] = ([ctx.batch_fetch_clusters(s, int(cnt[s, 4])) for s in range(a.scans)], st)
    total = sum(len(x) for x in out[2][0])
    print(f"{a.kind} {a.preset}: {a.scans} scans, {total} binned points")
    for mode in (0, 1):
        differ = scans_differ = 0
        for s in range(a.scans):
            d = int((out[mode][0][s] != out[2][0][s]).sum())
            differ += d
            scans_differ += d > 0
        print(f"mode {mode}: {out[mode][1]}\n   points whose cluster name differs from mode 2's: {differ} in {scans_differ} scans")
    print(f"mode 2: {out[2][1]}")


if __name__ == "__main__":
    main()
