#!/usr/bin/env python3
"""What the first steps of a FRESH ctx cost per kernel (hipEvents around every kernel) against the steady state: where the one-shot job's
extra milliseconds go.  usage: python tools/cold_steps.py [--scans 2761]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=2761)
    ap.add_argument("--skip", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import scvod_py
    import synth
    P = scvod_py.make_params("semantickitti")
    parts, offs, poses = [], [0], []
    for i in range(a.scans):
        p, _, pose = synth.make_scan(5, i, "K64", device="cuda")
        parts.append(p)
        offs.append(offs[-1] + p.shape[0])
        poses.append(pose)
    pts = torch.cat(parts).contiguous()
    del parts
    offs = np.asarray(offs, np.int32)
    poses = np.asarray(poses, np.float32)
    n = a.scans
    nxt = np.asarray([i + a.skip if i + a.skip < n else -1 for i in range(n)], np.int32)
    for rep in range(3):  # (the first ctx of the process also loads every code object: the second one is the fresh ctx of a warm process)
        ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 1024, max_scans=n)
        T = np.zeros((n, 12), np.float32)
        for s in range(n):
            if nxt[s] >= 0:
                T[s] = ctx.pose_delta(poses[s], poses[nxt[s]])
        ctx.set_timing(True)
        rows = []
        for step in range(a.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kt = {}
            for call in (lambda: ctx.batch_process(pts, offs, sync=False), lambda: ctx.batch_cluster(sync=False), lambda: ctx.batch_cluster_types(sync=False),
                         lambda: ctx.batch_track(T, next_scan=nxt, sync=False)):
                call()
                for name, ms in ctx.timings():
                    kt[name] = kt.get(name, 0.0) + ms
            torch.cuda.synchronize()
            rows.append((1e3 * (time.perf_counter() - t0), kt))
        last = rows[-1][1]
        print(f"ctx {rep}: wall ms per step (events around every kernel: slower than the asynchronous step) {[round(r[0], 2) for r in rows]}")
        for step in range(a.steps - 1):
            diff = sorted(((rows[step][1].get(k, 0.0) - last.get(k, 0.0), k) for k in set(last) | set(rows[step][1])), reverse=True)
            print(f"  step {step + 1} against step {a.steps}: sum of kernels {sum(rows[step][1].values()):.2f} vs {sum(last.values()):.2f} ms; largest differences:",
                  ", ".join(f"{k} {d:+.2f}" for d, k in diff[:6] if abs(d) > 0.05))
        ctx.close()


if __name__ == "__main__":
    main()
