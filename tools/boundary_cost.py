#!/usr/bin/env python3
"""Development tool: what one rank's share of the boundary exchange of a split sequence costs on the host and the device
(pyshim/shard.py resolve_chain_boundaries): the five exports of a chain's end state, their copy to the host, the compare call.
usage: python tools/boundary_cost.py [--scans 410]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
import scvod_py
import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=410)
    a = ap.parse_args()
    scvod_py.load_lib()
    P = scvod_py.make_params("semantickitti")
    n, skip = a.scans, 5
    scans = [synth.make_scan(5, i, "K64", device="cuda") for i in range(n)]
    d = torch.cat([s[0] for s in scans]).contiguous()
    offs = np.concatenate([[0], np.cumsum([len(s[0]) for s in scans])]).astype(np.int32)
    poses = np.asarray([s[2] for s in scans], np.float32)
    ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 64, max_scans=n)
    nxt = np.array([i + skip if i + skip < n else -1 for i in range(n)], np.int32)
    T = np.zeros((n, 12), np.float32)
    for s in range(n):
        if nxt[s] >= 0:
            T[s] = ctx.pose_delta(poses[s], poses[nxt[s]])
    halo = np.zeros(n, np.uint8)
    halo[:60] = 1
    ctx.set_track_halo(halo)
    for rep in range(3):
        ctx.batch_process(d, offs)
        ctx.batch_cluster()
        ctx.batch_cluster_types()
        ctx.batch_track(T, next_scan=nxt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        firsts = ctx.batch_track_chains()
        ends = ctx.chain_export_states(range(len(firsts)), 1)
        t1 = time.perf_counter()
        host = [e.cpu() for e in ends]
        t2 = time.perf_counter()
        back = [h.cuda() for h in host]
        snaps = [ctx.chain_export_state(c, 0) for c in range(len(firsts))]
        t3 = time.perf_counter()
        differs = ctx.batch_track_compare(snaps)
        t4 = time.perf_counter()
        print(f"rep {rep}: {len(firsts)} chains, records {[int(e.numel()) for e in ends]} bytes; export {1e3 * (t1 - t0):.2f} ms, to host {1e3 * (t2 - t1):.2f} ms, "
              f"compare call {1e3 * (t4 - t3):.2f} ms (differs {differs} against the chains' own snapshots)")


if __name__ == "__main__":
    main()
