#!/usr/bin/env python3
"""Development probe: does running V shards of the sequence on V HIP streams of ONE GPU overlap the latency-bound kernels of
one shard with the bandwidth-bound kernels of another?  (No boundary tables between the shards: timing only.)
--masks (round 6, verdict r5 #4): the streams of a V = 2 run are created with hipExtStreamCreateWithCUMask -- "160,96" gives shard 0
the first 160 CUs and shard 1 the other 96 (interleaved over the XCDs), so that the persistent grids of one shard cannot take the
other's workgroup slots; the shards' steps then drift against each other and the latency-bound kernels of one run beside the
bandwidth-bound kernels of the other.
usage: python tools/multi_stream_probe.py --scans 2760 --shards 1 2 3 [--masks 128,128 160,96 192,64]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=2760)
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--skip", type=int, default=5)
    ap.add_argument("--masks", nargs="*", default=[], help="CU splits for two shards, e.g. 160,96")
    a = ap.parse_args()
    import torch
    import scvod_py
    import synth
    dev = torch.device("cuda", 0)
    P = scvod_py.make_params("semantickitti")
    scans = [synth.make_scan(5, i, "K64", device=dev) for i in range(a.scans)]
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def masked_stream(cus, lo_first):
        """a stream restricted to `cus` of the 256 CUs: CU k of the mask's numbering belongs when (k % 256 * 256 // 256) falls in the share --
        bits are dealt round-robin so that both shares span all eight XCDs"""
        n = 256
        words = (C.c_uint32 * (n // 32))()
        take = set()
        # deal the CUs like cards: position k goes to the first share while k % 256 < ... (Bresenham: an even spread over the mask's numbering)
        acc = 0
        for k in range(n):  # (Bresenham: share 0 gets `cus` of the 256 positions, evenly spread; share 1 the others)
            acc += cus
            mine = acc >= n
            if mine:
                acc -= n
            if mine == bool(lo_first):
                take.add(k)
        for k in take:
            words[k // 32] |= 1 << (k % 32)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), n // 32, words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value, device=dev), len(take)

    runs = [(V, None) for V in a.shards] + [(2, tuple(int(x) for x in m.split(","))) for m in a.masks]
    for V, mask in runs:
        per = a.scans // V
        shards = []
        total = 0
        for v in range(V):
            sl = scans[v * per:(v + 1) * per]
            offs = np.zeros(per + 1, np.int32)
            offs[1:] = np.cumsum([p.shape[0] for p, _, _ in sl])
            pts = torch.cat([p for p, _, _ in sl], 0).contiguous()
            poses = np.asarray([pose for _, _, pose in sl], np.float32)
            ctx = scvod_py.Ctx(P, max_points_total=int(offs[-1]) + 1024, max_scans=per, device=0)
            nxt = np.asarray([i + a.skip if i + a.skip < per else -1 for i in range(per)], np.int32)
            T = np.zeros((per, 12), np.float32)
            for i in range(per):
                if nxt[i] >= 0:
                    T[i] = ctx.pose_delta(poses[i], poses[nxt[i]])
            if mask is None:
                stv = torch.cuda.Stream(device=dev)
            else:
                stv, got = masked_stream(mask[0], v == 0)
            shards.append((ctx, pts, offs, poses, nxt, T, stv))
            total += int(offs[-1])
        cells = 1 << int(np.ceil(np.log2(max(total * 0.25, 1 << 22))))
        smap = scvod_py.StaticMap(cells, device=0)
        first = shards[0][6]

        def step():
            smap.clear(stream=first.cuda_stream)
            cleared = torch.cuda.Event()
            cleared.record(first)
            for ctx, pts, offs, poses, nxt, T, st in shards:
                s = st.cuda_stream
                st.wait_event(cleared)
                ctx.batch_process(pts, offs, stream=s, sync=False)
                ctx.batch_cluster(stream=s, sync=False)
                ctx.batch_cluster_types(stream=s, sync=False)
                ctx.batch_track_tables(stream=s)
                ctx.batch_track(T, next_scan=nxt, stream=s, sync=False)
                smap.accumulate(ctx, poses, stream=s)

        for _ in range(3):  # (the chain planner cuts its segments by the times the first steps measured and may re-size its workspace once)
            step()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"shards {V}{' CU masks ' + str(mask) if mask else ''}: {dt * 1e3:.2f} ms per {per * V} scans, {per * V / dt:.0f} scans/s", flush=True)
        for sh in shards:
            sh[0].close()
        del shards, smap
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
