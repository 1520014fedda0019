#!/usr/bin/env python3
"""bench.py -- scans/s of the SCV-OD hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): a SemanticKITTI-seq-05-shaped sequence -- 2761 scans of a
64-beam sensor, ~110-120 k returns per scan, config/semantickitti.yaml parameters -- synthetic
(no dataset exists in this environment), resident in HBM before the timed region starts.
One "step" = one pass of the hot path over the rank's whole sequence shard:
    Patchwork ground segmentation -> curved-voxel binning -> per-voxel descriptors
    -> scan-vs-next-scan occupancy probe for every consecutive pair,
processed in chunks of --chunk scans through the C-ABI (libscvod.so).  With N > 1 GPUs every
rank owns its own seq-05-shaped sequence (scans are independent: weak scaling, no data-path
collective); value = scans of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      HBM roofline of the dominant kernel, measured live with hipEvents on the launch stream
  cpu_baseline  the oracle (CPU restatement of the reference, single thread) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# Algorithmic (compulsory) bytes per scan, SURVEY.md 8(d), split by stage so that a kernel is priced
# against the bytes of ITS stage (DESIGN.md "Roofline accounting"):
#   patchwork : 16 N (read xyzi)   [the per-point class byte of 8(d) is not materialised per batch any more:
#               the two index lists carry it; scvod_batch_fetch builds the class array of a scan on request]
#   binning   : 4 N (voxel_idx) + 1 N (dynamic/static label)           -> emit (fused)
#   voxels    : 20 V                                                   -> vx_* kernels
#   tracking  : 16 N_car + 4 N_car + 64                                -> track_* kernels
STAGE_OF = {"pw_classify": "patchwork", "pw_offsets": "patchwork", "pw_scatter": "patchwork",
            "pw_sort_wave": "patchwork", "pw_sort_256": "patchwork", "pw_sort_1024": "patchwork", "pw_sort_2048": "patchwork", "pw_sort_4096": "patchwork", "pw_sort_8192": "patchwork", "pw_order": "patchwork", "pw_fit": "patchwork", "pw_fit_large": "patchwork",
            "pw_arrange": "patchwork", "emit_offsets": "patchwork",
            "emit": "binning", "vx_count": "voxels", "vx_offsets": "voxels", "vx_order": "voxels", "vx_scatter": "voxels",
            "vx_bucket_256": "voxels", "vx_bucket_1024": "voxels", "vx_bucket_2048": "voxels", "vx_bucket_4096": "voxels", "vx_bucket_8192": "voxels", "vx_final_offsets": "voxels",
            "vx_final": "voxels", "track_probe": "tracking", "track_unique": "tracking"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def stage_bytes(n_pts, n_vox, n_car, n_scans):
    return {"patchwork": 16.0 * n_pts, "binning": 5.0 * n_pts, "voxels": 20.0 * n_vox,
            "tracking": 20.0 * n_car + 64.0 * n_scans}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scans", type=int, default=2761, help="scans per rank (seq 05 has 2761)")
    ap.add_argument("--chunk", type=int, default=0, help="scans per C-ABI batch call (0 = scans / streams)")
    ap.add_argument("--streams", type=int, default=1, help="independent ctx + HIP stream pairs the chunks rotate over")
    ap.add_argument("--kind", default="K64")
    ap.add_argument("--preset", default="semantickitti")
    ap.add_argument("--cpu-scans", type=int, default=400, help="bounded sample for the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-all", action="store_true", help="skip the multi-threaded CPU context number")
    ap.add_argument("--pseudo-clusters", action="store_true", help="differencing stage on synthetic index runs instead of GPU car clusters")
    ap.add_argument("--no-extras", action="store_true", help="skip the separately reported next-row stages (clustering, boxes, VoxelGrid): profiling runs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same-device", action="store_true", help="dry run: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU: the SCV-OD path has no CPU fallback"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    import scvod_py
    import shard
    import synth

    P = scvod_py.make_params(args.preset)
    seq = 5 + 11 * rank  # every rank scans its own seq-05-shaped sequence
    dev = torch.device("cuda", local)

    if args.chunk <= 0:
        args.chunk = (args.scans + args.streams - 1) // args.streams
    # ---- synthetic sequence, resident in HBM ----
    t0 = time.time()
    chunks = []
    for c0 in range(0, args.scans, args.chunk):
        cnt = min(args.chunk, args.scans - c0)
        pts, offs, poses, _ = synth.make_batch(seq, c0, cnt, args.kind, device=dev)
        chunks.append(dict(pts=pts, offs=np.asarray(offs, np.int32), poses=poses, first=c0))
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    max_pts = max(int(c["offs"][-1]) for c in chunks)
    total_pts = sum(int(c["offs"][-1]) for c in chunks)
    n_ctx = max(1, min(args.streams, len(chunks)))
    ctxs = [scvod_py.Ctx(P, max_points_total=max_pts + 1024, max_scans=args.chunk, device=local) for _ in range(n_ctx)]
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_ctx - 1)]
    for i, c in enumerate(chunks):
        c["ctx"] = ctxs[i % n_ctx]
        c["stream"] = tstreams[i % n_ctx].cuda_stream
    ctx = ctxs[0]

    # ---- clusters for the differencing stage (built once, outside the timed region): the `car` clusters the
    # GPU clustering + bounding-box rules find in every scan -- what SSC::tracking walks.
    # --pseudo-clusters falls back to every 5th apri point in runs of 256. ----
    tot_vox = 0
    tot_car = 0
    for c in chunks:
        ctx, stream = c["ctx"], c["stream"]
        ctx.batch_process(c["pts"], c["offs"], stream=stream, sync=True)
        cnt = ctx.batch_counts()
        tot_vox += int(cnt[:, 6].sum())
        n_sc = cnt.shape[0]
        members, cbegin, pbegin = [], [0], [0]
        if not args.pseudo_clusters:
            ctx.batch_cluster(stream=stream, sync=False)
            ctx.batch_cluster_types(stream=stream, sync=True)
        for s in range(n_sc - 1):
            if args.pseudo_clusters:
                m = np.arange(0, cnt[s, 4], 5, dtype=np.int32)
                sizes = [min(256, len(m) - k) for k in range(0, len(m), 256)]
            else:
                names = ctx.batch_fetch_clusters(s, int(cnt[s, 4]))
                types = ctx.batch_fetch_cluster_types(s, int(cnt[s, 4]))
                idx = np.nonzero(types == 2)[0].astype(np.int32)
                order = np.argsort(names[idx], kind="stable")
                m = idx[order]
                nm = names[m]
                starts = np.nonzero(np.concatenate([[True], nm[1:] != nm[:-1]]))[0] if len(m) else np.zeros(0, np.int64)
                sizes = np.diff(np.append(starts, len(m))).tolist()
            members.append(m)
            for sz in sizes:
                cbegin.append(cbegin[-1] + int(sz))
            pbegin.append(len(cbegin) - 1)
        mem = np.concatenate(members) if members else np.zeros(0, np.int32)
        tot_car += len(mem)
        T = np.stack([ctx.pose_delta(c["poses"][s], c["poses"][s + 1]) for s in range(n_sc - 1)]) if n_sc > 1 else np.zeros((0, 12), np.float32)
        c["members"] = torch.from_numpy(mem if len(mem) else np.zeros(1, np.int32)).to(dev)
        c["cbegin"] = np.asarray(cbegin, np.int32)
        c["pbegin"] = np.asarray(pbegin, np.int32)
        c["T"] = T.astype(np.float32)
        c["n_sc"] = n_sc

    def step():
        for c in chunks:
            c["ctx"].batch_process(c["pts"], c["offs"], stream=c["stream"], sync=False)
            if c["n_sc"] > 1:
                c["ctx"].batch_track(c["members"], c["cbegin"], c["pbegin"], c["T"], stream=c["stream"], sync=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # per-kernel hipEvent timing on the launch stream for the timed steps
    kt = {}

    def timed_step():
        for c in chunks:
            ctx, stream = c["ctx"], c["stream"]
            ctx.batch_process(c["pts"], c["offs"], stream=stream, sync=False)
            for name, ms in ctx.timings():
                a = kt.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += 1
            if c["n_sc"] > 1:
                ctx.batch_track(c["members"], c["cbegin"], c["pbegin"], c["T"], stream=stream, sync=False)
                for name, ms in ctx.timings():
                    a = kt.setdefault(name, [0.0, 0])
                    a[0] += ms
                    a[1] += 1

    # timed region 1 (the number reported): no per-kernel events, async launches
    for x in ctxs:
        x.set_timing(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # timed region 2: the same steps with hipEvents around every kernel (roofline attribution)
    for x in ctxs:
        x.set_timing(True)
    barrier()
    for _ in range(args.steps):
        timed_step()
    barrier()
    for x in ctxs:
        x.set_timing(False)

    # "next" row 8(f)-1, reported separately (not part of `value`): curved-voxel clustering on the resident batch
    cc_ms = ct_ms = None
    try:
        if args.no_extras:
            raise RuntimeError("skipped")
        barrier()
        t1 = time.perf_counter()
        for c in chunks:
            c["ctx"].batch_process(c["pts"], c["offs"], stream=c["stream"], sync=False)
        barrier()
        t_proc = time.perf_counter() - t1
        t1 = time.perf_counter()
        for c in chunks:
            c["ctx"].batch_process(c["pts"], c["offs"], stream=c["stream"], sync=False)
            c["ctx"].batch_cluster(stream=c["stream"], sync=False)
        barrier()
        cc_ms = 1e3 * ((time.perf_counter() - t1) - t_proc)
        t1 = time.perf_counter()
        for c in chunks:  # 8(f)-2: bounding boxes + type rules on top of the clusters
            c["ctx"].batch_process(c["pts"], c["offs"], stream=c["stream"], sync=False)
            c["ctx"].batch_cluster(stream=c["stream"], sync=False)
            c["ctx"].batch_cluster_types(stream=c["stream"], sync=False)
        barrier()
        ct_ms = 1e3 * ((time.perf_counter() - t1) - t_proc) - cc_ms
    except Exception as e:  # never let the optional stage break the bench line
        cc_ms = ct_ms = None
    # "next" row 8(f)-3, reported separately: loader-side VoxelGrid 0.08 m over the resident sequence (into a second buffer)
    vg_ms = vg_ratio = e2e_ms = None
    try:
        if args.no_extras:
            raise RuntimeError("skipped")
        c = chunks[0]
        d_out = torch.empty_like(c["pts"])
        c["ctx"].batch_voxelgrid(c["pts"], c["offs"], d_out)  # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        oo = c["ctx"].batch_voxelgrid(c["pts"], c["offs"], d_out)
        torch.cuda.synchronize()
        vg_ms = 1e3 * (time.perf_counter() - t1) * (args.scans / (len(c["offs"]) - 1))
        vg_ratio = float(oo[-1]) / float(c["offs"][-1])
        # the reference's real order: loader (filter + VoxelGrid) THEN the hot path on the downsampled scans, all resident
        c["ctx"].batch_process(d_out, oo, stream=c["stream"], sync=True)
        t1 = time.perf_counter()
        oo = c["ctx"].batch_voxelgrid(c["pts"], c["offs"], d_out)
        c["ctx"].batch_process(d_out, oo, stream=c["stream"], sync=True)
        e2e_ms = 1e3 * (time.perf_counter() - t1) * (args.scans / (len(c["offs"]) - 1))
        del d_out
    except Exception as e:
        vg_ms = vg_ratio = e2e_ms = None
    dt, all_scans, all_pts = shard.aggregate(dist, dev if args.backend == "nccl" else torch.device("cpu"), dt, args.scans, total_pts)

    if rank == 0:
        scans_per_s = all_scans * args.steps / dt
        # dominant kernel and its HBM roofline
        n_chunks = len(chunks)
        sb = stage_bytes(total_pts, tot_vox, tot_car, args.scans)  # bytes per step (this rank)
        dom = max(kt.items(), key=lambda kv: kv[1][0]) if kt else None
        roof = None
        kernels = {}
        for name, (ms, cnt) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
            kernels[name] = {"avg_ms": ms / cnt, "launches": cnt, "share": ms / sum(v[0] for v in kt.values())}
        if dom:
            name, (ms, cnt) = dom
            stage = STAGE_OF.get(name, "patchwork")
            bytes_per_launch = sb[stage] / n_chunks  # one launch processes one chunk of the sequence
            avg_s = ms / cnt / 1e3
            ach = bytes_per_launch / avg_s / 1e9
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
            # gfx950 FETCH doubling per MI355X_MICROARCH.md), stored per scan in profiles/*pmc_traffic.json
            traffic = None
            try:
                pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json"))
                if pm:
                    per_scan = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["by_bench_label"].get(name)
                    if per_scan is not None:
                        traffic = per_scan * (args.scans / n_chunks)
            except Exception:
                traffic = None
            roof = {"bound": "hbm", "kernel": name, "stage": stage, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_ms_per_launch": ms / cnt,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "path_GBps": (sum(sb.values()) * args.steps / dt) / 1e9}
        cpu = None
        if not args.no_cpu and world == 1:
            import oracle_py
            orc = oracle_py.load()
            ns = min(args.cpu_scans, int(chunks[0]["n_sc"]))
            offs = chunks[0]["offs"][: ns + 1]
            x = chunks[0]["pts"][: int(offs[-1])].cpu().numpy()
            t1 = time.perf_counter()
            stages, _ = orc.time_process(P, x, offs)
            cpu_dt = time.perf_counter() - t1
            cpu = {"value": ns / cpu_dt, "unit": "scans/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} scans of the same synthetic seq-05 sequence (Patchwork+binning+voxel descriptors, "
                             f"oracle/liboracle.so, g++ -O3 no -march, 1 thread; {os.cpu_count()} host cores present)",
                   "host_cpu": _cpu_model(),
                   "stage_ms_per_scan": {"patchwork": 1e3 * stages[0] / ns, "bin": 1e3 * stages[1] / ns,
                                         "voxelize": 1e3 * stages[2] / ns}}
        # context only: the same oracle with one scan per host thread (ctypes releases the GIL), bounded to ~10 s
        cpu_all = None
        if cpu is not None and not args.no_cpu_all:
            try:
                from concurrent.futures import ThreadPoolExecutor
                nthr = min(os.cpu_count() or 1, 64)
                per = max(2, min(8, int(chunks[0]["n_sc"]) // nthr))
                offs_all = chunks[0]["offs"]
                xs = chunks[0]["pts"][: int(offs_all[nthr * per])].cpu().numpy()

                def work(t):
                    o = offs_all[t * per:(t + 1) * per + 1]
                    orc.time_process(P, xs[int(o[0]):int(o[-1])], (o - o[0]).astype(np.int32))
                t1 = time.perf_counter()
                with ThreadPoolExecutor(nthr) as ex:
                    list(ex.map(work, range(nthr)))
                cpu_all = {"value": nthr * per / (time.perf_counter() - t1), "unit": "scans/s", "threads": nthr,
                           "note": "one scan per thread, same oracle; context, not the baseline"}
            except Exception:
                cpu_all = None
        out = {"metric": "scans/sec on SemanticKITTI-seq-05-shaped input (SCV-OD hot path)", "value": scans_per_s,
               "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"seq05-shaped {args.kind} sequence, {args.scans} scans/rank, {args.preset}.yaml grid, "
                                      f"chunks of {args.chunk} scans on {n_ctx} stream(s)", "scans_per_rank": args.scans,
                          "points_per_scan": total_pts / args.scans, "voxels_per_scan": tot_vox / args.scans,
                          "car_points_per_scan": tot_car / args.scans, "car_clusters": "pseudo" if args.pseudo_clusters else "gpu clustering + bbox rules", "sharding": f"1 sequence per GPU x{world}"},
               "mpts_per_s": all_pts * args.steps / dt / 1e6, "gen_seconds": gen_s,
               "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
               "extras": {"cluster_ms_per_sequence": cc_ms, "cluster_types_ms_per_sequence": ct_ms, "voxelgrid_ms_per_sequence": vg_ms, "voxelgrid_kept_fraction": vg_ratio, "voxelgrid_then_path_ms_per_sequence": e2e_ms, "cpu_all_threads": cpu_all}}
        print(json.dumps(out))
    for x in ctxs:
        x.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
