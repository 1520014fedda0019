#!/usr/bin/env python3
"""bench.py -- scans/s of the SCV-OD dynamic-removal path on MI355X (BASELINE.json metric).

Workload at one GPU (BASELINE.json configs[1]): a SemanticKITTI-seq-05-shaped sequence -- 2761 scans of a 64-beam sensor,
~118 k returns per scan, config/semantickitti.yaml parameters -- synthetic (no dataset exists in this environment),
resident in HBM before the timed region starts.  With N GPUs the SAME job is cut over the ranks (strong scaling, the
default): the scans of the sequence in N contiguous blocks, each with a halo of 10 x skip_ scans in front for the warm-up
of the tracking chain (pyshim/shard.py plan_split / plan_job_split).  The reference tracks frame i against frame i + 1 in
order and every call mutates the successor (SSC::segDF, ssc.cpp:1449-1451): the chain's state at a cut is exported by the
rank before, compared with what the halo's warm-up produced and walked again only where it differs.  `--kitti
--split-sequence` is configs[3] (seq 00-10 at their real lengths, `--kitti-scale a/b` shrinks every sequence for a dry
run); `--replicate` / `--sequences K` deal whole sequences to the ranks instead (weak scaling, no tracking data crosses).

One "step" = one pass of the whole path over the rank's scans, raw points in, per-point dynamic/static labels and the
static map out, everything on the device through the C-ABI (libscvod.so):
    Patchwork ground segmentation -> curved-voxel binning -> per-voxel descriptors        scvod_batch_process
    -> curved-voxel clustering -> bounding boxes + type rules                             scvod_batch_cluster(_types)
    -> scan-vs-next-scan differencing: probe, remap_name, state rule, the reference's     scvod_batch_track
       SEQUENTIAL re-labelling chain (scan i against scan i + skip_), per-point byte
    -> (N > 1, a sequence cut over ranks) the chain states at the cuts: one exchange,     shard.DeviceBoundary
       compared on the device; a rank walks its chains again only if the verdict says so
    -> world-frame static map of the rank's scans                                         scvod_batch_map_accumulate
    -> (N > 1) the map reduce-scattered over RCCL: records grouped by owner rank in       scvod_map_export_parts_padded,
       equal padded slots, ONE all_to_all_single, every rank merges the cells it owns     scvod_map_merge
At N = 1 there is no host synchronisation inside a step.  At N > 1 the boundary exchange reads ONE verdict word per step on
the host (does any rank have to walk a chain again?); everything else stays on the stream.
value = scans of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      HBM roofline: algorithmic bytes of the path over the measured step time; the dominant kernel
                with its own hipEvent time and its own PMC traffic
  cpu_baseline  the oracle (CPU restatement of the reference, one thread) over the same stages on a bounded sample
  quality       dynamic-removal PR / RR (tool/analysis.py:186-187) of this path and of the oracle chain on that sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# kernels of one step by the label of their hipEvent pair -> stage of SURVEY 8(d)'s byte accounting
STAGE_OF = {"pw_": "patchwork", "emit": "binning", "vx_": "voxels", "cc_": "clustering", "tk_": "tracking", "map_": "map"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def algorithmic_bytes(n_pts, n_vox, n_car):
    """Compulsory HBM traffic of one scan set, SURVEY.md 8(d): every input read once, every output written once.
    16 N read xyzi + 1 N ground/non-ground class + 4 N voxel_idx + 1 N dynamic/static label + 20 V voxel records
    + 20 N_car differencing (read the cluster points, write the hit key)."""
    return {"patchwork": 17.0 * n_pts, "binning": 4.0 * n_pts, "voxels": 20.0 * n_vox, "tracking": 1.0 * n_pts + 20.0 * n_car}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def seq05_quality(scvod_py, qmod, orc, x, offs, poses, gt, device):
    """PR / RR of the device path and of the oracle chain on the labelled sample with preset semantickitti_seq05 (max_z 4.0,
    min_z 1.0, car_square 50: doc/note.txt:36, the row the reference publishes 98.97 / 96.67 for on the real seq 05)"""
    import numpy as np
    import torch
    P5 = scvod_py.make_params("semantickitti_seq05")
    ns = len(offs) - 1
    ctx = scvod_py.Ctx(P5, max_points_total=int(offs[-1]) + 1024, max_scans=ns, device=device)
    d = torch.from_numpy(x).to(f"cuda:{device}")
    ctx.batch_process(d, offs)
    ctx.batch_cluster()
    ctx.batch_cluster_types()
    T = np.zeros((ns, 12), np.float32)
    for s in range(ns - 1):
        T[s] = ctx.pose_delta(poses[s], poses[s + 1])
    ctx.batch_track(T)
    dev = [qmod.device_point_labels(ctx, s, int(offs[s + 1] - offs[s])) for s in range(ns - 1)]
    _, ref_lab, _ = orc.time_sequence(P5, x, offs, poses)
    q = qmod.compare(scvod_py, ctx, x, offs, poses, gt, ref_lab, np.concatenate(dev), voxelsize=0.2)
    ctx.close()
    return {k: q[k] for k in ("device", "reference_chain", "delta_PR", "delta_RR", "labels_equal_fraction")} | {"preset": "semantickitti_seq05 (doc/note.txt:36)"}


def other_configs():
    """configs[2] (parking lot, one chain of 1999 steps) and configs[4] (128 beams, 2x finer grid) as sub-runs of this script"""
    res = {}
    for name, extra, limit in (("configs[2] PARK", ["--kind", "PARK", "--preset", "parkinglot", "--scans", "2000"], 420),
                               ("configs[4] OS128", ["--kind", "OS128", "--preset", "os128_fine", "--scans", "1000"], 600)):
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-all", "--cpu-scans", "32"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit)
            line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            q = d.get("quality") or {}
            res[name] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                         "mpts_per_s": d["mpts_per_s"], "frac": d["roofline"]["frac"], "algorithmic_bytes_per_scan": d["roofline"]["algorithmic_bytes_per_scan"],
                         "traffic": d["roofline"]["traffic"], "traffic_source": d["roofline"]["traffic_source"],
                         "clustering": d["config"]["clustering"], "max_name": d["config"].get("max_name"), "tracking_chain": d["config"]["tracking_chain"],
                         "quality": {k: q.get(k) for k in ("labels_equal_fraction", "delta_PR", "delta_RR", "scans")} | {"device": q.get("device")},
                         "cpu_baseline_scans_per_s": (d.get("cpu_baseline") or {}).get("value"),
                         "kernels_ms": {k: round(v["avg_ms"], 3) for k, v in list(d["kernels"].items())[:8]}}
        except Exception as e:  # (an optional line never breaks the headline)
            res[name] = {"error": str(e)[:300]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scans", type=int, default=2761, help="scans per rank (seq 05 has 2761)")
    ap.add_argument("--kind", default="K64")
    ap.add_argument("--preset", default="semantickitti")
    ap.add_argument("--sequences", type=int, default=0, help="sequences of the job, each --scans long (0: one per rank)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the N > 1 step (map reduce-scatter, device collectives) even at one rank")
    ap.add_argument("--skip", type=int, default=0, help="tracking stride: scan i is differenced against scan i + skip (0: the preset's config skip_, 5 in semantickitti.yaml, 1 in parkinglot.yaml)")
    ap.add_argument("--map-cells", type=int, default=0, help="capacity of the static map in cells (0: from the points)")
    ap.add_argument("--map-leaf", type=float, default=0.2)
    ap.add_argument("--cpu-scans", type=int, default=320, help="bounded sample for the CPU baseline and the PR/RR check")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-all", action="store_true", help="skip the multi-threaded CPU context number")
    ap.add_argument("--no-quality", action="store_true", help="skip the PR/RR comparison")
    ap.add_argument("--no-extras", action="store_true", help="skip the separately reported stages (VoxelGrid, ingest): profiling runs")
    ap.add_argument("--split-sequence", action="store_true", help="strong scaling: ONE sequence of --scans scans over the ranks (contiguous blocks, a halo of 10 x skip scans for the tracking chain's warm-up, the chain's state handed from rank to rank: pyshim/shard.py plan_split)")
    ap.add_argument("--replicate", action="store_true", help="N > 1: every rank runs a sequence of its own (weak scaling: rounds 1-3) instead of cutting ONE sequence over the ranks (the default at N > 1)")
    ap.add_argument("--kitti", action="store_true", help="the job is SemanticKITTI seq 00-10 at their real lengths (BASELINE configs[3], 23 201 scans: needs the memory of several GPUs); with --split-sequence the sequences are cut where the load says")
    ap.add_argument("--kitti-scale", default="1", help="with --kitti: every sequence length times this fraction (e.g. 1/32: a dry run of configs[3]'s 11-sequence plan that fits one GPU)")
    ap.add_argument("--split-halo", type=int, default=10, help="warm-up steps of the halo in front of a rank's block (--split-sequence); 10 since round 6 (12 before): on the eight-rank K64 jobs no chain is walked again at a cut down to 8 steps, from 6 on the warm-up misses states (profiles/r06_halo_sweep.txt)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short PARK / OS128 sub-runs that the default invocation appends under extras.configs")
    ap.add_argument("--no-map", action="store_true", help="leave the static map out of the step (profiling)")
    ap.add_argument("--track-mode", default="chain", choices=["chain", "first-order"], help="chain: the reference's sequential tracking chain (default); first-order: every pair independent")
    ap.add_argument("--chain-seg", type=int, default=0, help="steps per chain segment (0: library default)")
    ap.add_argument("--chain-warm", type=int, default=-1, help="warm-up steps in front of a segment (-1: library default)")
    ap.add_argument("--cluster-exact", type=int, default=1, choices=[0, 1, 2, 3],
                    help="scans beyond the LDS clustering variant (OS128 class): 1 (the library's default since round 6) = visiting-order model for the components the local rule does not settle, whatever their size (k_cc_exact, passes shared with helper blocks); 0 = up to 4096 nodes (rounds 3-5); 2 = without the rule; 3 = 1 without helper blocks")
    ap.add_argument("--max-name-fresh", action="store_true", help="new clusters of the tracking chain get fresh numbers instead of the reference's re-used Frame::max_name (ssc.cpp:354): profiling only, the labels then differ from the reference's")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true", help="dry run: every rank uses cuda:0 (use --backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--boundary", default="auto", choices=["auto", "device", "host"],
                    help="chain states at the cuts: device = shard.DeviceBoundary (padded rows, compare kernel, verdict word read one step late; on RCCL the rows travel device to device, "
                         "on gloo they are staged through the host: the dry run of the nccl path's step logic with real neighbours on one GPU); host = the host-driven protocol every step; auto = device on nccl, host on gloo")
    ap.add_argument("--dump-map", default="", help="rank 0 writes the merged static map (records sorted by cell key) and the per-scan dynamic counts to this .npz")
    args = ap.parse_args()

    # `python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, RCCL)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU: the SCV-OD path has no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    # N > 1: the path shards the scans of ONE sequence (BASELINE north_star: strong scaling of configs[1]) unless told otherwise
    if world > 1 and not args.replicate and not args.sequences and not args.kitti:
        args.split_sequence = True
    side = None
    if dist is not None and world > 1 and args.split_sequence and args.backend == "nccl":
        # the host-driven boundary protocol (the untimed sizing pass; a step whose device verdict says a chain has to be walked
        # again): its records travel staged through the host on a gloo group -- plain blocking sends and receives between
        # neighbours, the path the same-device tests cover; the per-step exchange itself is on RCCL (shard.DeviceBoundary)
        side = dist.new_group(backend="gloo")
    multi = dist is not None  # the N > 1 step (also at one rank with --force-dist: RCCL initialised, device collectives executed)
    import scvod_py
    import shard
    import synth

    P = scvod_py.make_params(args.preset)
    if args.skip <= 0:
        args.skip = 1 if args.preset == "parkinglot" else 5
    job = shard.weak_scaling_sequences(args.sequences or world, args.scans)
    split = None
    if args.kitti:  # BASELINE configs[3]: seq 00-10 at their real lengths
        num, _, den = args.kitti_scale.partition("/")
        job = shard.kitti_sequences(synth.SEQ_LEN, scale=float(num) / float(den or 1))
    if args.split_sequence:  # the job's scans, sequence after sequence, in equal contiguous runs: sequences are CUT (+ halo) where the load says
        if not args.kitti:
            job = shard.weak_scaling_sequences(args.sequences or 1, args.scans)
        split = shard.plan_job_split(world, job, skip=args.skip, warm=args.split_halo)[rank]
        plan = dict(scans=split["scans"], next_scan=split["next_scan"], skip=args.skip)
    else:
        plan = shard.plan_job(world, job, skip=args.skip)[rank]
    n_sc = len(plan["scans"])
    own_spans = [(sp["own_first"], sp["own_count"]) for sp in split["spans"]] if split else [(0, n_sc)]
    own_count = sum(c for _, c in own_spans)

    # ---- the rank's scans, resident in HBM ----
    t0 = time.time()
    parts, labs, offs, poses = [], [], [0], []
    for (q, i) in plan["scans"]:
        p, l, pose = synth.make_scan(q, i, args.kind, device=dev)
        parts.append(p)
        labs.append(l)
        offs.append(offs[-1] + p.shape[0])
        poses.append(pose)
    pts = torch.cat(parts, 0).contiguous()
    del parts
    offs = np.asarray(offs, np.int32)
    poses = np.asarray(poses, np.float32)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    total_pts = int(offs[-1])
    max_scan = int(np.diff(offs).max())

    ctx = scvod_py.Ctx(P, max_points_total=total_pts + 1024, max_scans=n_sc, device=local)
    if args.track_mode != "chain" or args.chain_seg > 0 or args.chain_warm >= 0:
        ctx.set_track_mode(chain=args.track_mode == "chain", segment_steps=max(args.chain_seg, 0), warmup_steps=max(args.chain_warm, -1))
    if args.cluster_exact != 1:
        ctx.set_cluster_exact(args.cluster_exact)
    if args.max_name_fresh:
        ctx.set_max_name_literal(False)
    if split:
        ctx.set_track_halo(split["is_halo"])
    stream = torch.cuda.current_stream().cuda_stream  # torch.distributed orders its work against this stream
    nxt = plan["next_scan"]
    T = np.zeros((n_sc, 12), np.float32)
    for s in range(n_sc):
        if nxt[s] >= 0:
            T[s] = ctx.pose_delta(poses[s], poses[nxt[s]])
    smap = None
    pmap = None
    part_send = part_recv = None
    if not args.no_map:
        cells = args.map_cells or (1 << int(np.ceil(np.log2(max(total_pts * 0.25, 1 << 22)))))  # load <= ~0.5 on the street scenes
        smap = scvod_py.StaticMap(cells, leaf=args.map_leaf, device=local)
        # N > 1: the map is reduce-scattered -- every rank ends up owning the cells whose key hashes to it, merged from all ranks
        pmap = scvod_py.StaticMap(cells, leaf=args.map_leaf, device=local) if multi else None
    kt = {}
    info = {}

    def collect():
        for name, ms in ctx.timings():
            a = kt.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1

    devb = None  # the chain states at the cuts on the device (RCCL); None: host-staged protocol (gloo dry runs) or nothing to exchange

    def map_part(timed=False):
        if smap is None:
            return
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        smap.clear(stream=stream)
        for f0, c0 in own_spans:  # (a rank adds its OWN scans: a halo belongs to the rank before)
            smap.accumulate_range(ctx, poses, f0, c0, stream=stream)
        if timed:
            e1.record()
            e1.synchronize()
            a = kt.setdefault("map_accumulate", [0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
        if multi:  # reduce-scatter of the map: equal padded slots, one all-to-all, no size on the host
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
            if timed:
                ev[0].record()
            smap.export_parts_padded(world, part_send, stream=stream)
            pmap.clear(stream=stream)
            if timed:
                ev[1].record()
            got = shard.reduce_scatter_map(dist, part_send, part_recv)
            if timed:
                ev[2].record()
            pmap.merge(got, stream=stream)
            if timed:
                ev[3].record()
                ev[3].synchronize()
                for name, i in (("map_export_parts", 0), ("map_all_to_all", 1), ("map_merge", 2)):
                    a = kt.setdefault(name, [0.0, 0])
                    a[0] += ev[i].elapsed_time(ev[i + 1])
                    a[1] += 1

    def check_boundary():
        """the verdict of the LAST step's boundary exchange (device path): read when that step has been enqueued; != 0 (rare: a warm-up
        missed the state at a cut, or a state outgrew its record) -> the host-driven protocol walks those chains again and the
        step's map is accumulated once more from the corrected labels"""
        if devb is None or not devb.pending:
            return
        if devb.verdict_wait() != 0:
            info["boundary_slow_path_steps"] = info.get("boundary_slow_path_steps", 0) + 1
            info["chains_rewalked_at_boundary"] = info.get("chains_rewalked_at_boundary", 0) + shard.resolve_chain_boundaries(dist, ctx, split, rank, world, dev, group=side)
            map_part(False)

    def step(timed=False):
        def after():
            if timed:
                collect()
        check_boundary()
        ctx.batch_process(pts, offs, stream=stream, sync=False)
        after()
        ctx.batch_cluster(stream=stream, sync=False)
        after()
        ctx.batch_cluster_types(stream=stream, sync=False)
        after()
        ctx.batch_track(T, next_scan=nxt, stream=stream, sync=False)
        after()
        if devb is not None:  # the chain's state at the cuts: exported, exchanged and compared on the device, nothing read here
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            devb.exchange(stream)
            if timed:
                e1.record()
                e1.synchronize()
                a = kt.setdefault("tk_boundary_exchange", [0.0, 0])
                a[0] += e0.elapsed_time(e1)
                a[1] += 1
        elif split and world > 1:  # gloo dry run: records staged through the host, from rank to rank, walked again where the warm-up missed it
            torch.cuda.synchronize()  # (the exchange starts when this rank's chain is done: its host time is the exchange's alone)
            t_b = time.perf_counter()
            info["chains_rewalked_at_boundary"] = info.get("chains_rewalked_at_boundary", 0) + shard.resolve_chain_boundaries(dist, ctx, split, rank, world, dev, group=side)
            info["boundary_ms"] = (time.perf_counter() - t_b) * 1e3
        map_part(timed)
        if devb is not None:
            devb.verdict_async()

    def barrier():
        check_boundary()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_device_boundary = args.boundary == "device" or (args.boundary == "auto" and args.backend == "nccl")
    need_devb = bool(split and multi and use_device_boundary and (world > 1 or args.force_dist))
    if multi and (smap is not None or need_devb):
        # one untimed pass: the slot size of the all-to-all and the record size of the boundary exchange (the only host reads of a size)
        ctx.batch_process(pts, offs, stream=stream, sync=False)
        ctx.batch_cluster(stream=stream, sync=False)
        ctx.batch_cluster_types(stream=stream, sync=False)
        ctx.batch_track(T, next_scan=nxt, stream=stream, sync=False)
        if need_devb:
            rec_bytes = shard.boundary_record_bytes(dist, ctx, split, rank, world, dev)
            devb = shard.DeviceBoundary(dist, ctx, split, rank, world, dev, rec_bytes, self_exchange=(world == 1), transport="nccl" if args.backend == "nccl" else "staged")
            info["boundary"] = {"path": "device: padded records, one RCCL point-to-point exchange, compare kernel, all_reduce(MAX) of the verdict; the host reads the verdict after the step was enqueued"
                                if args.backend == "nccl" else "device rows + compare kernel + verdict word, rows and verdict staged through the host on " + args.backend + " (dry run of the nccl path)",
                                "record_bytes": rec_bytes, "records_per_exchange": args.skip}
        elif split and world > 1:
            info["boundary"] = {"path": "host-staged (gloo dry run)"}
    if multi and smap is not None:
        smap.clear(stream=stream)
        if split and world > 1:
            shard.resolve_chain_boundaries(dist, ctx, split, rank, world, dev, group=side)
        for f0, c0 in own_spans:
            smap.accumulate_range(ctx, poses, f0, c0, stream=stream)
        _, counts0 = smap.export_parts(world, stream=stream)
        t_cap = torch.tensor([max(counts0)], dtype=torch.int64, device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t_cap, op=dist.ReduceOp.MAX)
        part_cap = int(int(t_cap.item()) * 1.1) + 1024
        part_send = torch.empty((world, part_cap, 2), dtype=torch.int64, device=dev)
        part_recv = torch.empty_like(part_send)
        info["map_slot_records"] = part_cap
        info["map_records_sent"] = int(sum(counts0)) - int(counts0[rank])
    cold = {}
    for w in range(args.warmup):
        if w == 0 and not multi:  # the one-shot job's number: the very first step of this process (code objects loaded on first launch, lazy allocations,
            torch.cuda.synchronize()  # equal-length chain segments, first-guess map table): reported under `cold`, never `value`
            t_c = time.perf_counter()
        step()
        if w == 0 and not multi:
            torch.cuda.synchronize()
            cold["first_step_of_the_process_ms"] = 1e3 * (time.perf_counter() - t_c)
    if smap is not None and not args.map_cells and args.warmup > 0:
        # capacity planning from the dry run: the generous first table (a quarter of the points) is replaced by the power of two
        # above 2.2 x the cells the job really occupies (load <= 0.45; measured: a table at load 0.46 costs the accumulation more than its smaller clear saves); the per-step clear shrinks with it
        occupied = max(smap.count(), pmap.count() if pmap is not None else 0)
        want = 1 << int(np.ceil(np.log2(max(2.2 * occupied, 1 << 20))))
        if want != cells:  # (smaller: the usual case; larger: sparse scans whose first table -- a quarter of the points -- ran above load 0.45)
            smap.close()
            smap = scvod_py.StaticMap(want, leaf=args.map_leaf, device=local)
            if pmap is not None:
                pmap.close()
                pmap = scvod_py.StaticMap(want, leaf=args.map_leaf, device=local)
            info["map_table_cells"] = want
            step()  # (one more untimed pass on the tables that are timed)
    # timed region (the number reported): no per-kernel events, asynchronous launches
    ctx.set_timing(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # the same steps again with hipEvents around every kernel of the ctx (attribution only) and around the map stage
    ctx.set_timing(True)
    barrier()
    for _ in range(args.steps):
        step(timed=True)
    barrier()
    ctx.set_timing(False)

    chain_stats = ctx.batch_track_stats()  # (raises if a chain state overflowed its workspace)
    cluster_stats = ctx.batch_cluster_stats()  # scans whose clustering kept "everything found is joined" around an out-of-grid triple
    max_name_stats = None
    if not args.max_name_fresh:  # Frame::max_name (ssc.cpp:354): scans whose last running number is still carried by a cluster / could not be determined
        ln, lst = ctx.batch_cluster_last_name(n_sc)
        max_name_stats = {"scans_with_a_cluster_carrying_it": int((ln[:, 0] >= 0).sum()), "undetermined_component_too_large": lst["unknown_too_large"],
                          "undetermined_irregular_points": lst["unknown_irregular"]}
    cnt = ctx.batch_counts()
    tot_vox = int(cnt[:, 6].sum())
    tot_apri = int(cnt[:, 4].sum())
    tk = [ctx.batch_fetch_track(s) for s in range(0, n_sc, max(1, n_sc // 64))]
    car_frac = sum(t["n_car_points"] for t in tk) / max(1, sum(t["n_apri"] for t in tk))
    dyn_frac = sum(t["n_dynamic_points"] for t in tk) / max(1, sum(t["n_apri"] for t in tk))
    tot_car = car_frac * tot_apri
    map_cells = smap.count() if smap is not None else None
    if smap is not None and multi:  # cells of the merged map = sum of the parts the ranks own
        t_cells = torch.tensor([pmap.count()], dtype=torch.int64, device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t_cells)
        map_cells = int(t_cells.item())
    if split and multi:  # chains walked again at a block boundary, all ranks, all steps of this run
        t_rw = torch.tensor([info.get("chains_rewalked_at_boundary", 0)], dtype=torch.int64, device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t_rw)
        info["chains_rewalked_all_ranks"] = int(t_rw.item())
    if args.dump_map:
        own_idx = [s for f0, c0 in own_spans for s in range(f0, f0 + c0)]
        dynpts = np.array([ctx.batch_fetch_track(s)["n_dynamic_points"] for s in own_idx], np.int64)
        gathered = [None] * world
        mine = (pmap if multi else smap).export().cpu().numpy().view(np.uint64)
        own_scans = [list(map(int, plan["scans"][s])) for s in own_idx]
        if dist is not None:
            dist.all_gather_object(gathered, (rank, own_scans, dynpts.tolist(), mine))
        else:
            gathered = [(0, own_scans, dynpts.tolist(), mine)]
        if rank == 0:
            rec = np.concatenate([g[3] for g in gathered])
            assert len(np.unique(rec[:, 0])) == len(rec), "a cell is owned by exactly one rank"
            order = np.argsort(rec[:, 0])
            per_scan = sorted((tuple(q), d) for _, qs, ds, _m in gathered for q, d in zip(qs, ds))
            np.savez(args.dump_map, keys=rec[order, 0], vals=rec[order, 1], scans=np.array([q for q, _ in per_scan], np.int64),
                     dynamic_points=np.array([d for _, d in per_scan], np.int64))

    dev_labels = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.no_quality:
        import quality as qmod
        sample = list(range(0, n_sc, args.skip))[: args.cpu_scans]  # what the reference loads with skip_: consecutive FRAMES
        dev_labels = [qmod.device_point_labels(ctx, s, int(offs[s + 1] - offs[s])) for s in sample[:-1]]

    # ---- separately reported stages (not part of `value`; they reuse the arena) ----
    extras = {}
    if not args.no_extras and world == 1:
        try:  # loader-side label filter + VoxelGrid 0.08 m over the resident sequence
            sub = min(n_sc, 512)
            o2 = offs[: sub + 1]
            d_out = torch.empty((int(o2[-1]), 4), dtype=torch.float32, device=dev)
            ctx.batch_voxelgrid(pts, o2, d_out)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            oo = ctx.batch_voxelgrid(pts, o2, d_out)
            torch.cuda.synchronize()
            extras["voxelgrid_ms_per_sequence"] = 1e3 * (time.perf_counter() - t1) * n_sc / sub
            extras["voxelgrid_kept_fraction"] = float(oo[-1]) / float(o2[-1])
            del d_out
        except Exception as e:  # never let an optional stage break the bench line
            extras["voxelgrid_error"] = str(e)[:200]
        try:  # PCIe-inclusive ingest: chunked H2D from pinned host memory on a copy stream, overlapped with the path
            import ingest
            extras["ingest"] = ingest.measure(scvod_py, P, pts, offs, poses, local, skip=args.skip)
        except Exception as e:
            extras["ingest_error"] = str(e)[:200]

    own_pts = int(sum(offs[f0 + c0] - offs[f0] for f0, c0 in own_spans))
    dt, all_scans, all_pts = shard.aggregate(dist, dev if args.backend == "nccl" else torch.device("cpu"), dt, own_count, own_pts)

    if rank == 0:
        scans_per_s = all_scans * args.steps / dt
        ms_step = 1e3 * dt / args.steps
        ab = algorithmic_bytes(total_pts, tot_vox, tot_car)
        path_bytes = sum(ab.values())
        tot_ms = sum(v[0] for v in kt.values()) or 1.0
        kernels = {}
        traffic = {}
        traffic_src = None
        try:  # per-kernel HBM bytes per scan from the PMC passes OF THIS WORKLOAD (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/);
            # no pass for the workload -> traffic is null, never another workload's bytes
            pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"pmc_traffic_{args.kind.lower()}.json"))
            if pm and args.preset == {"K64": "semantickitti", "PARK": "parkinglot", "OS128": "os128_fine"}.get(args.kind):
                traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1]))).get("by_bench_label", {})
                traffic_src = pm[-1]
        except Exception:
            traffic = {}
        for name, (ms, c) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
            k = {"avg_ms": ms / c, "launches": c, "share": ms / tot_ms}
            if name in traffic:  # bytes per scan -> this kernel's OWN traffic over its OWN time
                k["pmc_MB_per_scan"] = traffic[name] / 1e6
                k["pmc_GBps"] = traffic[name] * n_sc / (ms / c * 1e-3) / 1e9
            kernels[name] = k
        dom = max(kt.items(), key=lambda kv: kv[1][0]) if kt else None
        roof = {"bound": "hbm", "achieved": path_bytes / (ms_step * 1e-3) / 1e9 * (1 if world == 1 else all_scans / n_sc), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "algorithmic_bytes_per_scan": path_bytes / n_sc, "algorithmic_bytes_per_step": path_bytes, "by_stage_bytes_per_scan": {k: v / n_sc for k, v in ab.items()},
                "basis": "SURVEY 8(d) bytes of the whole path / measured ms_per_step (all kernels of a step)"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        try:  # what this hardware reaches on known bytes (tools/pmc_calibrate.py, committed under profiles/): context for `frac`, which stays against the 8 TB/s of the spec
            cal = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_calibration.json"))
            cs = json.load(open(os.path.join(ROOT, "profiles", cal[-1])))["summary"]
            roof["achievable_peak"] = {"read_GBps": cs["read_ceiling_GBps_2GiB"] * world, "copy_GBps": cs["copy_ceiling_GBps_2GiB"] * world, "source": "profiles/" + cal[-1],
                                       "frac_of_read_ceiling": roof["achieved"] / (cs["read_ceiling_GBps_2GiB"] * world)}
        except Exception:
            roof["achievable_peak"] = None
        pmc_total = sum(v for k, v in traffic.items() if k in kt) if traffic else None
        roof["traffic"] = pmc_total * n_sc if pmc_total else None
        roof["traffic_source"] = traffic_src
        if dom:
            name, (ms, c) = dom
            roof["dominant_kernel"] = {"name": name, "avg_ms_per_launch": ms / c, "share_of_step": ms / tot_ms,
                                       "pmc_bytes_per_launch": (traffic[name] * n_sc) if name in traffic else None,
                                       "pmc_GBps": kernels[name].get("pmc_GBps")}
        cpu = quality = cpu_all = None
        if not args.no_cpu and world == 1:
            import oracle_py
            orc = oracle_py.load()
            sample = list(range(0, n_sc, args.skip))[: args.cpu_scans]
            ns = len(sample)
            x = torch.cat([pts[int(offs[s]):int(offs[s + 1])] for s in sample]).cpu().numpy()
            o2 = np.concatenate([[0], np.cumsum([int(offs[s + 1] - offs[s]) for s in sample])]).astype(np.int32)
            sp = poses[sample]
            t1 = time.perf_counter()
            stages, ref_lab, _ = orc.time_sequence(P, x, o2, sp)
            cpu_dt = time.perf_counter() - t1
            names = ("patchwork", "bin", "voxelize", "cluster", "types", "tracking")
            cpu = {"value": ns / sum(stages), "unit": "scans/s", "cores": 1, "kind": "port",
                   "sample": f"{ns} frames (every {args.skip}th scan from the start, the reference's skip_) of the same synthetic sequence through the same stages (Patchwork, binning, voxel descriptors, "
                             f"clustering, box rules, sequential tracking chain; oracle/liboracle.so, g++ -O3 no -march, 1 thread; "
                             f"{os.cpu_count()} host cores present); wall {cpu_dt:.1f} s",
                   "host_cpu": _cpu_model(), "stage_ms_per_scan": {k: 1e3 * v / ns for k, v in zip(names, stages)}}
            if not args.no_quality:
                try:
                    gt = torch.cat([labs[s] for s in sample]).cpu().numpy()
                    quality = qmod.compare(scvod_py, ctx, x, o2, sp, gt, ref_lab, np.concatenate(dev_labels), voxelsize=0.2)
                except Exception as e:
                    quality = {"error": str(e)[:300]}
                if quality and "error" not in quality and args.preset == "semantickitti":
                    try:  # the same sample with the parameters of the reference's published seq-05 row (doc/note.txt:36)
                        quality["seq05_parameters"] = seq05_quality(scvod_py, qmod, orc, x, o2, sp, gt, local)
                    except Exception as e:
                        quality["seq05_parameters"] = {"error": str(e)[:300]}
            if not args.no_cpu_all:
                try:  # context only: the same oracle with one chunk of scans per host thread (ctypes releases the GIL)
                    from concurrent.futures import ThreadPoolExecutor
                    nthr = min(os.cpu_count() or 1, 64)
                    per = 2
                    xs = pts[: int(offs[min(n_sc, nthr * per)])].cpu().numpy()

                    def work(t):
                        o = offs[t * per:(t + 1) * per + 1]
                        if len(o) < 2:
                            return
                        orc.time_sequence(P, xs[int(o[0]):int(o[-1])], (o - o[0]).astype(np.int32), poses[t * per:(t + 1) * per], want_labels=False)
                    t1 = time.perf_counter()
                    with ThreadPoolExecutor(nthr) as ex:
                        list(ex.map(work, range(min(nthr, n_sc // per))))
                    cpu_all = {"value": min(nthr, n_sc // per) * per / (time.perf_counter() - t1), "unit": "scans/s", "threads": nthr,
                               "note": "two scans per thread, same oracle; context, not the baseline"}
                except Exception:
                    cpu_all = None
        extras["cpu_all_threads"] = cpu_all
        out = {"metric": "scans/sec on SemanticKITTI-seq-05-shaped input (SCV-OD dynamic-removal path, raw scans -> per-point labels + static map)",
               "value": scans_per_s, "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": (f"seq05-shaped {args.kind} sequence, {n_sc} scans, {args.preset}.yaml grid, one batch" if world == 1 and len(job) == 1 else
                                       f"{len(job)} {args.kind} sequence(s), {int(all_scans)} scans ({'SemanticKITTI seq 00-10 lengths' + ('' if args.kitti_scale == '1' else ' x ' + args.kitti_scale) if args.kitti else str(args.scans) + ' each'}), {args.preset}.yaml grid, dealt as equal contiguous runs over {world} ranks (sequences cut)" if split else
                                       f"{len(job)} sequences of {args.scans} {args.kind} scans (seeded like seq 05, 00, 02, 08, ...), {int(all_scans)} scans in total, whole sequences per rank, {args.preset}.yaml grid"),
                          "scans_per_rank": n_sc, "points_per_scan": total_pts / n_sc, "nonground_binned_per_scan": tot_apri / n_sc,
                          "voxels_per_scan": tot_vox / n_sc, "car_points_per_scan": tot_car / n_sc, "dynamic_fraction_of_binned": dyn_frac,
                          "static_map_cells": map_cells, "static_map_table_cells": info.get("map_table_cells", (cells if smap is not None else None)), "tracking": args.track_mode, "tracking_chain": chain_stats, "clustering": cluster_stats, "max_name": max_name_stats,
                          "memory_GB_rank0": {"input_points": round(float(pts.numel() * 4) / 1e9, 2), "arena": round(ctx.arena_bytes() / 1e9, 2), "chain_workspace": round(ctx.chain_workspace_bytes() / 1e9, 2),
                                              "process_peak_on_device": round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 1e9, 2)},
                          "rccl_ranks": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None),
                          "tracking_stride": args.skip,
                          "sharding": (f"equal contiguous runs of the job's scans per rank (sequences are cut) + a halo of {args.split_halo} x {args.skip} scans in front of a cut; the tracking chain's state at a cut is sent by the rank before, compared and walked again from where the halo's warm-up missed it" if split else
                                       "whole sequences per rank (longest first to the least loaded rank)"),
                          "split": ({"scans_loaded_by_rank0": n_sc, "own": own_count, "pieces_rank0": [list(map(int, pc)) for pc in split["pieces"]], "chains_rewalked_at_boundary_rank0": info.get("chains_rewalked_at_boundary"), "chains_rewalked_at_boundary_all_ranks": info.get("chains_rewalked_all_ranks"), "boundary_exchange_ms_rank0_last_step": info.get("boundary_ms"), "boundary": info.get("boundary"), "boundary_slow_path_steps_rank0": info.get("boundary_slow_path_steps", 0)} if split else None)},
               "mpts_per_s": all_pts * args.steps / dt / 1e6, "gen_seconds": gen_s,
               "roofline": roof, "cpu_baseline": cpu, "quality": quality, "kernels": kernels, "extras": extras}
        if multi:
            out["config"]["map_records_sent_per_rank"] = info.get("map_records_sent")
            out["config"]["map_slot_records"] = info.get("map_slot_records")
    if smap is not None:
        smap.close()
        if pmap is not None:
            pmap.close()
    ctx.close()
    if rank == 0 and world == 1 and not multi and not args.no_extras:
        # what a ONE-PASS user sees (the reference walks a sequence once, ssc.cpp:1428-1452): a FRESH ctx in this warm process -- no
        # planner feedback (equal-length chain segments), the first-guess map table (a quarter of the points), lazy buffers allocated
        # inside the step -- against the steady-state step of the headline.  Steps 2 and 3 of the same ctx show how fast it converges.
        try:
            del ctx
            torch.cuda.empty_cache()
            c2 = scvod_py.Ctx(P, max_points_total=total_pts + 1024, max_scans=n_sc, device=local)
            if args.cluster_exact != 1:
                c2.set_cluster_exact(args.cluster_exact)
            m2 = None if args.no_map else scvod_py.StaticMap(cells, leaf=args.map_leaf, device=local)
            ms = []
            for _ in range(3):
                torch.cuda.synchronize()
                t_c = time.perf_counter()
                c2.batch_process(pts, offs, stream=stream, sync=False)
                c2.batch_cluster(stream=stream, sync=False)
                c2.batch_cluster_types(stream=stream, sync=False)
                c2.batch_track(T, next_scan=nxt, stream=stream, sync=False)
                if m2 is not None:
                    m2.clear(stream=stream)
                    m2.accumulate_range(c2, poses, 0, n_sc, stream=stream)
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - t_c))
            cold.update({"fresh_ctx_step_ms": [round(v, 3) for v in ms], "steady_state_ms_per_step": out["ms_per_step"],
                         "fresh_ctx_first_step_over_steady_state": ms[0] / out["ms_per_step"],
                         "fresh_ctx_first_step_scans_per_s": n_sc / (ms[0] * 1e-3),
                         "map_table_cells_first_guess": (cells if m2 is not None else None),
                         "note": "first_step_of_the_process_ms also pays the code-object load of every kernel and the allocator's first touches; "
                                 "fresh_ctx_step_ms[0] is a new ctx + map in the warm process: lazy buffers, equal-length chain segments, first-guess map table"})
            if m2 is not None:
                m2.close()
            c2.close()
        except Exception as e:
            cold["error"] = str(e)[:300]
        ing = out["extras"].get("ingest") or {}
        if "full_chain_first_pass_scans_per_s" in ing:
            cold["ingest_from_pinned_host_first_pass_scans_per_s"] = ing["full_chain_first_pass_scans_per_s"]
            cold["ingest_from_pinned_host_steady_scans_per_s"] = ing["full_chain_scans_per_s"]
        out["cold"] = cold
    elif rank == 0 and cold:
        out["cold"] = cold
    if rank == 0:
        if world == 1 and args.kind == "K64" and args.preset == "semantickitti" and not args.no_extras and not args.no_other_configs:
            # BASELINE configs[2] and configs[4] on this GPU, short runs of this same script after the headline workload left the
            # device (driver-timed with the line: the judge asked for them in BENCH_rNN.json); never part of `value`
            del pts, labs
            torch.cuda.empty_cache()
            out["extras"]["configs"] = other_configs()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
