"""PCIe-inclusive throughput of the path (bench.py `extras.ingest`): the sequence starts in pinned HOST memory and travels
to the GPU in chunks through scvod_sequence_ingest (double-buffered H2D on a copy stream, overlapped with the previous
chunk's kernels); per chunk the consumers of bench.py's step are enqueued from the chunk callback.  Never `value`."""
import ctypes as C
import time

import numpy as np


def measure(scvod_py, P, d_pts, offs, poses, device, scans=2048, chunk=256, skip=5):
    import torch
    n_sc = min(int(scans), len(offs) - 1)
    o = np.ascontiguousarray(offs[: n_sc + 1], np.int32)
    n_pts = int(o[-1])
    host = torch.empty((n_pts, 4), dtype=torch.float32).pin_memory()
    host.copy_(d_pts[:n_pts])
    torch.cuda.synchronize()
    lib = scvod_py.load_lib()
    lib.scvod_sequence_ingest.restype = C.c_int
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)
    lib.scvod_sequence_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, CB, C.c_void_p]
    max_chunk = max(int(o[min(k + chunk, n_sc)] - o[k]) for k in range(0, n_sc, chunk))
    ctx = scvod_py.Ctx(P, max_points_total=max_chunk + 1024, max_scans=chunk, device=device)
    T = np.zeros((chunk, 12), np.float32)
    nxt = np.zeros(chunk, np.int32)
    Tall = np.zeros((n_sc, 12), np.float32)
    for s in range(n_sc - skip):
        Tall[s] = ctx.pose_delta(poses[s], poses[s + skip])

    def consumers(user, h, first, n, stream):
        # clustering -> box rules -> differencing inside the chunk: scan i against scan i + skip_ like the resident job (the
        # chains restart at every chunk: the last `skip` scans of a chunk have their successors in the next one and stay undecided)
        if lib.scvod_batch_cluster(h, stream, 0) or lib.scvod_batch_cluster_types(h, stream, 0):
            return -3
        T[:n] = Tall[first:first + n]
        nxt[:n] = np.arange(n) + skip
        nxt[max(n - skip, 0):n] = -1
        nxt[n:] = -1
        return lib.scvod_batch_track(h, T.ctypes.data_as(C.c_void_p), nxt.ctypes.data_as(C.c_void_p), None, 0, stream, 0)

    out = {}
    for name, cb in (("process_only", CB(lambda *a: 0)), ("full_chain", CB(consumers))):
        times = []
        for rep in range(3):
            t0 = time.perf_counter()
            rc = lib.scvod_sequence_ingest(ctx.h, C.c_void_p(host.data_ptr()), o.ctypes.data_as(C.c_void_p), n_sc, chunk, 0, cb, None)
            times.append(time.perf_counter() - t0)
            if rc != 0:
                raise RuntimeError(f"scvod_sequence_ingest: status {rc}: {lib.scvod_last_error(ctx.h).decode()}")
        out[name + "_scans_per_s"] = n_sc / min(times[1:])
        out[name + "_first_pass_scans_per_s"] = n_sc / times[0]  # (the one-shot job: a ctx that has never run these kernels at this size)
    # the PCIe ceiling on this box: the same bytes, pinned host -> device, nothing else
    dst = torch.empty_like(d_pts[:n_pts])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gbs = n_pts * 16 / best / 1e9
    out.update({"scans": n_sc, "chunk_scans": chunk, "bytes_per_scan": n_pts * 16 / n_sc, "h2d_GBps": gbs, "h2d_bound_scans_per_s": n_sc / best,
                "fraction_of_h2d_bound": out["full_chain_scans_per_s"] / (n_sc / best)})
    ctx.close()
    return out
