#!/usr/bin/env python3
"""Development sweep for the SHARED exact re-clustering (k_cc_exact, round 6): large random clouds on fine grids -- walls, a ground disc, blobs,
1-2 % of the points with an index triple outside the grid -- so that the generic clustering variant runs, components of more than 6144
listed nodes appear and their passes go on the claim board.  Device (default mode) against the oracle's literal loop, point for point.
usage: python tests/devtools/cluster_shared_fuzz.py [--seed 1] [--clouds 40]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def big_cloud(rng):
    n = int(rng.integers(120000, 250000))
    # (coarse cells: several points per voxel, so that the listed nodes of a giant component stay below 0.35 x the points -- the room the
    #  16-bit rows have in the scan's box scratch -- while the scan still runs the generic variant: more than 65 536 binned points)
    kw = dict(range_res=float(rng.choice([0.4, 0.8])), sector_res=float(rng.choice([1.2, 2.4])), azimuth_res=float(rng.choice([2.0, 4.0])))
    kind = rng.random(n)
    r = 30.0 * np.sqrt(rng.uniform(0.0003, 1.0, n))  # (uniform over a disc of 30 m: dense enough for components of tens of thousands of voxels)
    th = rng.uniform(0, 2 * np.pi, n)
    x = np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-3, 10, n), rng.uniform(0, 255, n)], 1)
    wall = kind < 0.35
    x[wall, 0] = np.round(x[wall, 0] / 8) * 8 + rng.normal(0, 0.04, wall.sum())   # walls across the x axis
    disc = (kind >= 0.35) & (kind < 0.75)
    x[disc, 2] = -1.7 + rng.normal(0, 0.03, disc.sum())                           # a ground disc: one large component
    few = rng.choice(np.nonzero(disc | wall)[0], size=int(rng.integers(1, 12)), replace=False)
    x[few, 1] = 0.0                                                               # a handful of returns at polar angle exactly 0 (sector index -1) INSIDE the large components
    x[few, 0] = np.abs(x[few, 0])
    return kw, x.astype(np.float32)


def canonical(labels):
    labels = np.asarray(labels)
    order = np.argsort(labels, kind="stable")
    first = np.ones(len(labels), bool)
    first[1:] = labels[order][1:] != labels[order][:-1]
    mins = np.minimum.reduceat(order, np.nonzero(first)[0])
    out = np.empty(len(labels), np.int64)
    out[order] = np.repeat(mins, np.diff(np.append(np.nonzero(first)[0], len(labels))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--clouds", type=int, default=40)
    ap.add_argument("--mode", type=int, default=2, help="scvod_set_cluster_exact: 2 (default here) = without the local rule, so that EVERY component with an irregular run is clustered again -- the large ones on the claim board; 1 = the library's default")
    a = ap.parse_args()
    import oracle_py
    import scvod_py
    orc = oracle_py.load()
    rng = np.random.default_rng(a.seed)
    shared = chunks = generic = again = bad = 0
    for case in range(a.clouds):
        kw, x = big_cloud(rng)
        P = scvod_py.make_params("semantickitti", **kw)
        apri = orc.bin(P, x, case % 3 != 0)["apri"]
        if len(apri) == 0:
            continue
        ctx = scvod_py.Ctx(P, max_points_total=len(apri) + 64, max_scans=1)
        ctx.set_cluster_exact(a.mode)
        got = ctx.cluster(apri)
        st = ctx.batch_cluster_stats()
        can = canonical(orc.cluster(P, apri)[0])
        ok = np.array_equal(canonical(got), can) and st["scans_approximated"] == 0
        nv = len(np.unique(apri["voxel_idx"]))
        generic += nv > 14336
        shared += st["scans_that_shared_their_rounds"]
        chunks += st["chunks_taken_by_helpers"]
        again += st["runs_clustered_again"]
        bad += not ok
        big = int(np.bincount(np.unique(np.stack([canonical(got), apri["voxel_idx"].astype(np.int64)], 1), axis=0)[:, 0]).max())
        print(f"case {case}: {len(apri)} points, {nv} voxels (largest cluster {big} voxels), clustered again {st['runs_clustered_again']}, shared {st['scans_that_shared_their_rounds']}, "
              f"helper chunks {st['chunks_taken_by_helpers']}: {'ok' if ok else 'DIFFERS ' + str(int((canonical(got) != can).sum()))}", flush=True)
        ctx.close()
    print(f"seed {a.seed}: {a.clouds} clouds, {generic} beyond the LDS variant, {again} runs clustered again, {shared} shared their passes ({chunks} helper chunks), {bad} differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
