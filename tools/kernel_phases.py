#!/usr/bin/env python3
"""Development tool: where k_cc_scan and the large sort tiers spend their time.  Needs the profiling build (make -C dr-using-scv-od_amd/csrc prof):
the kernel then sums the 100 MHz wall clock between its phases over all workgroups (thread 0, behind a barrier).
usage: python tools/kernel_phases.py [--kind K64|PARK|OS128] [--preset semantickitti] [--scans 256]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
KERNELS = [
    ("k_cc_scan (per scan)", ["regularity + start bits", "runs (two gathers)", "prefix + key table + init", "neighbour search + unions",
                              "extra runs, canonical names", "flatten + compact ids", "names per slot", "boxes", "types, voxel table, car lists",
                              "type + members per point", "exact path: tables + listed voxels", "exact path: (rest of the rounds)", "exact path: q + unions",
                              "exact path: rounds", "exact path: number of rounds"]),
    ("k_pw_sort<4096> (per scan)", ["load keys", "bitonic sort", "gather + write sorted points"]),
    ("k_pw_sort<8192> (per scan)", ["load keys", "bitonic sort", "gather + write sorted points"]),
    ("k_vx_bucket<4096> (per scan)", ["load keys", "bitonic sort", "heads + voxel starts", "stage intensities", "per-voxel sums", "final writes"]),
    ("k_cc_scan generic variant, inside 'neighbour search + unions' (per scan)",
     ["opener triples + regular bits", "spill, plane starts, init", "windows: plan", "windows: load", "windows: search in LDS", "windows: write-out",
      "(from 'extra runs, canonical names') extra runs", "(from 'extra runs, canonical names') rest of the affected-components block",
      "irregular runs: rule, marks, sample, list", "exact path: run records + listed voxels", "exact path: rounds", "exact path: q", "exact path: unions",
      "exact path: number of rounds"]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="K64")
    ap.add_argument("--preset", default="semantickitti")
    ap.add_argument("--scans", type=int, default=256)
    ap.add_argument("--irregular", type=int, default=0, help="returns at polar angle exactly 0 appended to every scan (sector index -1)")
    ap.add_argument("--lib", default="libscvod_prof.so", help="profiling build to load (file name under csrc/)")
    a = ap.parse_args()
    import torch
    import scvod_py
    import synth
    scvod_py.LIB_PATH = os.path.join(ROOT, "dr-using-scv-od_amd", "csrc", a.lib)
    lib = scvod_py.load_lib()
    dev = torch.device("cuda", 0)
    parts, offs = [], [0]
    for i in range(a.scans):
        p, _, _ = synth.make_scan(5, i * 7, a.kind, device=dev)
        if a.irregular:
            g = torch.Generator(device="cpu").manual_seed(i)
            e = torch.stack([torch.rand(a.irregular, generator=g) * 22 + 3, torch.zeros(a.irregular), torch.rand(a.irregular, generator=g) * 1.8 - 0.6,
                             torch.rand(a.irregular, generator=g)], 1).to(dev)
            p = torch.cat([p, e], 0)
        parts.append(p)
        offs.append(offs[-1] + p.shape[0])
    pts = torch.cat(parts, 0).contiguous()
    offs = np.asarray(offs, np.int32)
    ctx = scvod_py.Ctx(scvod_py.make_params(a.preset), max_points_total=int(offs[-1]) + 1024, max_scans=a.scans, device=0)
    out = (C.c_ulonglong * 128)()
    for rep in range(2):
        lib.scvod_debug_profile(out)
        ctx.batch_process(pts, offs)
        ctx.batch_cluster()
        lib.scvod_debug_profile(out)
    t = np.asarray(list(out), np.float64).reshape(8, 16) * 0.01 / a.scans  # us per scan, summed over the workgroups of a kernel
    print(f"{a.kind} {a.preset}: {a.scans} scans, {offs[-1] / a.scans:.0f} points per scan; workgroup-time per scan in us "
          "(a kernel with W workgroups resident per CU overlaps W of them)")
    for k, (name, phases) in enumerate(KERNELS):
        tot = t[k, :len(phases)].sum()
        if tot <= 0:
            continue
        print(f" {name}: sum {tot:.1f}")
        for ph, v in zip(phases, t[k]):
            print(f"    {ph:32s} {v:8.1f}  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
