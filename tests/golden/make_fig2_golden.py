"""Builds tests/golden/fig2_509.npz from the only Patchwork outputs the reference ships: doc/fig2/509_g.pcd (the ground
cloud of scan 509, 40 346 points) and doc/fig2/509_seg.pcd (its segmented non-ground cloud, 29 940 points).  Run in the
build container (reads /root/reference); only DATA is committed: the xyz coordinates of the two clouds as float32."""
import os
import sys

import numpy as np

REF = "/root/reference/doc/fig2"


def load_pcd_xyz(path):
    lines = open(path).read().splitlines()
    i = [k for k, l in enumerate(lines) if l.startswith("DATA")][0]
    assert lines[i].split()[1] == "ascii"
    return np.array([[float(v) for v in l.split()[:3]] for l in lines[i + 1:] if l.strip()], np.float32)


if __name__ == "__main__":
    g = load_pcd_xyz(os.path.join(REF, "509_g.pcd"))
    s = load_pcd_xyz(os.path.join(REF, "509_seg.pcd"))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fig2_509.npz")
    np.savez_compressed(out, ground=g, seg=s)
    print(out, g.shape, s.shape, os.path.getsize(out), "bytes")
