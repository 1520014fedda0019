"""Builds tests/golden/fig2_509_seg.npz from doc/fig2/509_seg.pcd: the segmented non-ground cloud the reference's own
binary wrote for scan 509 (SSC::saveSegCloud mode 1, src/ssc.cpp:468-548: every cluster of frame_ssc.cluster_set in one
random colour, its occupy_pts in order).  Run in the build container (reads /root/reference); only DATA is committed:
xyz as float32 and the packed rgb word of every point (= a cluster id as the reference saw it)."""
import os

import numpy as np

REF = "/root/reference/doc/fig2/509_seg.pcd"

if __name__ == "__main__":
    lines = open(REF).read().splitlines()
    i = [k for k, l in enumerate(lines) if l.startswith("DATA")][0]
    assert lines[i].split()[1] == "ascii" and lines[2].split()[1:] == ["x", "y", "z", "rgb"]
    rows = [l.split() for l in lines[i + 1:] if l.strip()]
    xyz = np.array([[float(v) for v in r[:3]] for r in rows], np.float32)
    rgb = np.array([int(r[3]) for r in rows], np.uint32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fig2_509_seg.npz")
    np.savez_compressed(out, xyz=xyz, rgb=rgb)
    print(out, xyz.shape, len(np.unique(rgb)), "colours", os.path.getsize(out), "bytes")
