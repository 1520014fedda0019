"""Utility::saveCloud / loadCloud of the facade (host/utility.h; reference include/utility.h:408-428 = pcl::io::savePCDFile /
loadPCDFile): the writer's text against a file PCL itself wrote -- tests/golden/fig2_509_g_head.pcd holds the first 21 lines
(header + ten points) of the reference's doc/fig2/509_g.pcd -- and the reader on ASCII and binary files.  No device needed."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dr-using-scv-od_amd", "host")
HEAD = os.path.join(ROOT, "tests", "golden", "fig2_509_g_head.pcd")


def _exe():
    exe = os.path.join(HOST, "scvod_sequence")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST])
    return exe


def _copy(tmp_path, src, ident=7, name="_copy.pcd"):
    r = subprocess.run([_exe(), "--pcd-copy", str(src), str(tmp_path) + "/", str(ident), name], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return (tmp_path / f"{ident}{name}").read_text().splitlines(), r.stdout


def test_writer_formats_like_pcl(tmp_path):
    ref = open(HEAD).read().splitlines()
    pts = [l.split() for l in ref[11:]]
    src = tmp_path / "in.pcd"  # the same ten points as an x y z intensity cloud
    src.write_text("\n".join(["# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7", "FIELDS x y z intensity", "SIZE 4 4 4 4", "TYPE F F F F",
                              "COUNT 1 1 1 1", f"WIDTH {len(pts)}", "HEIGHT 1", "VIEWPOINT 0 0 0 1 0 0 0", f"POINTS {len(pts)}", "DATA ascii"] +
                             [f"{p[0]} {p[1]} {p[2]} {0.25 * i}" for i, p in enumerate(pts)]) + "\n")
    out, log = _copy(tmp_path, src)
    assert f"points {len(pts)}" in log
    # header: the lines that do not depend on the point type or count are PCL's, byte for byte
    for k in (0, 1, 3, 5, 7, 8, 10):
        assert out[k] == ref[k], (out[k], ref[k])
    assert out[2] == "FIELDS x y z intensity" and out[4] == "TYPE F F F F"
    assert out[6] == f"WIDTH {len(pts)}" and out[9] == f"POINTS {len(pts)}"
    # the coordinates print exactly as PCL printed them (eight significant digits, shortest form)
    for line, p in zip(out[11:], pts):
        assert line.split()[:3] == p[:3]
    assert [l.split()[3] for l in out[11:]] == ["0", "0.25", "0.5", "0.75", "1", "1.25", "1.5", "1.75", "2", "2.25"]


def test_reader_takes_binary_files_and_extra_fields(tmp_path):
    rng = np.random.default_rng(5)
    x = rng.normal(0, 20, (257, 4)).astype(np.float32)
    x[3, 2] = np.float32(1e-7)
    x[4, 0] = np.float32(123456792.0)
    ring = np.arange(257, dtype=np.uint16)
    src = tmp_path / "bin.pcd"
    with open(src, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z ring intensity\nSIZE 4 4 4 2 4\nTYPE F F F U F\nCOUNT 1 1 1 1 1\n"
                 "WIDTH 257\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 257\nDATA binary\n").encode())
        for p, r in zip(x, ring):
            f.write(struct.pack("<fffHf", p[0], p[1], p[2], int(r), p[3]))
    out, _ = _copy(tmp_path, src, ident=12, name=".pcd")
    got = np.array([[np.float32(t) for t in l.split()] for l in out[11:]], np.float32)
    assert got.shape == (257, 4)
    # eight significant digits are not always enough to name a float32 exactly (nine are): within one unit of the eighth digit
    assert np.allclose(got, x, rtol=1e-7, atol=0)
    assert out[14].split()[2] == "1e-07" and out[15].split()[0] == "1.2345679e+08"


def test_empty_cloud_is_not_written(tmp_path):
    src = tmp_path / "empty.pcd"
    src.write_text("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 0\nHEIGHT 1\n"
                   "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 0\nDATA ascii\n")
    r = subprocess.run([_exe(), "--pcd-copy", str(src), str(tmp_path) + "/", "1", ".pcd"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "save error" in r.stderr and not (tmp_path / "1.pcd").exists()   # utility.h:414-416
