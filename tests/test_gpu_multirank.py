"""The N > 1 job of bench.py on ONE device: two ranks (gloo, device buffers staged through the host) shard 48 scans of a
sequence in round-robin blocks, exchange the boundary tables, track, accumulate their static maps and reduce them on
rank 0 -- per-scan dynamic counts and the merged map must equal the single-rank run bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, name, extra):
    out = os.path.join(tmp, name + ".npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu", "--no-extras", "--backend", "gloo",
           "--same-device", "--kind", "PARK", "--preset", "parkinglot", "--dump-map", out] + extra
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    return json.loads(line), np.load(out)


def test_two_ranks_reproduce_the_single_rank_job(tmp_path):
    one, m1 = _run(str(tmp_path), "one", ["--gpus", "1", "--scans", "48", "--blocks-per-rank", "4"])
    two, m2 = _run(str(tmp_path), "two", ["--gpus", "2", "--scans", "24", "--blocks-per-rank", "2"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2
    assert two["config"]["boundary_tables_per_step"] >= 1
    assert np.array_equal(m1["scans"], m2["scans"])
    assert np.array_equal(m1["dynamic_points"], m2["dynamic_points"])
    assert m1["dynamic_points"].sum() > 0
    assert np.array_equal(m1["keys"], m2["keys"]) and np.array_equal(m1["vals"], m2["vals"])
    assert len(m1["keys"]) == one["config"]["static_map_cells"] == two["config"]["static_map_cells"]
