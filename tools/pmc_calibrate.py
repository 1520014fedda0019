#!/usr/bin/env python3
"""rocprofv3 FETCH_SIZE / WRITE_SIZE calibrated against microkernels with known bytes (tools/pmc_calibrate.hip), on the GPU box:
    python tools/pmc_calibrate.py gpurun_out/pmc_cal
Three runs of the binary: plain (hipEvent times), under `rocprofv3 --pmc FETCH_SIZE`, under `rocprofv3 --pmc WRITE_SIZE` (separate
passes: the two do not fit the TCC slots together; no trace domains beside the counters).  Writes calibration.json and
calibration.md (copied to profiles/r06_pmc_calibration.{json,md}); profiles/refresh.py reads the factors from the JSON."""
import collections
import csv
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_cal")
os.makedirs(out, exist_ok=True)
exe = os.path.join(HERE, "pmc_calibrate")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", exe, os.path.join(HERE, "pmc_calibrate.hip")])
env = dict(os.environ, TMPDIR="/tmp")
plain = json.loads([l for l in subprocess.run([exe], capture_output=True, text=True, check=True, cwd="/tmp", env=env).stdout.splitlines() if l.startswith("{")][-1])
counters = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"/tmp/pmc_cal_{c}"
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--pmc", c, "--kernel-include-regex", "cal_", "--output-format", "csv", "-d", d, "--", exe], capture_output=True, text=True, cwd="/tmp", env=env)
    open(os.path.join(out, f"rocprof_{c}.err"), "w").write(r.stderr[-4000:])
    f = subprocess.run(["find", d, "-name", "*counter_collection.csv"], capture_output=True, text=True).stdout.split()
    acc = collections.defaultdict(list)
    if f:
        for row in csv.DictReader(open(f[0])):
            if row["Counter_Name"] == c:
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                acc[k].append(float(row["Counter_Value"]))
        subprocess.run(["cp", f[0], os.path.join(out, f"{c}_counter_collection.csv")])
    counters[c] = {k: sum(v) / len(v) for k, v in acc.items()}  # (two dispatches per kernel: warm + timed)
rows = []
for r in plain["rows"]:
    k = r["kernel"]
    f_kb, w_kb = counters["FETCH_SIZE"].get(k), counters["WRITE_SIZE"].get(k)
    row = dict(r)
    row["FETCH_SIZE_KB_raw"], row["WRITE_SIZE_KB_raw"] = f_kb, w_kb
    row["fetch_bytes_raw"] = None if f_kb is None else f_kb * 1024
    row["write_bytes_raw"] = None if w_kb is None else w_kb * 1024
    row["known_over_fetch_raw"] = (r["known_read_bytes"] / (f_kb * 1024)) if f_kb else None
    row["known_over_write_raw"] = (r["known_write_bytes"] / (w_kb * 1024)) if w_kb else None
    if r["accesses"]:
        row["fetch_raw_bytes_per_access"] = (f_kb * 1024 / r["accesses"]) if f_kb is not None else None
        row["write_raw_bytes_per_access"] = (w_kb * 1024 / r["accesses"]) if w_kb is not None else None
    row["GBps_known_bytes"] = (r["known_read_bytes"] + r["known_write_bytes"]) / (r["ms"] * 1e-3) / 1e9
    rows.append(row)
by = {(r["kernel"].split("<")[0], r["working_set"]): r for r in rows}


def g(name, ws, key):
    return (by.get((name, ws)) or {}).get(key)


summary = {
    "fetch_factor_coalesced_16B": g("cal_read16", "2GiB", "known_over_fetch_raw"),
    "fetch_factor_coalesced_4B": g("cal_read4", "2GiB", "known_over_fetch_raw"),
    "write_factor_coalesced_16B": g("cal_write16", "2GiB", "known_over_write_raw"),
    "fetch_raw_bytes_per_random_4B_gather_2GiB": g("cal_gather4", "2GiB", "fetch_raw_bytes_per_access"),
    "fetch_raw_bytes_per_random_4B_gather_128MiB": g("cal_gather4", "128MiB", "fetch_raw_bytes_per_access"),
    "fetch_raw_bytes_per_gather_PAIR_in_one_128B_line_2GiB": g("cal_gatherpair4", "2GiB", "fetch_raw_bytes_per_access"),
    "fetch_raw_bytes_per_gather_PAIR_in_one_128B_line_128MiB": g("cal_gatherpair4", "128MiB", "fetch_raw_bytes_per_access"),
    "fetch_raw_bytes_per_atomicmin64_2GiB": g("cal_atomic64", "2GiB", "fetch_raw_bytes_per_access"),
    "write_raw_bytes_per_atomicmin64_2GiB": g("cal_atomic64", "2GiB", "write_raw_bytes_per_access"),
    "fetch_raw_bytes_per_atomicmin64_128MiB": g("cal_atomic64", "128MiB", "fetch_raw_bytes_per_access"),
    "write_raw_bytes_per_atomicmin64_128MiB": g("cal_atomic64", "128MiB", "write_raw_bytes_per_access"),
    "copy_ceiling_GBps_2GiB": g("cal_copy16", "2GiB", "GBps_known_bytes"),
    "read_ceiling_GBps_2GiB": g("cal_read16", "2GiB", "GBps_known_bytes"),
    "write_ceiling_GBps_2GiB": g("cal_write16", "2GiB", "GBps_known_bytes"),
    "read_GBps_128MiB_16_passes": g("cal_read16", "128MiB", "GBps_known_bytes"),
}
small, large = g("cal_read16", "128MiB", "known_over_fetch_raw"), g("cal_read16", "2GiB", "known_over_fetch_raw")
if small and large:
    # 16 passes over 128 MiB: only the first can come from HBM.  If the counter saw (about) all 16 passes, Infinity-Cache hits are counted.
    summary["infinity_cache_hits_counted_in_FETCH_SIZE"] = bool(small < 2.0 * large)
    summary["read16_128MiB_counted_fraction_of_the_passes"] = large / small
pair, single = summary["fetch_raw_bytes_per_gather_PAIR_in_one_128B_line_2GiB"], summary["fetch_raw_bytes_per_random_4B_gather_2GiB"]
if pair and single:
    # one request per pair: the gather fetched the whole 128-B line and the raw counter tallied it at 64 B (factor 2); two requests: 64-B sectors (factor 1)
    summary["fetch_factor_random_gather"] = 2.0 if pair < 1.5 * single else 1.0
    summary["random_gather_true_bytes_per_access"] = 128 if pair < 1.5 * single else 64
json.dump({"rows": rows, "summary": summary}, open(os.path.join(out, "calibration.json"), "w"), indent=1)


def fm(v, s="%.1f"):
    return "-" if v is None else s % v


with open(os.path.join(out, "calibration.md"), "w") as f:
    f.write("# rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 against known bytes (tools/pmc_calibrate.py)\n\n"
            "Microkernels of tools/pmc_calibrate.hip, one dispatch each (average of the warm and the timed dispatch), separate `--pmc` passes.  raw = counter value x 1024 B.\n"
            "known / raw = the factor a raw counter has to be multiplied by for that access pattern.  128 MiB sets sit inside the 256 MiB Infinity Cache and are passed over 16 times.\n\n")
    f.write("| kernel | working set | ms | known read MB | FETCH raw MB | known/raw | known write MB | WRITE raw MB | known/raw | raw B/access (F, W) | GB/s (known) |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['kernel']}` | {r['working_set']} | {r['ms']:.3f} | {r['known_read_bytes'] / 1e6:.0f} | {fm(r['fetch_bytes_raw'] and r['fetch_bytes_raw'] / 1e6)} | {fm(r['known_over_fetch_raw'], '%.3f')} | "
                f"{r['known_write_bytes'] / 1e6:.0f} | {fm(r['write_bytes_raw'] and r['write_bytes_raw'] / 1e6)} | {fm(r['known_over_write_raw'], '%.3f')} | "
                f"{fm(r.get('fetch_raw_bytes_per_access'))}, {fm(r.get('write_raw_bytes_per_access'))} | {r['GBps_known_bytes']:.0f} |\n")
    f.write("\n## Summary\n\n")
    for k, v in summary.items():
        f.write(f"* `{k}`: {v if not isinstance(v, float) else round(v, 3)}\n")
print(json.dumps(summary, indent=1))
