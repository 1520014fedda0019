"""development: one line per build of tools/ab_bench.sh -- step time and the kernels whose time moved"""
import json
import sys

d = sys.argv[1]
rows = {}
for v in sys.argv[2:]:
    try:
        rows[v] = json.loads([l for l in open(f"{d}/bench_{v}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(v, "ERR", e)
names = list(rows)
if names:
    keys = sorted({k for r in rows.values() for k in r["kernels"]}, key=lambda k: -max(r["kernels"].get(k, {}).get("avg_ms", 0) for r in rows.values()))
    print("%-18s" % "kernel", *["%10s" % n for n in names])
    print("%-18s" % "ms_per_step", *["%10.2f" % rows[n]["ms_per_step"] for n in names])
    print("%-18s" % "scans/s", *["%10.0f" % rows[n]["value"] for n in names])
    for k in keys:
        print("%-18s" % k, *["%10.3f" % rows[n]["kernels"].get(k, {}).get("avg_ms", float("nan")) for n in names])
    print("%-18s" % "sum", *["%10.2f" % sum(x["avg_ms"] for x in rows[n]["kernels"].values()) for n in names])
