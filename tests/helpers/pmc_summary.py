"""Summarise a rocprofv3 --pmc counter_collection.csv per scvod kernel (mean per dispatch)."""
import collections
import csv
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("scvod::", "")
    d[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = sorted({c for v in d.values() for c in v})
print("kernel".ljust(36), " ".join(c.replace("SQ_", "")[:14].rjust(14) for c in cols))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1.0
    out = []
    for c in cols:
        val = m.get(c, 0.0)
        out.append((f"{val:14.3e}" if c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES") else f"{val / wc:14.3f}"))
    print(k[:36].ljust(36), " ".join(out))
