"""Turns a gpurun_out/<dir> measurement set into the committed summaries under profiles/.
usage: python profiles/refresh.py gpurun_out/r1c r01
expects: bench_full.json, bench_under_rocprof.json, rocprofv3_kernel_stats.csv,
         FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv (512-scan PMC runs)"""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
for f in os.listdir(here):
    if f.startswith(tag + "_"):
        os.remove(os.path.join(here, f))
rows = [r for r in csv.DictReader(open(os.path.join(src, "rocprofv3_kernel_stats.csv"))) if "scvod::" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(here, f"{tag}_rocprofv3_kernel_stats_scvod.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfScvod", "MinNs", "MaxNs"])
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], round(100 * int(r["TotalDurationNs"]) / tot, 2),
                    r["MinNs"], r["MaxNs"]])
for name in ("bench_full", "bench_under_rocprof"):
    json.dump(json.load(open(os.path.join(src, name + ".json"))), open(os.path.join(here, f"{tag}_{name}.json"), "w"), indent=1)


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("scvod::", "")].append(float(r["Counter_Value"]))
    return d


def label(k):
    m = {"k_pw_classify": "pw_classify", "k_pw_offsets": "pw_offsets", "k_pw_scatter": "pw_scatter", "k_pw_fit": "pw_fit", "k_pw_fit_coop": "pw_fit_large",
         "k_pw_arrange": "pw_arrange", "k_emit_offsets": "emit_offsets", "k_emit": "emit", "k_vx_count": "vx_count",
         "k_vx_offsets": "vx_offsets", "k_vx_scatter": "vx_scatter", "k_vx_final_offsets": "vx_final_offsets",
         "k_vx_final": "vx_final", "k_track_probe": "track_probe", "k_track_probe_pair": "track_probe", "k_pw_sort_wave": "pw_sort_wave"}
    if k in m:
        return m[k]
    if k.startswith("k_pw_fit_coop"):
        return "pw_fit_large"
    if k.startswith("k_pw_arrange"):
        return "pw_arrange"
    if k.startswith("k_cc_link_starts"):
        return "k_cc_link_starts"
    if k.startswith("k_track_unique"):
        return "track_unique"
    if k.startswith("k_pw_order"):
        return "pw_order"
    if k.startswith("k_vx_order"):
        return "vx_order"
    for pre, l in (("k_pw_sort", "pw_sort"), ("k_vx_bucket", "vx_bucket")):
        if k.startswith(pre):
            cap = int(re.search(r"<(\d+)", k).group(1))
            return l + "_" + str(cap)
    return k


F, W = load(os.path.join(src, "FETCH_SIZE_counter_collection.csv")), load(os.path.join(src, "WRITE_SIZE_counter_collection.csv"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(os.path.join(src, c + "_counter_collection.csv"), os.path.join(here, f"{tag}_pmc_{c}_counter_collection.csv"))
scans = 512
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --scans 512 --steps 1 --warmup 0 --no-cpu --no-cpu-all --no-extras; "
               "raw values are KB per dispatch, bytes = KB*1024; FETCH doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide "
               "coalesced read; uncalibrated for gathers, so read-side numbers of gather-heavy kernels are upper bounds)",
       "scans_per_launch": scans, "kernels": {}}
by, total, extra = collections.defaultdict(float), 0.0, 0.0
for k in F:
    f = sum(F[k]) / len(F[k])
    w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0])))
    b = (2 * f + w) * 1024
    out["kernels"][k] = {"launches": len(F[k]), "fetch_KB_raw": f, "write_KB_raw": w, "hbm_bytes_per_launch_corrected": b,
                         "hbm_bytes_per_scan": b / scans}
    by[label(k)] += b / scans
    if k.startswith("k_cc_") or k in ("k_apri_expand", "k_cls_from_lists"):
        extra += b / scans  # clustering = "next" row (bench.py: prep + extras only); expansion kernels run on fetch/clustering
    else:
        total += b / scans
out["by_bench_label"], out["total_hbm_bytes_per_scan"], out["extras_hbm_bytes_per_scan"] = dict(by), total, extra
json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1)
d = json.load(open(os.path.join(src, "bench_full.json")))
print("scans/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "path_GBps")})
print("cpu", d["cpu_baseline"]["value"], "extras", d.get("extras"))
print("total HBM MB/scan", round(total / 1e6, 2))
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {k:18s} {v / 1e6:6.2f} MB/scan")
for k, v in list(d["kernels"].items())[:10]:
    print(f"  {k:18s} {v['avg_ms']:7.3f} ms")
