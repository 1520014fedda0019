"""Turns a gpurun_out/<dir> measurement set (profiles/measure.sh) into the committed summaries under profiles/.
usage: python profiles/refresh.py gpurun_out/r2 r02
expects: bench_full.json, bench_park.json, bench_os128.json, bench_under_rocprof_{k64,park,os128}.json,
         rocprofv3_kernel_stats_{k64,park,os128}.csv, FETCH_SIZE / WRITE_SIZE / SQ _counter_collection.csv"""
import collections
import csv
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
GENERATED = ("_bench_", "_rocprofv3_kernel_stats_", "_pmc_traffic_", "_sq_pmc_summary", "_kernel_phases", "_kernel_table")
for f in os.listdir(here):
    if f.startswith(tag + "_") and any(g in f for g in GENERATED):  # (only what this script writes: notes, calibration and sweeps of the round stay)
        os.remove(os.path.join(here, f))


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


for name in ("k64", "park", "os128"):
    p = os.path.join(src, f"rocprofv3_kernel_stats_{name}.csv")
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if "scvod::" in r["Name"]]
    tot = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    with open(os.path.join(here, f"{tag}_rocprofv3_kernel_stats_{name}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfScvod", "MinNs", "MaxNs"])
        for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], round(100 * int(r["TotalDurationNs"]) / tot, 2), r["MinNs"], r["MaxNs"]])
for name in ("bench_full", "bench_park", "bench_os128", "bench_under_rocprof_k64", "bench_under_rocprof_park", "bench_under_rocprof_os128"):
    p = os.path.join(src, name + ".json")
    d = last_json(p) if os.path.exists(p) else None
    if d is not None:
        json.dump(d, open(os.path.join(here, f"{tag}_{name}.json"), "w"), indent=1)


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("scvod::", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


LABELS = {"k_pw_classify": "pw_classify", "k_pw_offsets": "pw_offsets", "k_pw_scatter": "pw_scatter", "k_pw_fit": "pw_fit", "k_emit_offsets": "emit_offsets",
          "k_emit": "emit", "k_vx_partition": "vx_partition", "k_vx_final_offsets": "vx_final_offsets",
          "k_vx_final": "vx_final", "k_pw_sort_wave": "pw_sort_wave", "k_cc_scan": "cc_scan", "k_tk_init": "tk_init", "k_tk_probe": "tk_probe", "k_tk_dyn": "tk_dyn",
          "k_map_accumulate": "map_accumulate", "k_cc_exact": "cc_exact", "k_cc_order": "cc_scan"}


def label(k):
    if k in LABELS:
        return LABELS[k]
    for pre, l in (("k_pw_fit_coop", "pw_fit_large"), ("k_pw_arrange", "pw_arrange"), ("k_pw_order", "pw_order"), ("k_vx_order", "vx_order"), ("k_tk_decide", "tk_decide")):
        if k.startswith(pre):
            return l
    if k.startswith("k_cc_lastname"):  # the three passes of scvod_lastname.hip by table size (<1792, 1024>: the first pass of 128-beam batches)
        m = re.search(r"<(\d+), *(\d+)", k)
        cap, th = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        return "cc_lastname_huge" if cap > 16384 else ("cc_lastname_big" if cap > 4096 else ("cc_lastname_mid" if cap > 1024 and th == 256 else "cc_lastname"))
    for pre, l in (("k_pw_sort", "pw_sort"), ("k_vx_bucket", "vx_bucket")):
        if k.startswith(pre):
            return l + "_" + str(int(re.search(r"<(\d+)", k).group(1)))
    return k


LABELS.update({"k_tk_chain": "tk_chain", "k_tk_chain_cmp": "tk_chain_fix", "k_tk_chain_fix": "tk_chain_fix", "k_tk_chain_spec": "tk_chain_fix"})
# the counters against known bytes (tools/pmc_calibrate.py -> profiles/*_pmc_calibration.json): factors and the measured ceilings
CAL = {}
for f in sorted(os.listdir(here)):
    if f.endswith("_pmc_calibration.json"):
        CAL = json.load(open(os.path.join(here, f))).get("summary", {})
FETCH_FACTOR = float(CAL.get("fetch_factor_coalesced_16B") or 2.0)  # (gathers: the same factor, see the calibration notes)
WRITE_FACTOR = float(CAL.get("write_factor_coalesced_16B") or 1.0)
COPY_CEIL = float(CAL.get("copy_ceiling_GBps_2GiB") or 4600.0)
READ_CEIL = float(CAL.get("read_ceiling_GBps_2GiB") or 6000.0)
HBM_BOUND_GBPS = 0.6 * COPY_CEIL
total_by = {}
for name in ("k64", "park", "os128"):
    fp, wp = os.path.join(src, f"FETCH_SIZE_counter_collection_{name}.csv"), os.path.join(src, f"WRITE_SIZE_counter_collection_{name}.csv")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    scans = int(open(os.path.join(src, f"pmc_scans_{name}.txt")).read().split()[0])
    F, W = load(fp), load(wp)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py [workload] --steps 1 --warmup 0 --no-cpu --no-extras; raw values are "
                   "KB per dispatch, bytes = KB * 1024; FETCH x 2 and WRITE x 1 as calibrated on known bytes (profiles/r06_pmc_calibration.md: coalesced reads of any width AND "
                   "random gathers -- a gather moves its whole 128-B line; an atomic without return counts 32 B on the write side only); Infinity-Cache hits are counted: this is "
                   "L2-fabric traffic, an upper bound of HBM traffic; a kernel launched several times per step is summed",
           "workload": name, "scans_per_launch": scans, "kernels": {}}
    by, total = collections.defaultdict(float), 0.0
    # the profiled command runs its step twice (the timed region and the hipEvent attribution pass): per-step = sum / passes
    passes = max(1, len(F.get("k_pw_classify", {}).get("FETCH_SIZE", [0.0])))
    out["passes_in_the_profiled_run"] = passes
    for k in F:
        f = sum(F[k]["FETCH_SIZE"]) / passes
        w = sum(W.get(k, {}).get("WRITE_SIZE", [0.0])) / passes
        b = (FETCH_FACTOR * f + WRITE_FACTOR * w) * 1024
        out["kernels"][k] = {"launches_per_step": len(F[k]["FETCH_SIZE"]) / passes, "fetch_KB_raw_per_step": f, "write_KB_raw_per_step": w, "hbm_bytes_per_step_corrected": b,
                             "hbm_bytes_per_scan": b / scans}
        by[label(k)] += b / scans
        total += b / scans
    out["by_bench_label"], out["total_hbm_bytes_per_scan"] = dict(by), total
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic_{name}.json"), "w"), indent=1)
    total_by[name] = (total, dict(by))
total, by = total_by.get("k64", (0.0, {}))

# SQ counters: share of the wave cycles that issued an instruction / a VALU / an LDS instruction, that waited, bank conflicts
p = os.path.join(src, "SQ_counter_collection.csv")
if os.path.exists(p):
    S = load(p)
    with open(os.path.join(here, f"{tag}_sq_pmc_summary.txt"), "w") as f:
        f.write("rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT\n"
                "  --kernel-include-regex scvod -- python bench.py --scans 256 --steps 1 --warmup 0 --no-cpu --no-extras   (sums over the dispatches of a kernel)\n"
                "shares are of SQ_WAVE_CYCLES: issue = ACTIVE_INST_ANY, valu / lds = ACTIVE_INST_VALU / _LDS, wait = WAIT_ANY (s_waitcnt / barrier), stall = WAIT_INST_ANY;\n"
                "conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS\n\n")
        f.write(f"{'kernel':44s} {'waves':>10s} {'issue':>7s} {'valu':>7s} {'lds':>7s} {'wait':>7s} {'stall':>7s} {'conflict':>9s}\n")
        rows = []
        for k, c in S.items():
            g = lambda n: sum(c.get(n, [0.0]))
            wc = g("SQ_WAVE_CYCLES") or 1.0
            rows.append((g("SQ_WAVE_CYCLES"), k, g("SQ_WAVES"), g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc,
                         g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_LDS_BANK_CONFLICT") / (g("SQ_ACTIVE_INST_LDS") or 1.0)))
        for r in sorted(rows, reverse=True):
            f.write(f"{r[1][:44]:44s} {r[2]:10.0f} {r[3]:7.3f} {r[4]:7.3f} {r[5]:7.3f} {r[6]:7.3f} {r[7]:7.3f} {r[8]:9.3f}\n")

p = os.path.join(src, "kernel_phases.txt")
if os.path.exists(p):
    open(os.path.join(here, f"{tag}_kernel_phases.txt"), "w").write(
        "tools/kernel_phases.py on the profiling build (make -C dr-using-scv-od_amd/csrc prof): 100 MHz wall clock between phase marks, thread 0 of every\n"
        "workgroup behind a barrier, summed over the workgroups of a kernel and divided by the scans\n\n" + open(p).read())
d = last_json(os.path.join(src, "bench_full.json"))

# the per-kernel table DESIGN.md section 4 points at: one row per bench label of the K64 line
if d is not None:
    sq = {}
    if os.path.exists(os.path.join(src, "SQ_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for k, c in load(os.path.join(src, "SQ_counter_collection.csv")).items():
            for n, vals in c.items():
                acc[label(k)][n] += sum(vals)
        for l, c in acc.items():
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            sq[l] = (c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, c.get("SQ_WAIT_ANY", 0.0) / wc,
                     c.get("SQ_LDS_BANK_CONFLICT", 0.0) / (c.get("SQ_ACTIVE_INST_LDS", 0.0) or 1.0))
    n_sc = d["config"]["scans_per_rank"]
    with open(os.path.join(here, f"{tag}_kernel_table.md"), "w") as f:
        f.write(f"Per-kernel table of the K64 bench line ({n_sc} scans per step; generated by profiles/refresh.py from bench_full.json, the PMC passes and the SQ pass).\n"
                "ms: hipEvents around the kernel, average per step of the attribution pass.  MB/scan and GB/s: the kernel's OWN L2-fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate\n"
                "rocprofv3 --pmc passes; factors calibrated on known bytes, Infinity-Cache hits included: an upper bound of its HBM bytes) over its OWN time.  issue / valu / lds / wait:\n"
                "shares of SQ_WAVE_CYCLES; conflicts: LDS bank conflicts per LDS instruction.\n"
                f"Measured ceilings of this box (tools/pmc_calibrate.py, 2 GiB sets): read {READ_CEIL:.0f} GB/s, copy (read + write) {COPY_CEIL:.0f} GB/s; spec 8000.  bound: hbm only when the kernel\n"
                f"moves >= 0.6 x the copy ceiling = {HBM_BOUND_GBPS:.0f} GB/s; lds when the LDS share > 0.05 and it waits < 0.5; issue when > 0.45 of the cycles issue (or VALU share x resident\n"
                "waves says so: see DESIGN 8); latency otherwise.\n\n")
        f.write("| kernel | ms / step | share | MB / scan | GB/s | issue | valu | lds | wait | conflicts | bound |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for name, k in d["kernels"].items():
            mb = k.get("pmc_MB_per_scan")
            gb = k.get("pmc_GBps")
            q = sq.get(name)
            bound = "-"
            if q:
                bound = "hbm" if (gb or 0) >= HBM_BOUND_GBPS else ("lds" if q[2] > 0.05 and q[3] < 0.5 else ("issue" if q[0] > 0.45 else "latency"))
            elif gb:
                bound = "hbm" if gb >= HBM_BOUND_GBPS else "latency"
            f.write(f"| `{name}` | {k['avg_ms']:.3f} | {100 * k['share']:.1f} % | {('%.2f' % mb) if mb is not None else '-'} | {('%.0f' % gb) if gb else '-'} | "
                    + (" | ".join(f"{v:.2f}" for v in q[:4]) + f" | {q[4]:.1f}" if q else "- | - | - | - | -") + f" | {bound} |\n")
print("scans/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "roofline frac", round(d["roofline"]["frac"], 4))
print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], "quality", d.get("quality") and {k: d["quality"][k] for k in ("delta_PR", "delta_RR")})
print("total HBM MB/scan", round(total / 1e6, 2), "algorithmic MB/scan", round(d["roofline"]["algorithmic_bytes_per_scan"] / 1e6, 2))
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:18s} {v / 1e6:6.2f} MB/scan")
