"""gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) -> profiles/<tag>_kernel_stats.csv, <tag>_pmc_{fetch,write}_size.csv,
<tag>_bench_line*.json and profiles/pmc_traffic.json (HBM bytes per launch of every kernel of this library:
(2 x FETCH_SIZE + WRITE_SIZE) x 1024, the factor 2 per MI355X_MICROARCH.md's HBM section for gfx950)."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    m = re.search(r"cabac::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def per_kernel(path):
    acc = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) < 4 or row[2] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            acc.setdefault(short(row[1]), []).append(float(row[3]))
    return acc


fetch, write = per_kernel(os.path.join(src, "pmc_fetch.csv")), per_kernel(os.path.join(src, "pmc_write.csv"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_fetch.csv"), os.path.join(dst, tag + "_pmc_fetch_size.csv"))
shutil.copy(os.path.join(src, "pmc_write.csv"), os.path.join(dst, tag + "_pmc_write_size.csv"))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, tag + "_bench_line.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_line_under_rocprof.json"))
if os.path.exists(os.path.join(src, "main_kernel_stats.csv")):   # the same without the residual leg: the timed kernels only
    shutil.copy(os.path.join(src, "main_kernel_stats.csv"), os.path.join(dst, tag + "_main_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_main_under_rocprof.json"), os.path.join(dst, tag + "_main_bench_line_under_rocprof.json"))

# launches of one kernel differ by role in bench.py (e.g. residual sizes pass vs records pass): keep every distinct value
# group — (launch index modulo the pattern is not reconstructed here) — as the median of the timed-region launches and the
# full list
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    n = min(len(f), len(w)) if f and w else max(len(f), len(w))
    rows = []
    for i in range(n):
        fi = f[i] if i < len(f) else 0.0
        wi = w[i] if i < len(w) else 0.0
        rows.append({"FETCH_SIZE_KiB_raw": fi, "WRITE_SIZE_KiB": wi, "hbm_bytes": int((2 * fi + wi) * 1024)})
    kernels[k] = rows
out = {"note": "HBM bytes per launch from rocprofv3 PMC counters, separate passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE with "
               "--kernel-trace only): bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 tallies 128-B read requests at 64 B, "
               "MI355X_MICROARCH.md HBM section).  Launches in bench.py order; tag " + tag + ".",
       "command": "tools/collect_profiles.sh " + tag,
       "launches": kernels}
json.dump(out, open(os.path.join(dst, "pmc_traffic_" + tag + ".json"), "w"), indent=1)
for k, rows in kernels.items():
    print("%-44s launches %3d  last: fetch x2 %.1f MB  write %.1f MB  total %.1f MB" %
          (k, len(rows), 2 * rows[-1]["FETCH_SIZE_KiB_raw"] / 1024 * 1.048576, rows[-1]["WRITE_SIZE_KiB"] / 1024 * 1.048576, rows[-1]["hbm_bytes"] / 1e6))

# bench.py's `traffic` fields read profiles/pmc_traffic.json: refresh the C4 entries of the kernels measured in this run
path = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(path)) if os.path.exists(path) else {"workloads": {}}
c4 = t.setdefault("workloads", {}).setdefault("C4", {})


def entry(rows):
    return {"FETCH_SIZE_KiB_raw": rows["FETCH_SIZE_KiB_raw"], "WRITE_SIZE_KiB": rows["WRITE_SIZE_KiB"], "hbm_bytes_per_launch": rows["hbm_bytes"],
            "measured": tag}


def first(prefix):
    for k, rows in kernels.items():
        if k.startswith(prefix) and rows:
            return rows[0]      # the first launch is the C4 batch of the timed region (later ones: the residual leg's substreams)
    return None


for name, prefix in (("encode_kernel_v6", "encode_kernel_v6"), ("encode_kernel_v7", "encode_kernel_v7"), ("decode_kernel_v4", "decode_kernel_v4"), ("estimate_kernel", "estimate_kernel"),
                     ("residual_parse_kernel", "residual_parse_kernel")):
    r = first(prefix)
    if r:
        c4[name] = entry(r)
for name, flag in (("residual_kernel_count", "residual_kernel<false"), ("residual_kernel_write", "residual_kernel<true")):
    rows = [v[-1] for k, v in kernels.items() if k.startswith(flag) and v]
    if rows:
        c4[name] = {"FETCH_SIZE_KiB_raw": sum(r["FETCH_SIZE_KiB_raw"] for r in rows), "WRITE_SIZE_KiB": sum(r["WRITE_SIZE_KiB"] for r in rows),
                    "hbm_bytes_per_launch": sum(r["hbm_bytes"] for r in rows), "measured": tag,
                    "note": "upper bound: FETCH_SIZE x 2 over-counts the direct variant's 64-byte requests (DESIGN.md section 5)"}
t["note_" + tag] = "entries with measured = %s: tools/collect_profiles.sh %s, summarised by tools/pmc_summary.py" % (tag, tag)
json.dump(t, open(path, "w"), indent=1)
