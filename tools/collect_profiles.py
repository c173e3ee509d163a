#!/usr/bin/env python3
"""Copies what tools/profile_bench.py left under gpurun_out/profile_bench/ into profiles/<round>/ (kernel stats, the JSON lines of
the runs, the scan kernels' PMC rows, the FETCH_SIZE calibration, traffic.json -> also profiles/traffic.json) and prints the
figures the README quotes.  Refuses a PMC file that names a kernel its kernel-stats file never saw (a stale or mismatched pass).
    python tools/collect_profiles.py r03"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "profile_bench"), os.path.join(ROOT, "profiles", sys.argv[1])
os.makedirs(dst, exist_ok=True)
traffic = json.load(open(os.path.join(src, "traffic.json")))
for tag in ("q6_tuned", "q6_sibling", "q6_everything", "q1_tuned", "q1_sibling", "q1_everything"):
    if not os.path.exists(os.path.join(src, tag + "_kernel_stats.csv")):
        continue
    stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, tag + "_kernel_stats.csv")))}
    for r in csv.DictReader(open(os.path.join(src, tag + "_fetch_pmc.csv"))):
        if r["Kernel_Name"] not in stats:
            raise SystemExit("%s_fetch_pmc.csv names %r, which %s_kernel_stats.csv never saw" % (tag, r["Kernel_Name"], tag))
    for f in (tag + "_kernel_stats.csv", tag + "_fetch_pmc.csv"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "bench_sf100_" + f))
    shutil.copy(os.path.join(src, tag + "_bench.json"), os.path.join(dst, "bench_sf100_%s_under_rocprof.json" % tag))
for q in ("q6", "q1"):
    shutil.copy(os.path.join(src, q + "_plain.json"), os.path.join(dst, "bench_sf100_%s_plain_unprofiled.json" % q))
shutil.copy(os.path.join(src, "default_kernel_stats.csv"), os.path.join(dst, "bench_sf100_default_kernel_stats.csv"))
shutil.copy(os.path.join(src, "default_bench.json"), os.path.join(dst, "bench_sf100_default_under_rocprof.json"))
if os.path.exists(os.path.join(src, "fetch_calib.txt")):
    shutil.copy(os.path.join(src, "fetch_calib.txt"), os.path.join(dst, "fetch_calib.txt"))
shutil.copy(os.path.join(src, "traffic.json"), os.path.join(dst, "traffic.json"))
shutil.copy(os.path.join(src, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
for label, t in traffic.items():
    print("%-62s %s" % (label, t["rocprof_kernel"][:70]))
    print("    rocprof avg %.1f us (%d calls) | bench events %.1f us | FETCH_SIZE x2 %.5g B | counted %.5g B (ratio %.5f) | algorithmic %.5g B | frac of 8 TB/s: %.4f (rocprof time), %.4f (event time)" % (
        t["rocprof_avg_ns"] / 1e3, t["rocprof_calls"], t["bench_kernel_us"], t["hbm_bytes_per_launch"], t["bytes_moved_per_launch_counted"],
        t["hbm_bytes_per_launch"] / t["bytes_moved_per_launch_counted"], t["algorithmic_bytes_per_launch"],
        t["hbm_bytes_per_launch"] / (t["rocprof_avg_ns"] * 1e-9) / 8e12, t["hbm_bytes_per_launch"] / (t["bench_kernel_us"] * 1e-6) / 8e12))
d = json.loads([l for l in open(os.path.join(src, "default_bench.json")).read().splitlines() if l.startswith("{")][-1])
print("default run under the kernel trace:", d["roofline"]["kernel"], "frac %.4f" % d["roofline"]["frac"], "| read-everything:", d["roofline"]["read_everything_kernel"])
for k, v in d.get("also", {}).items():
    print("   also", k, {x: v.get(x) for x in ("kernel_us", "roofline_frac", "ms_per_query_results_left_in_hbm", "ms_per_query", "error") if x in v})
print("   tuner:", d["scan_kernels"]["note"])
