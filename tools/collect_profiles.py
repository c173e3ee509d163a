#!/usr/bin/env python3
"""Copies what tools/profile_bench.sh left under gpurun_out/profile_bench/ into profiles/<round>/ (kernel stats, the JSON lines
of the runs, the scan kernels' PMC rows, traffic.json -> also profiles/traffic.json) and prints the figures the README quotes.
    python tools/collect_profiles.py r02"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "profile_bench"), os.path.join(ROOT, "profiles", sys.argv[1])
os.makedirs(dst, exist_ok=True)
for q in ("q6", "q6late", "q1", "default"):
    shutil.copy(os.path.join(src, q + "_kernel_stats.csv"), os.path.join(dst, "bench_sf100_%s_kernel_stats.csv" % q))
    shutil.copy(os.path.join(src, q + "_bench.json"), os.path.join(dst, "bench_sf100_%s_under_rocprof.json" % q))
    d = json.loads(open(os.path.join(src, q + "_bench.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(q, r["kernel"], "events %.1f us" % r["kernel_us"], "frac %.4f" % r["frac"], "of traffic", r.get("frac_of_traffic"), "verified", d["verified_bit_exact_vs_cpu"],
          "ms/step %.4f" % d["ms_per_step"])
    rows = list(csv.DictReader(open(os.path.join(src, q + "_kernel_stats.csv"))))
    for row in rows[:3]:
        if "jit" in row["Name"] or "k_scan<" in row["Name"]:
            print("   rocprof:", row["Name"][:60], row["Calls"], "launches, avg", row["AverageNs"], "ns")
    if q == "default":
        for k, v in d.get("also", {}).items():
            print("   also", k, {x: v.get(x) for x in ("kernel_us", "roofline_frac", "ms_per_query_results_left_in_hbm", "ms_per_query")})
        print("   tuner:", d["scan_kernels"]["note"])
for t in ("q6_fetch", "q6_write", "q1_fetch", "q6late_fetch"):
    f = glob.glob(os.path.join(src, t, "*", "*counter_collection.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "k_scan<" in r["Kernel_Name"] or "vdl_jit_mscan" in r["Kernel_Name"] or "k_mscan<" in r["Kernel_Name"]]
    w = csv.DictWriter(open(os.path.join(dst, t + "_pmc.csv"), "w"), fieldnames=rows[0].keys())
    w.writeheader(); w.writerows(keep)
shutil.copy(os.path.join(src, "traffic.json"), os.path.join(dst, "traffic.json"))
shutil.copy(os.path.join(src, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
print(json.dumps({k: (v["kernel"], v["hbm_bytes_per_launch"], round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"], 5)) for k, v in json.load(open(os.path.join(src, "traffic.json"))).items()}, indent=1))
