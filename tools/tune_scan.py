#!/usr/bin/env python3
"""Sweep the fused-scan tuning knobs (VDL_SCAN_TUNE) on Q6 in ONE process, variants interleaved
round-robin (cdna_hip_programming.md rule 24); prints median/min kernel time and GB/s per variant.
Usage: python tools/tune_scan.py [--sf sf100] [--rounds 7] [--out gpurun_out/tune.json]"""
import argparse
import itertools
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", default="sf100")
    ap.add_argument("--rows", type=int, default=0, help="row count instead of a scale factor (e.g. one of eight shards of SF100)")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--variants", default="", help="semicolon-separated VDL_SCAN_TUNE strings (overrides the default sweep)")
    args = ap.parse_args()
    import mplan2vdl_amd as m
    from mplan2vdl_amd import datagen

    n = args.rows or datagen.LINEITEM_ROWS[args.sf]
    eng = m.Engine(device=0)
    for name in datagen.Q6_COLUMNS:
        eng.generate(datagen.LINEITEM[name], 0, n)
    text = open(os.path.join(ROOT, "tests", "golden", "q6.vdl")).read()
    plan = eng.parse(text)
    plan.set_profiling(True)
    variants = []
    for bs, u in ((256, 2), (256, 4), (256, 8), (512, 2), (512, 4), (512, 8), (1024, 1), (1024, 2)):
        for nt in (0, 1):
            for chunk in (0, 1):
                for gridmul in ((1,) if args.quick else (1, 2)):
                    variants.append("u=%d,nt=%d,bs=%d,chunk=%d,gridmul=%d" % (u, nt, bs, chunk, gridmul))
    if args.variants:
        variants = [v for v in args.variants.split(";") if v]
    times = {v: [] for v in variants}
    ref = None
    for r in range(args.rounds + 1):
        for v in variants:
            os.environ["VDL_SCAN_TUNE"] = v
            out = plan.run()
            rev = out["results"]["tmp42"][".revenue"]
            ref = ref or rev
            assert rev == ref, (v, rev, ref)
            if r > 0:                                   # round 0 = warm-up
                times[v].append(plan.scan_stats()[2])
    rows = []
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        rows.append({"variant": v, "median_us": med, "min_us": mn, "gbps_median": n * 28 / med / 1e3, "gbps_best": n * 28 / mn / 1e3,
                     "label": [k for k in out["timings"]][0] if False else None})
    rows.sort(key=lambda x: x["median_us"])
    for x in rows:
        print("%-40s median %8.1f us  %7.1f GB/s   best %8.1f us %7.1f GB/s" % (x["variant"], x["median_us"], x["gbps_median"], x["min_us"], x["gbps_best"]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"rows": n, "rounds": args.rounds, "results": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
