#!/usr/bin/env python3
"""Every TPC-H plan the front end compiles, on the GPU over a synthetic catalog at a chosen scale
(fraction of the SF10 catalog the metadata describes): wall time per query, summed per-statement kernel
time, the slowest statements, and the oracle's single-core time on the same columns for comparison.
    python tools/run_plans.py [scale=0.05] [plans...]
RUN_PLANS_CLUSTERED=1: lineitem comes clustered by order (its join index into orders is non-decreasing), as dbgen writes it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
plans = [int(x) for x in sys.argv[2:]] or [1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20]
check = os.environ.get("RUN_PLANS_ORACLE", "1") == "1"
cfg = frontend.load_metadata(META)
print("scale %g of SF10: lineitem %d rows" % (scale, catalog.scaled_rows(59986052, scale)))
for n in plans:
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=scale, clustered=("lineitem.lineitem_orders",) if os.environ.get("RUN_PLANS_CLUSTERED") == "1" else ())
    in_bytes = sum(v.nbytes for v in cols.values())
    e = m.Engine(0)
    for k, v in cols.items():
        e.upload(k, v)
    plan = e.parse(text)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); out = plan.run(); best = min(best, time.perf_counter() - t0)
    plan.set_profiling(True)
    prof = plan.run()
    kern = sum(prof["timings"].values()) / 1e3
    top = sorted(prof["timings"].items(), key=lambda kv: -kv[1])[:int(os.environ.get("RUN_PLANS_TOP", "3"))]
    line = "Q%02d %3d stmts fused=%d in=%7.1f MB  wall %8.2f ms  kernels %8.2f ms  out rows %d" % (
        n, len(text.splitlines()), plan.is_fused, in_bytes / 1e6, best * 1e3, kern, max(len(list(v.values())[0]) for v in out["results"].values()))
    if check:
        from helpers import oracle_run
        t0 = time.perf_counter(); want = oracle_run(text, cols); cpu = time.perf_counter() - t0
        line += "  oracle %8.1f ms  %s" % (cpu * 1e3, "ok" if want == out["results"] else "MISMATCH")
    print(line, " top:", ", ".join("%s %.0fus" % (k.replace("timeInMicrosecondsForStatement", ""), v) for k, v in top), flush=True)
    e.close()
