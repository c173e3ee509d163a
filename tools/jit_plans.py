#!/usr/bin/env python3
"""The fused TPC-H plans at a chosen scale with run-time specialisation and the tuner on: which form each scan settles on and its time.
    python3 tools/jit_plans.py [scale=0.3] [plans...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend
from helpers import oracle_run
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
plans = [int(x) for x in sys.argv[2:]] or [1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20]
clustered = ("lineitem.lineitem_orders",) if os.environ.get("RUN_PLANS_CLUSTERED", "1") == "1" else ()
cfg = frontend.load_metadata(META)
for n in plans:
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=scale, clustered=clustered)
    e = m.Engine(0)
    for k, v in cols.items():
        e.upload(k, v)
    p = e.parse(text)
    p.set_profiling(True)
    p.set_jit(True, tune=True)
    out = None
    for _ in range(6):
        out = p.run()
    ok = out["results"] == oracle_run(text, cols) if os.environ.get("JIT_PLANS_ORACLE", "1") == "1" else None
    t = {k.replace("timeInMicrosecondsFor", "")[:70]: round(v) for k, v in out["timings"].items() if "Scan" in k or "Front" in k}
    total = sum(out["timings"].values()) / 1e3
    t0 = time.perf_counter()
    p.set_profiling(False)
    for _ in range(5):
        p.run()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    print("Q%02d fused=%d  %.2f ms of kernels / statements, %.2f ms wall (results to the host)  matches the oracle: %s\n    %s\n    %s" % (n, p.is_fused, total, wall, ok, t, p.jit_note()[:700]), flush=True)
    e.close()
