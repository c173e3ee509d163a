#!/usr/bin/env python3
"""TPC-H Q14 (fused join scan: lineitem x part through the join index, p_type LIKE 'PROMO%') at a chosen size: kernel time, bytes
moved (counted), per-phase times.      python3 tools/q14_probe.py [n_lineitem] [runs]
    rocprofv3 --kernel-trace --stats -- python3 tools/q14_probe.py 600037902
    rocprofv3 --pmc FETCH_SIZE ... --kernel-trace -- python3 tools/q14_probe.py 59986052
part has n_lineitem / 30 rows (TPC-H: 200 000 x SF parts for 6 000 000 x SF lineitems)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen

n_li = int(sys.argv[1]) if len(sys.argv) > 1 else datagen.LINEITEM_ROWS["sf10"]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
e = m.Engine(0)
keep = datagen.register_q14_columns(e, n_li)
p = e.parse(open(os.path.join(ROOT, "tests", "golden", "q14.vdl")).read())
assert p.is_fused, p.describe()
p.set_profiling(True)
mode = os.environ.get("Q14_JIT", "tune")
if mode != "off":
    p.set_jit(True, tune=mode == "tune")
wall, kern = [], []
for _ in range(runs):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = p.run(); wall.append(time.perf_counter() - t0)
    kern.append(p.scan_stats()[2])
moved, detail = p.scan_traffic()
k_us = sum(kern[2:]) / len(kern[2:])
label = next((k for k in r["timings"] if "FusedScan" in k), "?")
print("Q14 %d lineitems, %d parts: query %.1f us, scan kernel %.1f us (%s)" % (n_li, max(n_li // 30, 1), 1e6 * sum(wall[2:]) / len(wall[2:]), k_us, label.replace("timeInMicrosecondsForFusedScan_", "")))
print("  bytes moved per launch (lineitem columns, counted): %d = %.2f B/row -> %.2f TB/s = %.3f of 8 TB/s; algorithmic 28 B/row -> %.2f TB/s" %
      (moved, moved / n_li, moved / k_us / 1e6, moved / k_us / 1e6 / 8, n_li * 28 / k_us / 1e6))
print("  " + detail)
print("  timings:", {k.replace("timeInMicrosecondsFor", ""): v for k, v in r["timings"].items()})
print("  note:", p.jit_note()[:600])
e.close()
