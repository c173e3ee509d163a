"""Q3 at SF10 (60 M lineitems) on one GPU: wall time per query with the results left in HBM and copied to the host, and the
per-statement times of a profiled run (each statement is followed by a synchronise there: they include launch gaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
e = m.Engine(0)
keep = datagen.register_q3_columns(e, 15000000)
p = e.parse(open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read())
for dev in (True, False):
    p.set_device_outputs(dev)
    for _ in range(3): p.execute()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): p.execute()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("Q3 SF10 wall per query, results %s: %.3f ms (%.1f G lineitem rows/s)" % ("left in HBM" if dev else "copied to the host", dt * 1e3, 60e6 / dt / 1e9))
p.set_profiling(True)
p.execute()
t = p.collect(as_numpy=True)["timings"]
print("sum of statement times %.2f ms" % (sum(t.values()) / 1e3))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-100s %7.0f us" % (k.replace("timeInMicrosecondsForStatement", "")[:100], v))
