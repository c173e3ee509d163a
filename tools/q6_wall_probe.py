"""Wall time of one fused Q6 run (18 M rows) with profiling on / off and specialisation on / off: run() = launch + finalise + results to the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
n = 18000000
e = m.Engine(0)
for c in datagen.Q6_COLUMNS:
    e.generate(datagen.LINEITEM[c], 0, n)
text = open(os.path.join(ROOT, "tests", "golden", "q6.vdl")).read()
for jit in (False, True):
    for prof in (True, False):
        p = e.parse(text)
        p.set_profiling(prof)
        if jit:
            p.set_jit(True, tune=True)
        for _ in range(4):
            p.run()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); p.run(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("jit %-5s profiling %-5s: median %.3f ms, min %.3f, max %.3f" % (jit, prof, ts[10] * 1e3, ts[0] * 1e3, ts[-1] * 1e3), flush=True)
e.close()
# toggling profiling on a tuned plan: every run's wall time
e = m.Engine(0)
for c in datagen.Q6_COLUMNS:
    e.generate(datagen.LINEITEM[c], 0, n)
p = e.parse(text); p.set_profiling(True); p.set_jit(True, tune=True)
for _ in range(6): p.run()
p.set_profiling(False)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); p.run(); ts.append((time.perf_counter() - t0) * 1e3)
print("after set_profiling(False) on a tuned plan:", ["%.3f" % t for t in ts], p.jit_note()[-120:])
e.close()
