"""TPC-H Q6 and Q1 at SF10 statement by statement (fusion switched off) next to the planner's fused scans."""
import sys, time
sys.path.insert(0, "/root/repo")
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
n = 59986052
e = m.Engine(0)
for name in datagen.Q1_COLUMNS:
    e.generate(datagen.LINEITEM[name], 0, n)
for q in ("q6", "q1"):
    p = e.parse(open("/root/repo/tests/golden/%s.vdl" % q).read())
    res = {}
    for fused in (True, False):
        p.set_fusion(fused)
        for _ in range(3): p.execute()
        t0 = time.perf_counter()
        for _ in range(10): p.execute()
        dt = (time.perf_counter() - t0) / 10
        res[fused] = (dt, p.collect()["results"])
    print("%s SF10: fused %.2f ms, statement by statement %.2f ms, same results %s" % (q, res[True][0] * 1e3, res[False][0] * 1e3, res[True][1] == res[False][1]), flush=True)
e.close()
