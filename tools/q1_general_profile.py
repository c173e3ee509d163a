import sys
sys.path.insert(0, "/root/repo")
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
n = 59986052
e = m.Engine(0)
for name in datagen.Q1_COLUMNS: e.generate(datagen.LINEITEM[name], 0, n)
p = e.parse(open("/root/repo/tests/golden/q1.vdl").read()); p.set_fusion(False)
for _ in range(2): p.execute()
p.set_profiling(True); p.execute()
t = p.collect(as_numpy=True)["timings"]
print("sum %.2f ms" % (sum(t.values()) / 1e3))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:14]: print("  %-40s %7.0f us" % (k.replace("timeInMicrosecondsForStatement", ""), v))
