import os, sys
sys.path.insert(0, "/root/repo")
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
e = m.Engine(0)
keep = datagen.register_q3_columns(e, 15000000)
p = e.parse(open("/root/repo/tests/golden/q3.vdl").read())
for _ in range(2): p.execute()
p.set_profiling(True)
p.execute()
t = p.collect(as_numpy=True)["timings"]
tot = sum(t.values())
print("sum of statement times %.2f ms" % (tot / 1e3))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:28]:
    print("  %-60s %7.0f us" % (k.replace("timeInMicrosecondsForStatement", ""), v))
