#!/usr/bin/env python3
"""Marginal cost of the parts of the grouped fused scan: Q1 with subsets of its outputs (the planner drops
the aggregates nobody reads) at SF10; prints the fused-scan kernel time for each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen

rows = int(sys.argv[1]) if len(sys.argv) > 1 else datagen.LINEITEM_ROWS["sf10"]
text = open(os.path.join(ROOT, "tests", "golden", "q1.vdl")).read()
lines = text.splitlines()
outs = [(l.split(",")[0], lines[i - 1].split(",")[2]) for i, l in enumerate(lines) if ",MaterializeCompact," in l]
e = m.Engine(0)
for name in datagen.Q1_COLUMNS:
    e.generate(datagen.LINEITEM[name], 0, rows)
def run(keep, label):
    import re
    by_id = {l.split(",")[0]: l for l in lines}
    live, stack = set(), list(keep)
    while stack:                                     # dead-code elimination from the kept outputs
        i = stack.pop()
        if i in live:
            continue
        live.add(i)
        stack += re.findall(r"Id (\d+)", by_id[i])
    t = "\n".join(l for l in lines if l.split(",")[0] in live) + "\n"
    p = e.parse(t)
    assert p.is_fused, p.describe()
    p.set_profiling(True)
    us = []
    for _ in range(5):
        us.append(list(p.run()["timings"].values())[0])
    nagg = p.describe().count(" sum ") + p.describe().count(" first ")
    print("%-46s aggregates %d  kernel %7.1f us  %5.2f TB/s of the 44 B/row" % (label, nagg, min(us), rows * 44 / min(us) / 1e6))
ids = {name: i for i, name in outs}
print(outs)
run(set(ids.values()), "all outputs")
run({ids["count_order"]}, "count(*) only")
run({ids["sum_qty"]}, "sum(qty)")
run({ids["sum_qty"], ids["sum_base_price"]}, "sum(qty), sum(ep)")
run({ids["sum_disc_price"]}, "sum(ep*(100-disc))")
run({ids["sum_charge"]}, "sum(ep*(100-disc)*(100+tax))")
run({ids["l_returnflag__lineitem__l_returnflag"]}, "first(returnflag)")
run({ids["sum_qty"], ids["sum_base_price"], ids["sum_disc_price"], ids["sum_charge"]}, "4 sums")
e.close()
