#!/usr/bin/env python3
"""Times the multi-aggregate global scan (k_scan<8,8,...>) on an ungrouped Q1-style program."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
from helpers import prog
text = prog(
    "1,Load,lineitem.l_quantity", "2,Project,val,Id 1,l_quantity",
    "3,Load,lineitem.l_shipdate", "4,Project,val,Id 3,l_shipdate",
    "5,RangeV,val,729999,Id 2,0", "6,Greater,val,Id 5,val,Id 4,val", "7,Equals,val,Id 4,val,Id 5,val",
    "8,LogicalOr,val,Id 6,val,Id 7,val", "9,RangeV,val,0,Id 8,1", "10,FoldSelect,val,Id 9,val,Id 8,val",
    "11,Gather,Id 2,Id 10,val", "12,RangeV,val,0,Id 11,0",
    "13,FoldSum,val,Id 12,val,Id 11,val", "14,Project,sum_qty,Id 13,val", "15,MaterializeCompact,Id 14",
    "16,Load,lineitem.l_extendedprice", "17,Project,val,Id 16,l_extendedprice", "18,Gather,Id 17,Id 10,val",
    "19,Load,lineitem.l_discount", "20,Project,val,Id 19,l_discount", "21,Gather,Id 20,Id 10,val",
    "22,RangeV,val,100,Id 11,0", "23,Subtract,val,Id 22,val,Id 21,val", "24,Multiply,val,Id 18,val,Id 23,val",
    "25,FoldSum,val,Id 12,val,Id 24,val", "26,Project,sum_disc_price,Id 25,val", "27,MaterializeCompact,Id 26",
    "28,Load,lineitem.l_tax", "29,Project,val,Id 28,l_tax", "30,Gather,Id 29,Id 10,val",
    "31,Add,val,Id 22,val,Id 30,val", "32,Multiply,val,Id 24,val,Id 31,val",
    "33,FoldSum,val,Id 12,val,Id 32,val", "34,Project,sum_charge,Id 33,val", "35,MaterializeCompact,Id 34",
    "36,RangeV,val,1,Id 11,0", "37,FoldSum,val,Id 12,val,Id 36,val",
    "41,Project,count_order,Id 37,val", "42,MaterializeCompact,Id 41",
    "43,FoldSum,val,Id 12,val,Id 18,val", "44,Project,sum_base,Id 43,val", "45,MaterializeCompact,Id 44",
    "46,FoldSum,val,Id 12,val,Id 21,val", "47,Project,sum_disc,Id 46,val", "48,MaterializeCompact,Id 47")
n = datagen.LINEITEM_ROWS[sys.argv[1] if len(sys.argv) > 1 else "sf10"]
e = m.Engine(0)
for c in ["lineitem.l_shipdate", "lineitem.l_quantity", "lineitem.l_extendedprice", "lineitem.l_discount", "lineitem.l_tax"]:
    e.generate(datagen.LINEITEM[c], 0, n)
p = e.parse(text); p.set_profiling(True)
assert p.is_fused
ts = []
for i in range(12):
    out = p.run()
    if i >= 2: ts.append(p.scan_stats()[2])
med = statistics.median(ts)
print(list(out["timings"])[0], "median us", med, "GB/s", n * 36 / med / 1e3)
