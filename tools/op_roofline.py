#!/usr/bin/env python3
"""Per-operator kernels of the general path against the HBM roofline: one statement each over n rows (sparse vectors and
fusion off, so every statement really runs its own kernels), time from the per-statement HIP events, algorithmic bytes
as the statement's inputs + outputs (8 B values, 1/8 B validity)."""
import os, sys
os.environ["VDL_NO_SPARSE"] = "1"; os.environ["VDL_NO_EXPR_FUSION"] = "1"; os.environ["VDL_NO_FILTER_FUSION"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 59986052
e = m.Engine(0)
for c in datagen.Q1_COLUMNS:
    e.generate(datagen.LINEITEM[c], 0, n)
e.generate(datagen.ColumnSpec("lineitem.fk", np.int64, 0, n // 4 - 1, 1, 0), 0, n)
e.generate(datagen.ColumnSpec("dim.x", np.int64, 0, 1000, 1, 0), 0, n // 4)
lines = ["1,Load,lineitem.l_quantity", "2,Project,val,Id 1,l_quantity", "3,Load,lineitem.l_extendedprice", "4,Project,val,Id 3,l_extendedprice",
         "5,Load,lineitem.l_discount", "6,Project,val,Id 5,l_discount", "7,Load,lineitem.fk", "8,Project,val,Id 7,fk", "9,Load,dim.x", "10,Project,val,Id 9,x",
         "11,Multiply,val,Id 2,val,Id 4,val",                                   # binary: 16 B in, 8 B out
         "12,RangeV,val,5,Id 6,0", "13,Greater,val,Id 6,val,Id 12,val",         # compare with a constant: 8 in, 8 out
         "14,RangeV,val,0,Id 13,1", "15,FoldSelect,val,Id 14,val,Id 13,val",   # filter (45 % pass): 8 B in, 1 bit out
         "16,Gather,Id 10,Id 8,val",                                            # FK gather: 8 B idx + 8 B random + 8 B out
         "17,Gather,Id 4,Id 15,val",                                            # identity gather: a view
         "18,RangeV,val,0,Id 17,0", "19,FoldSum,val,Id 18,val,Id 17,val", "20,MaterializeCompact,Id 19",   # global fold of a filtered vector
         "21,RangeV,val,31,Id 6,0", "22,BitwiseAnd,val,Id 4,val,Id 21,val", "23,RangeC,val,0,32,1",
         "24,Partition,val,Id 22,val,Id 23,val",                                # dense-domain partition (1 pass)
         "25,RangeV,val,0,Id 22,1", "26,Scatter,Id 22,Id 25,val,Id 24,val", "27,Scatter,Id 2,Id 25,val,Id 24,val",   # scatter by a permutation
         "28,FoldSum,val,Id 26,val,Id 27,val", "29,MaterializeCompact,Id 28",   # fold over runs (32 runs)
         "30,RangeC,val,0,1099511627776,1", "31,Multiply,val,Id 4,val,Id 2,val", "32,Partition,val,Id 31,val,Id 30,val",   # 2^40 domain: 5 passes
         # keep every measured statement alive through a cheap global fold
         "33,RangeV,val,0,Id 11,0", "34,FoldMax,val,Id 33,val,Id 11,val", "35,MaterializeCompact,Id 34",
         "36,RangeV,val,0,Id 16,0", "37,FoldMax,val,Id 36,val,Id 16,val", "38,MaterializeCompact,Id 37",
         # the sparse-domain Partition's readers are Scatters, as in every compiled GROUP BY (its positions then stay in rank order)
         "39,RangeV,val,0,Id 31,1", "40,Scatter,Id 31,Id 39,val,Id 32,val", "41,Scatter,Id 2,Id 39,val,Id 32,val",
         "42,RangeV,val,0,Id 40,0", "43,FoldMax,val,Id 42,val,Id 40,val", "44,MaterializeCompact,Id 43",
         "45,RangeV,val,0,Id 41,0", "46,FoldMax,val,Id 45,val,Id 41,val", "47,MaterializeCompact,Id 46"]
p = e.parse("\n".join(lines) + "\n")
p.set_fusion(False)
p.execute(); p.set_profiling(True); p.execute()
t = p.collect()["timings"]
algo = {"11_Multiply": 24, "13_Greater": 16, "15_FoldSelect": 8.125, "16_Gather": 24, "24_Partition": 16, "26_Scatter": 24.25, "27_Scatter": 24.25,
        "28_FoldSum": 16.25, "19_FoldSum": 8.25, "34_FoldMax": 8, "32_Partition": 16, "22_BitwiseAnd": 16, "31_Multiply": 24,
        "40_Scatter": 24, "41_Scatter": 24}
print("%d rows" % n, len(t), list(t)[:4])
for k, v in t.items():
    name = k.replace("timeInMicrosecondsForStatement", "")
    if name in algo:
        print("  %-14s %8.0f us   %6.2f TB/s of algorithmic bytes (%g B/row)" % (name, v, n * algo[name] / v / 1e6, algo[name]))
e.close()
