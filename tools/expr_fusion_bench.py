import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
n = 59986052
def run(fuse, depth):
    if fuse: os.environ.pop("VDL_NO_EXPR_FUSION", None)
    else: os.environ["VDL_NO_EXPR_FUSION"] = "1"
    e = m.Engine(0)
    for c in datagen.Q6_COLUMNS: e.generate(datagen.LINEITEM[c], 0, n)
    lines = ["1,Load,lineitem.l_quantity", "2,Project,val,Id 1,l_quantity", "3,Load,lineitem.l_discount", "4,Project,val,Id 3,l_discount",
             "5,Load,lineitem.l_extendedprice", "6,Project,val,Id 5,l_extendedprice"]
    k, cur = 7, 2
    ops = ["Add", "Multiply", "Subtract", "BitwiseOr", "Add", "Greater", "Add", "BitwiseAnd"]
    for d in range(depth):
        other = [4, 6, 2][d % 3]
        lines.append("%d,%s,val,Id %d,val,Id %d,val" % (k, ops[d % len(ops)], cur, other)); cur = k; k += 1
    lines += ["%d,RangeV,val,0,Id %d,0" % (k, cur), "%d,FoldSum,val,Id %d,val,Id %d,val" % (k + 1, k, cur), "%d,MaterializeCompact,Id %d" % (k + 2, k + 1)]
    p = e.parse("\n".join(lines) + "\n"); p.set_fusion(False)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); p.execute(); best = min(best, time.perf_counter() - t0)
    r = p.collect()["results"]
    e.close()
    return best * 1e3, list(r.values())[0]
for depth in (1, 2, 4, 8):
    a, ra = run(True, depth); b, rb = run(False, depth)
    print("chain of %d operators over 60M rows (+ global FoldSum): fused %.2f ms, one kernel per operator %.2f ms, same result %s" % (depth, a, b, ra == rb))
