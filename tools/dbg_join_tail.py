import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
import mplan2vdl_amd as m
from helpers import engine_with, oracle_run, prog
text = prog("1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.fk", "4,Project,val,Id 3,fk", "5,Load,u.x", "6,Project,val,Id 5,x",
            "7,Gather,Id 6,Id 4,val", "8,Multiply,val,Id 2,val,Id 7,val", "9,RangeV,val,0,Id 8,0", "10,FoldSum,val,Id 9,val,Id 8,val",
            "11,MaterializeCompact,Id 10", "12,FoldCount,val,Id 9,val,Id 8,val", "13,MaterializeCompact,Id 12")
for n in (5, 300, 2047, 2048, 2049, 4096, 5000, 29993):
    rng = np.random.default_rng(n)
    cols = {"t.a": rng.integers(1, 10, n).astype(np.int64), "t.fk": rng.integers(0, 50, n).astype(np.int64), "u.x": rng.integers(1, 5, 50).astype(np.int64)}
    e = engine_with(cols); p = e.parse(text)
    got = p.run()["results"]; want = oracle_run(text, cols)
    print(n, "fused", p.is_fused, "OK" if got == want else "DIFF", got, want)
    if n == 5: print(p.describe())
    e.close()
