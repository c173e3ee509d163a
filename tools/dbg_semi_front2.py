import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import engine_with, oracle_run
from test_random_semijoins import Gen
for seed in (0, 2, 6, 9, 1):
    text, cols = Gen(seed, short_fact=True, sparse_domain=True).build()
    want = oracle_run(text, cols)
    nt, nu = len(cols["t.a"]), len(cols["u.x"])
    for rep in range(3):
        e = engine_with(cols)
        p = e.parse(text)
        p.set_profiling(True)
        out = p.run()
        lens = [len(list(v.values())[0]) for v in out["results"].values()]
        print("seed", seed, "nt", nt, "nu", nu, "ok" if out["results"] == want else "DIFFERS", lens, [len(list(v.values())[0]) for v in want.values()],
              [k for k in out["timings"] if "Front" in k or "Abandon" in k], flush=True)
        e.close()
