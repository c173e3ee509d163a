"""Debug: random semi-join programs whose set belongs to a fused FRONT (sparse Partition domain), traced against the oracle."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import engine_with, compare_traced, oracle_run
from test_random_semijoins import Gen
for short in (False, True):
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        text, cols = Gen(seed, short_fact=short, sparse_domain=True).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        got = p.run()["results"]
        ok = got == want
        os.environ["VDL_NO_PROJECTION"] = "1"
        p2 = e.parse(text); nofront = p2.run()["results"] == want
        del os.environ["VDL_NO_PROJECTION"]
        if not ok:
            bad += 1
            if bad <= 2:
                orc = oracle.Oracle(); orc.keep_vectors(True)
                for k, v in cols.items(): orc.add_column(k, v)
                orc.run(text)
                p.set_trace(True)
                p.run()
                print("seed", seed, "short", short, "nt", len(cols["t.a"]), "nu", len(cols["u.x"]), json.dumps(compare_traced(p, orc, text))[:1500])
                print(p.describe().split("fusion disabled")[0])
                g = [len(list(v.values())[0]) for v in got.values()], [len(list(v.values())[0]) for v in want.values()]
                print("lens", g, got if len(str(got)) < 400 else "")
        print("seed", seed, "short", short, "front ok" if ok else "FRONT DIFFERS", "| no front:", "ok" if nofront else "DIFFERS", flush=True)
        e.close()
