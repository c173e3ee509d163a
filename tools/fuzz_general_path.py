"""Longer run of tests/test_random_programs.py: 600 random programs x 3 executor modes against the oracle (GPU box)."""
import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_random_programs as t
bad = 0
for mode in (None, "VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE"):
    for k in ("VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE"): os.environ.pop(k, None)
    if mode: os.environ[mode] = "1"
    for seed in range(1000, 1600):
        try:
            t.check(seed, 5 + seed % 60)
        except AssertionError as e:
            bad += 1
            print("FAIL", mode, seed); print(str(e)[:1500])
            if bad > 3: sys.exit(1)
        except Exception as e:
            bad += 1
            print("ERROR", mode, seed, type(e).__name__, str(e)[:300])
            if bad > 3: sys.exit(1)
print("done, failures:", bad)
