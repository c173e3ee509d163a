"""Longer run of tests/test_random_programs.py: 600 random programs x 3 executor modes against the oracle (GPU box).
    python tools/fuzz_general_path.py [rows [programs]]      rows: tables of about that many rows (default: up to 4000)"""
import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1: os.environ["VDL_FUZZ_ROWS"] = sys.argv[1]
NPROG = int(sys.argv[2]) if len(sys.argv) > 2 else 600
import test_random_programs as t
bad = 0
for mode in (None, "VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE"):
    for k in ("VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE"): os.environ.pop(k, None)
    if mode: os.environ[mode] = "1"
    for seed in range(1000, 1000 + NPROG):
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
