"""Repeats tests/test_random_programs.py's programs (seeds 0..119, as the test runs them) to catch results that differ
from run to run: python tools/stress_random_programs.py [repeats]   (set VDL_SPARSE_ALWAYS=1 etc. outside)"""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_random_programs as t
from helpers import engine_with, oracle_run
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
progs = [(seed,) + t.Gen(seed).build(10 + seed % 30) for seed in range(120)]
wants = {seed: oracle_run(text, cols) for seed, text, cols in progs}
bad = 0
for rep in range(reps):
    for seed, text, cols in progs:
        e = engine_with(cols)
        try:
            got = e.run_vdl(text)["results"]
        except Exception as ex:
            bad += 1
            print("ERROR rep %d seed %d: %s" % (rep, seed, str(ex)[:300]))
            e.close()
            continue
        e.close()
        if got != wants[seed]:
            bad += 1
            print("MISMATCH rep %d seed %d" % (rep, seed))
            for k in wants[seed]:
                if got.get(k) != wants[seed][k]:
                    print("   ", k, "got", str(got.get(k))[:300], "want", str(wants[seed][k])[:300])
            if bad <= 2:
                print(text)
    print("rep", rep, "done, problems so far", bad, flush=True)
