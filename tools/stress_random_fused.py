"""Repeats tests/test_random_fused.py's programs many times (fused and statement by statement) to catch results that
differ from run to run: python tools/stress_random_fused.py [repeats]"""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_random_fused as t
from helpers import engine_with, oracle_run
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
progs = [(seed,) + t.Gen(seed).build() for seed in range(250)]
wants = {seed: oracle_run(text, cols) for seed, text, cols in progs}
bad = 0
for rep in range(reps):
    for seed, text, cols in progs:
        e = engine_with(cols)
        p = e.parse(text)
        got = p.run()["results"]
        p.set_fusion(False)
        unf = p.run()["results"]
        e.close()
        for tag, r in (("fused" if p.is_fused else "general(1st)", got), ("statement by statement", unf)):
            if r != wants[seed]:
                bad += 1
                print("MISMATCH rep %d seed %d %s" % (rep, seed, tag))
                for k in wants[seed]:
                    if r.get(k) != wants[seed][k]:
                        print("   ", k, "got", str(r.get(k))[:200], "want", str(wants[seed][k])[:200])
                if bad == 1:
                    print(text)
    print("rep", rep, "done, mismatches so far", bad, flush=True)
