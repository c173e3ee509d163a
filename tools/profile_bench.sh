#!/bin/bash
# rocprofv3 summaries of the headline bench (Q6 SF100) and of Q1 SF100, copied under profiles/ by hand afterwards
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profile_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q6 -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/q6.json 2> $OUT/q6.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q1 -- python3 $ROOT/bench.py --query q1 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/q1.json 2> $OUT/q1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/q6_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/q6_fetch.json 2> $OUT/q6_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/q6_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/q6_write.json 2> $OUT/q6_write.err
for q in q6 q1; do
  f=$(ls $OUT/$q/*/*kernel_stats.csv | head -n 1); echo "== $q"; head -n 6 $f | cut -c1-160; tail -n 1 $OUT/$q.json | cut -c1-300
done
python3 - $OUT <<'PY'
import csv, glob, sys
for tag in ("q6_fetch", "q6_write"):
    f = glob.glob(sys.argv[1] + "/%s/*/*counter_collection.csv" % tag)[0]
    tot, cnt = 0.0, set()
    for r in csv.DictReader(open(f)):
        if "k_scan<" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); cnt.add(r["Dispatch_Id"])
    print(tag, "k_scan dispatches", len(cnt), "counter per dispatch %.0f (KB units per the guide)" % (tot / max(len(cnt), 1)))
PY
