#!/bin/bash
# Q3 at SF10 and SF100 on one GPU (bench.py --query q3) + the kernel timeline of one SF10 query and the SF100 kernel table.   tools/q3_front_check.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-q3}; mkdir -p $OUT
cd $ROOT
python3 bench.py --query q3 --sf sf10 --steps 20 --warmup 3 > $OUT/q3_sf10.json 2> $OUT/q3_sf10.err || tail -n 5 $OUT/q3_sf10.err
python3 bench.py --query q3 --sf sf100 --steps 10 --warmup 3 > $OUT/q3_sf100.json 2> $OUT/q3_sf100.err || tail -n 5 $OUT/q3_sf100.err
python3 - $OUT <<'PY'
import json, sys
for sf in ("sf10", "sf100"):
    try:
        j = json.loads(open(sys.argv[1] + "/q3_%s.json" % sf).read().strip().splitlines()[-1])
        print(sf, "ms/query (results in HBM) %.3f" % j["ms_per_step"], "verified", j.get("verified_vs_torch_sql_checksums"), {k: v for k, v in j.items() if "host" in k})
    except Exception as e:
        print(sf, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t10 -- python3 $ROOT/tools/q3_timeline.py run > $OUT/t10.log 2>&1
python3 $ROOT/tools/q3_timeline.py show $OUT/t10/*/*kernel_trace.csv > $OUT/q3_sf10_timeline.txt 2>&1; cat $OUT/q3_sf10_timeline.txt | cut -c1-150
Q3_ORDERS=150000000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t100 -- python3 $ROOT/tools/q3_timeline.py run > $OUT/t100.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t100/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-90s %5s x %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
