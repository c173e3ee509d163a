#!/bin/bash
# HIP API time of the statement-by-statement route for Q3 (where does the wall time of vdl_run go?)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-15000000}
OUT=$ROOT/gpurun_out/prof_q3_host
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q3_ONLY=general rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/run_q3.py $N > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for pat in ("*hip_api_stats.csv", "*kernel_stats.csv"):
    for f in glob.glob(sys.argv[1] + "/t/*/" + pat):
        rows = list(csv.DictReader(open(f)))
        print(pat)
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
            print("   %-50s calls %7s  total %9.2f ms" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
tail -n 3 $OUT/run.log
