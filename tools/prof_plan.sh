#!/bin/bash
# kernel-level profile of one TPC-H plan through tools/run_plans.py:  tools/prof_plan.sh <scale> <plan>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_plan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN_PLANS_ORACLE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/run_plans.py $1 $2 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/t/*/*kernel_stats.csv"))[-1]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-72s calls %4s avg %8.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
