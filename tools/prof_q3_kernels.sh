#!/bin/bash
# per-kernel time of Q3 statement by statement at a chosen scale (default SF100), results left on the device
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-150000000}
OUT=$ROOT/gpurun_out/prof_q3_kernels
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export Q3_ONLY=general Q3_DEVICE_OUTPUTS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/run_q3.py $N > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    runs = 8
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("%.2f ms of kernels per run" % (tot / runs / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
        print("   %-72s calls/run %5.1f  %8.3f ms/run  avg %8.1f us" % (r["Name"][:72], int(r["Calls"]) / runs, float(r["TotalDurationNs"]) / runs / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep -v "^  Id" $OUT/run.log | tail -n 3
