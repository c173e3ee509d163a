#!/bin/bash
# tools/op_roofline.py twice: plain (per-statement HIP-event times), then under rocprofv3 --kernel-trace --stats (kernel table).
#   tools/prof_op_roofline.sh <tag>      -> gpurun_out/<tag>/op_roofline.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-op}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/op_roofline.py > $OUT/op_roofline.txt 2>&1 || { tail -n 20 $OUT/op_roofline.txt; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/op_roofline.py > $OUT/run.log 2>&1 || { tail -n 20 $OUT/run.log; exit 1; }
python3 - $OUT >> $OUT/op_roofline.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/*/*kernel_stats.csv"):
    print("\n%-100s %6s %10s %9s" % ("kernel", "calls", "total us", "avg us"))
    for r in list(csv.DictReader(open(f)))[:32]:
        print("%-100s %6s %10.1f %9.1f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
cat $OUT/op_roofline.txt
