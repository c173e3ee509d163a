#!/bin/bash
# kernel-time comparison of the two single-GPU routes for Q3 (statement by statement vs local phase + compaction + tail)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-15000000}
OUT=$ROOT/gpurun_out/prof_q3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for path in general exchange; do
  Q3_ONLY=$path rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$path -- python3 $ROOT/tools/run_q3.py $N > $OUT/$path.log 2>&1
  f=$(ls $OUT/$path/*/*kernel_stats.csv | head -n 1)
  python3 - "$f" "$path" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
runs = 8 if sys.argv[2] == "general" else 3        # tools/run_q3.py: 3 timed runs (+ 5 vdl_run-only runs on the general route)
print("%s: %.2f ms of kernels per run (%d runs)" % (sys.argv[2], tot / runs / 1e6, runs))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("   %-70s calls %5s  %8.3f ms/run" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / runs / 1e6))
PY
done
