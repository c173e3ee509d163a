#!/bin/bash
# The dimension scans of Q3 at SF100 (orders: 150 M rows, date filter + customer-bitmap lookup) at several grids: blocks per CU of the projection scans.
#   tools/dimscan_sweep.sh <tag> [values...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-dimscan}; mkdir -p $OUT
shift
cd /tmp && export TMPDIR=/tmp
for v in ${@:-default 2 4 12 16}; do
  if [ "$v" = default ]; then unset VDL_PROJ_BLOCKS_PER_CU; else export VDL_PROJ_BLOCKS_PER_CU=$v; fi
  Q3_ORDERS=150000000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$v -- python3 $ROOT/tools/q3_timeline.py run > $OUT/b$v.log 2>&1
  python3 - $OUT/b$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "project_select" in r["Kernel_Name"] or "project_front" in r["Kernel_Name"]]
sel = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "select" in r["Kernel_Name"])
fr = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "front" in r["Kernel_Name"])
big = sel[len(sel) // 2:]
print("blocks per CU %-8s orders scan median %.1f us (customer scan %.1f us), front median %.1f us" % (sys.argv[2], big[len(big) // 2], sel[len(sel) // 4], fr[len(fr) // 2]))
PY
done
