#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_ops
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/op_roofline.py > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "fold" in r["Kernel_Name"] or "seg" in r["Kernel_Name"]:
            print("%-70s %8.1f us grid %s" % (r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
PY
