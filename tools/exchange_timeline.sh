#!/bin/bash
# One rank of 8 of the sharded Q3 at SF100 (tools/exchange_phases.py) under the kernel trace: the kernels of its last exchange_begin / pack / finish
# with the gaps between them (host round trips show up as gaps).   tools/exchange_timeline.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-extl}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q3_DEVICE_OUTPUTS=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/exchange_phases.py > $OUT/run.log 2>&1
grep "^begin" $OUT/run.log | tail -2
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "k_gen_column" not in r["Kernel_Name"]]
# the fourth timed iteration: kernels between the third and the fourth-from-last project_front launches ... simpler: the 4th front launch onwards up to the 5th
fronts = [i for i, r in enumerate(rows) if "project_front" in r["Kernel_Name"]]
lo = fronts[3] - 3 if len(fronts) > 4 else 0
hi = fronts[4] - 3 if len(fronts) > 4 else len(rows)
q = rows[lo:hi]
t0 = int(q[0]["Start_Timestamp"]); prev = t0; busy = 0
for r in q:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%7.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    busy += e - s; prev = max(prev, e)
print("%.1f us from first kernel start to last kernel end, %.1f us of kernels, %d launches" % ((prev - t0) / 1e3, busy / 1e3, len(q)))
PY
