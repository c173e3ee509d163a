#!/bin/bash
# Device work per rank of TPC-H Q18 on the chain route (tools/q18_chain.py), from rocprofv3's kernel table: worlds 1 (unsharded), 2, 4 on one GPU.
#   tools/q18_chain_profile.sh [scale=0.3] [runs=5]        -> gpurun_out/q18_chain/
set -e
SCALE=${1:-0.3}; RUNS=${2:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/q18_chain
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in 1 2 4; do
  rocprofv3 --kernel-trace --stats -d $OUT/w$W -o q18 --output-format csv -- python3 $ROOT/tools/q18_chain.py $W $SCALE $RUNS > $OUT/w$W.log 2>&1
  grep "^Q18\|^unsharded\|^route" $OUT/w$W.log
done
python3 - $OUT $RUNS <<'PY'
import csv, glob, sys
out, runs = sys.argv[1], int(sys.argv[2])
def table(w):
    f = glob.glob("%s/w%d/**/*kernel_stats.csv" % (out, w), recursive=True)[0]
    # (the runtime's own copy / fill kernels are the HOST TRANSPORT's staging here -- device to host and back for every collective --, which an
    # RCCL job does not have: left out)
    return {r["Name"]: (float(r["TotalDurationNs"]), int(r["Calls"])) for r in csv.DictReader(open(f)) if not r["Name"].startswith("__amd_rocclr_")}
base = table(1)
u = sum(v[0] for v in base.values())
print("unsharded: %.3f ms of kernels per run (%d runs in the process)" % (u / (runs + 1) / 1e6, runs + 1))
for w in (2, 4):
    t = table(w)
    s = sum(v[0] for v in t.values()) - u                      # every process also does the unsharded leg
    per = s / ((runs + 1) * w)
    print("world %d: %.3f ms of kernels per rank and run (%.2f of the unsharded run's; ideal %.2f)" % (w, per / 1e6, per / (u / (runs + 1)), 1.0 / w))
    top = sorted(((v[0] - base.get(k, (0, 0))[0], v[1] - base.get(k, (0, 0))[1], k) for k, v in t.items()), reverse=True)[:10]
    for ns, calls, k in top:
        print("    %8.1f us per rank and run  %5.1f launches  %s" % (ns / ((runs + 1) * w) / 1e3, calls / ((runs + 1) * w), k[:110]))
PY
