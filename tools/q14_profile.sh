#!/bin/bash
# Q14's fused join scan at SF10 and at SF100-sized lineitem: times, kernel table, counters.   tools/q14_profile.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-q14}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 59986052 600037902; do
    python3 $ROOT/tools/q14_probe.py $n > $OUT/probe_$n.txt 2>&1; cat $OUT/probe_$n.txt | cut -c1-400
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$n -- python3 $ROOT/tools/q14_probe.py $n 8 > $OUT/t_$n.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pa_$n -- python3 $ROOT/tools/q14_probe.py $n 6 > $OUT/pa_$n.log 2>&1
    rocprofv3 --pmc VALUBusy MemUnitStalled --kernel-trace --output-format csv -d $OUT/pb_$n -- python3 $ROOT/tools/q14_probe.py $n 6 > $OUT/pb_$n.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pc_$n -- python3 $ROOT/tools/q14_probe.py $n 6 > $OUT/pc_$n.log 2>&1
    python3 - $OUT $n <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], sys.argv[2]
for f in glob.glob("%s/t_%s/*/*kernel_stats.csv" % (out, n)):
    for r in list(csv.DictReader(open(f)))[:8]:
        if "gen_column" in r["Name"] or "at::" in r["Name"]: continue
        print("    %-80s %4s x %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
for tag in ("pa", "pb", "pc"):
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, n), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "mscan" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("    counters of %s: %s" % (k, {c: "%.4g (last of %d)" % (x[-1], len(x)) for c, x in v.items()}))
PY
done
