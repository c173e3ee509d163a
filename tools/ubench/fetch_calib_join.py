#!/usr/bin/env python3
"""Joins the table fetch_calib prints (a plain run: times; or the run under the profiler) with the counter rows of
    rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -d DIR -- tools/ubench/fetch_calib
by dispatch order (each configuration is launched `reps` times; the counter of a configuration = mean over its launches but the first)
and prints, per configuration, the counter against the bytes of the 16-B lanes, 32-B, 64-B and 128-B units the kernel asked for.
    python tools/ubench/fetch_calib_join.py table.txt DIR/.../*counter_collection.csv [reps]"""
import csv, sys
table, pmc = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = [l.split() for l in open(table) if l.strip() and l.split()[0].isdigit() and len(l.split()) == 12]
disp = {}
for r in csv.DictReader(open(pmc)):
    if "k_calib" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
assert len(ids) == reps * len(rows), (len(ids), reps, len(rows))
names = sorted({k for d in disp.values() for k in d})
print("config            " + "".join("%16s" % n for n in names) + "   counter[0] x 1024 / bytes of:  lanes    32-B    64-B   128-B")
for i, r in enumerate(rows):
    mine = [disp[j] for j in ids[i * reps + (1 if reps > 1 else 0):(i + 1) * reps]]
    mean = {n: sum(d.get(n, 0.0) for d in mine) / len(mine) for n in names}
    lanes, u32, u64, u128 = (float(x) for x in r[4:8])
    c0 = mean[names[0]] * 1024
    print("%-9s k=%s p=%-7s" % (r[1], r[2], r[3]) + "".join("%16.0f" % mean[n] for n in names) +
          "                             %7.3f %7.3f %7.3f %7.3f" % (c0 / lanes, c0 / u32, c0 / u64, c0 / u128))
