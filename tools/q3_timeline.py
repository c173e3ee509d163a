"""Q3 at SF10, results left in HBM: the timeline of ONE query from a rocprofv3 kernel trace -- every kernel with its start
relative to the query's first kernel, its duration and the idle gap in front of it (host round trips and launch gaps show
up as gaps).      rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/q3_timeline.py run
                  python3 tools/q3_timeline.py show DIR/*/*kernel_trace.csv"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch
    import mplan2vdl_amd as m
    from mplan2vdl_amd import datagen
    e = m.Engine(0)
    n_orders = int(os.environ.get("Q3_ORDERS", "15000000"))
    keep = datagen.register_q3_columns(e, n_orders)
    sys.path.insert(0, ROOT)
    import bench
    p = e.parse(bench.q3_program(n_orders))
    if os.environ.get("Q3_JIT", "1") == "1":
        p.set_jit(True)
    p.set_device_outputs(True)
    for _ in range(6):
        p.execute()
        torch.cuda.synchronize()
    e.close()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # the last query = the kernels after the last gap of more than 200 us ... simpler: find the last launch of the first kernel name of a query
    names = [r["Kernel_Name"] for r in rows]
    first = next(i for i in range(len(rows) - 1, -1, -1) if ("project_select" in names[i] or "project_front" in names[i]) and (i == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 100000))
    q = rows[first:]
    t0 = int(q[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in q:
        s, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.1f us  +%7.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e_ - s) / 1e3, r["Kernel_Name"][:90]))
        busy += e_ - s
        prev_end = max(prev_end, e_)
    print("query: %.1f us from first kernel start to last kernel end, %.1f us of kernels, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(q)))
