#!/usr/bin/env python3
"""TPC-H Q3 (machine-generated VDL, tests/golden/q3.vdl) on the GPU at a chosen scale; columns are built
on the device with torch (join indices are arithmetic), verified against numpy at small scale."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen

n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 15000000
e = m.Engine(0)
n_cust, n_li = max(n_orders // 10, 1), 4 * n_orders
keep = {}
def reg(name, t):
    keep[name] = t; e.register_tensor(name, t)
e.generate(datagen.CUSTOMER["customer.c_mktsegment"], 0, n_cust)
for name in datagen.ORDERS: e.generate(datagen.ORDERS[name], 0, n_orders)
e.generate(datagen.ColumnSpec("orders.orders_customer", np.int64, 0, n_cust - 1, 1, 0), 0, n_orders)
for name in ("lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"): e.generate(datagen.LINEITEM[name], 0, n_li)
reg("customer.customer_c_custkey_pkey", torch.zeros(n_cust, dtype=torch.int64, device="cuda"))
reg("orders.orders_o_orderkey_pkey", torch.zeros(n_orders, dtype=torch.int64, device="cuda"))
reg("lineitem.lineitem_l_orderkey_l_linenumber_pkey", torch.zeros(n_li, dtype=torch.int64, device="cuda"))
lo = torch.arange(n_li, dtype=torch.int64, device="cuda") // 4
reg("lineitem.lineitem_orders", lo)
reg("lineitem.l_orderkey", (1 + (lo // 8) * 32 + (lo % 8)).to(torch.int32))
torch.cuda.synchronize()
text = open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read()
plan = e.parse(text)
print("fused:", plan.is_fused)
for it in range(3):
    t0 = time.perf_counter(); out = plan.run(); dt = time.perf_counter() - t0
    rows = len(out["results"]["tmp110"][".revenue"])
    print("run %d: %.1f ms, %d result rows, %.2f M lineitem rows/s" % (it, dt * 1e3, rows, n_li / dt / 1e6))
if n_orders <= 2000000:
    from helpers import sql_q3
    t = {k: e.download(k) for k in datagen.Q3_COLUMNS}
    flat = {list(v.keys())[0][1:]: list(v.values())[0] for v in out["results"].values()}
    print("matches numpy SQL evaluation:", flat == sql_q3(t))
plan.set_profiling(True)
out = plan.run()
top = sorted(out["timings"].items(), key=lambda kv: -kv[1])[:12]
print("total profiled us", sum(out["timings"].values()))
for k, v in top: print("  ", k, v)
