#!/usr/bin/env python3
"""TPC-H Q3 (machine-generated VDL, tests/golden/q3.vdl) on the GPU at a chosen scale.

    python tools/run_q3.py [n_orders]                                      one GPU, statement by statement
    python -m torch.distributed.run --nproc-per-node N ... tools/run_q3.py [n_orders]
                                                                            lineitem sharded by rows, orders and
                                                                            customer replicated, rows exchanged by
                                                                            key range (mplan2vdl_amd.run_exchange)
Columns are built on the device (join indices are arithmetic); the result is verified against a numpy
evaluation of the SQL at small scale.  VDL_Q3_SHARE_DEVICE=1 + VDL_Q3_BACKEND=gloo rehearse N ranks on one GPU.
Sharded runs co-partition orders with lineitem (each rank holds the orders rows its lineitems reference, join index
rebased); Q3_REPLICATE_ORDERS=1 replicates the whole orders table instead.  customer is always replicated."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen

n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 15000000
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = 0 if os.environ.get("VDL_Q3_SHARE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group(os.environ.get("VDL_Q3_BACKEND", "nccl"), rank=rank, world_size=world)
dev = "cuda:%d" % local
e = m.Engine(local)
e.use_torch_stream()
n_cust, n_li = max(n_orders // 10, 1), 4 * n_orders
r0, r1 = m.shard_rows(n_li, rank, world)
if os.environ.get("Q3_FAKE_SHARD"):            # "k/N": time what ONE of N ranks would do locally (its lineitem shard, full dimension tables)
    k, N = [int(x) for x in os.environ["Q3_FAKE_SHARD"].split("/")]
    r0, r1 = m.shard_rows(n_li, k, N)
    n_li = r1 - r0
copart = (world > 1 or bool(os.environ.get("Q3_FAKE_SHARD"))) and os.environ.get("Q3_REPLICATE_ORDERS") != "1"
keep = datagen.register_q3_columns(e, n_orders, (r0, r1), device=dev, copartition=copart)
text = open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read()
if n_orders > 15000000:
    # the fixture was compiled against the SF10 catalog (order keys up to 6e7 -> a 2^38 group-key domain); larger
    # data needs the program for its own bounds (here: the SF10 metadata scaled like TPC-H scales, 2^42 at SF100)
    from mplan2vdl_amd import catalog, frontend
    meta = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
    factor = -(-n_orders // 15000000)
    text = frontend.compile_plan(open(os.path.join(meta, "03.sql.mplan")).read(), catalog.tpch_scaled_config(frontend.load_metadata(meta), factor))
plan = e.parse(text)
if os.environ.get("Q3_DEVICE_OUTPUTS"):        # the 4 result columns stay in HBM (vdl_plan_set_device_outputs)
    plan.set_device_outputs(True)
say = print if rank == 0 else (lambda *a, **k: None)
say("fused:", plan.is_fused, " exchange columns:", plan.exchange_columns("lineitem"), " ranks:", world)

def timed(label, fn):
    out = None
    for it in range(3):
        if world > 1: dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        if world > 1: dist.barrier()
        dt = time.perf_counter() - t0
        rows = len(out["results"]["tmp110"][".revenue"])
        say("%s run %d: %.1f ms, %d result rows on rank 0, %.2f M lineitem rows/s" % (label, it, dt * 1e3, rows, n_li / dt / 1e6))
    return out

only = os.environ.get("Q3_ONLY")            # "exchange" / "general": one path only (for rocprofv3 runs)
if only == "general":
    timed("statement-by-statement", plan.run)
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); dt = time.perf_counter() - t0
        say("vdl_run alone (outputs left in the plan) run %d: %.2f ms, %.2f M lineitem rows/s" % (it, dt * 1e3, n_li / dt / 1e6))
    if os.environ.get("Q3_STATEMENTS"):         # where the time goes, statement by statement (vdl_set_profiling)
        plan.set_profiling(True); plan.execute()
        t = plan.collect(as_numpy=True)["timings"]
        say("sum of statement times %.2f ms" % (sum(t.values()) / 1e3))
        for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:int(os.environ["Q3_STATEMENTS"])]:
            say("  %-60s %7.0f us" % (k.replace("timeInMicrosecondsForStatement", ""), v))
    e.close(); sys.exit(0)
out = timed("exchange", lambda: m.run_exchange(plan, dist if world > 1 else None, device=dev, sharded_table="lineitem", as_numpy=True))
flat = {list(v.keys())[0][1:]: list(v.values())[0].tolist() for v in out["results"].values()}
if world > 1:
    parts = [None] * world
    dist.all_gather_object(parts, flat)
    flat = {k: sum((p[k] for p in parts), []) for k in flat}
if only == "exchange":
    e.close(); sys.exit(0)
if world == 1:
    whole = timed("statement-by-statement", plan.run)
    same = {list(v.keys())[0][1:]: list(v.values())[0] for v in whole["results"].values()} == flat
    say("exchange path == unsharded path:", same)
    if not same: sys.exit(1)
if n_orders <= 2000000 and rank == 0:
    from helpers import sql_q3
    ok = flat == sql_q3(datagen.q3_tables(n_orders))
    print("matches numpy SQL evaluation:", ok)
    if not ok: sys.exit(1)
if world == 1:
    plan.set_profiling(True)
    out = plan.run()
    top = sorted(out["timings"].items(), key=lambda kv: -kv[1])[:12]
    print("total profiled us", sum(out["timings"].values()))
    for k, v in top: print("  ", k, v)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
e.close()
