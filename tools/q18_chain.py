#!/usr/bin/env python3
"""TPC-H Q18 through the "chain" route of vdl_run_sharded, rehearsed on ONE GPU: WORLD ranks (threads, one context each) over the host
transport, lineitem split by rows, the other tables replicated.  Prints the wall time per run (all ranks share the one GPU, so this is
NOT a scaling number) and checks every rank's answer against an unsharded run of the same plan; run under `rocprofv3 --kernel-trace
--stats` (tools/q18_chain_profile.sh) the kernel table divided by WORLD x runs is the device work ONE rank does per query -- what a
rank of a real N-GPU job would spend in kernels.
    python tools/q18_chain.py WORLD [scale=0.3] [runs=5]          Q18_CLUSTERED=0: lineitem in random order (default: clustered by order, as dbgen writes it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend, shard_rows
from helpers import run_ranks, engine_with

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
clustered = os.environ.get("Q18_CLUSTERED", "1") == "1"
cfg = frontend.load_metadata(META)
if scale > 1:                                                # (the program depends on the catalog's bounds: beyond SF10 it is compiled for the larger catalog)
    cfg = catalog.tpch_scaled_config(cfg, scale)
text = frontend.compile_plan(open(os.path.join(META, "18.sql.mplan")).read(), cfg)
cols = catalog.synth_columns(META, cfg, text, scale=min(scale, 1.0), clustered=("lineitem.lineitem_orders",) if clustered else ())
n = len(cols["lineitem.l_quantity"])
print("Q18, lineitem %d rows (%s), %d rank(s) on one GPU over the host transport, %d run(s) after one warm-up" % (n, "clustered by order" if clustered else "random order", world, runs), flush=True)

e = engine_with(cols)
p = e.parse(text)
want = p.run()["results"]
t0 = time.perf_counter()
for _ in range(runs):
    p.run()
print("unsharded: %.2f ms per run (wall), %d result rows" % ((time.perf_counter() - t0) / runs * 1e3, max(len(list(v.values())[0]) for v in want.values())), flush=True)
e.close()
if world == 1:
    sys.exit(0)


# one rank at a time on the GPU: a rank holds the token while it computes and hands it over while it waits in a collective, so that
# the kernel durations rocprofv3 records are those of a rank that has the GPU to itself (as on a real N-GPU job)
import threading
token = threading.Lock()


def handing_over(fn):
    def wrapped(x):
        token.release()
        try:
            return fn(x)
        finally:
            token.acquire()
    return wrapped


def work(rank, rv):
    r0, r1 = shard_rows(n, rank, world)
    c = {k: (v[r0:r1] if k.startswith("lineitem.") and not k.endswith(".heap") else v) for k, v in cols.items()}
    e = engine_with(c)
    e.comm_init_host(rank, world, *[handing_over(f) for f in rv.transport(rank)])
    p = e.parse(text)
    p.set_sharded_table("lineitem")
    p.set_row_offset(r0)
    route = p.sharded_route()
    with token:
        ok = p.run_sharded()["results"] == want
    rv.barrier.wait()
    t0 = time.perf_counter()
    for _ in range(runs):
        with token:
            p.run_sharded()
    dt = (time.perf_counter() - t0) / runs
    e.close()
    return route, ok, dt


got = run_ranks(world, work, timeout=600)
print("route %s; every rank's answer equals the unsharded one: %s; %.2f ms per run (wall, %d ranks sharing one GPU and one host transport)" % (
    got[0][0], all(g[1] for g in got), max(g[2] for g in got) * 1e3, world), flush=True)
assert all(g[1] for g in got)
