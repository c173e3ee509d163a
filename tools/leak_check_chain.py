"""Device and host memory over 300 runs of TPC-H Q18 on the chain route (2 ranks = threads over the host transport, one GPU): both must be flat
after the first runs.   python tools/leak_check_chain.py"""
import os, sys, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend, shard_rows
from helpers import run_ranks, engine_with

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
cfg = frontend.load_metadata(META)
text = frontend.compile_plan(open(os.path.join(META, "18.sql.mplan")).read(), cfg)
cols = catalog.synth_columns(META, cfg, text, scale=0.02, clustered=("lineitem.lineitem_orders",))
n = len(cols["lineitem.l_quantity"])
world = 2


def work(rank, rv):
    r0, r1 = shard_rows(n, rank, world)
    c = {k: (v[r0:r1] if k.startswith("lineitem.") and not k.endswith(".heap") else v) for k, v in cols.items()}
    e = engine_with(c)
    e.comm_init_host(rank, world, *rv.transport(rank))
    p = e.parse(text)
    p.set_sharded_table("lineitem")
    p.set_row_offset(r0)
    first = None
    for it in range(300):
        res = p.run_sharded()["results"]
        first = first or res
        assert res == first
        if rank == 0 and it in (5, 50, 150, 299):
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            print("after %3d runs: %.1f MB of device memory in use, %.1f MB resident on the host" % (it + 1, (total - free) / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3), flush=True)
    e.close()
    return True


print("Q18 chain route, %d lineitems, %d ranks" % (n, world), flush=True)
run_ranks(world, work, timeout=900)
free, total = torch.cuda.mem_get_info()
print("after close: %.1f MB in use" % ((total - free) / 1e6))
