"""Q3 at SF10 through vdl_run_sharded with a ONE-rank RCCL communicator (the exchange route with itself as the only peer: local
phase, owner ranks, packing, tail on the packed rows) beside the plain vdl_run: what the route itself costs on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mplan2vdl_amd as m
from mplan2vdl_amd import datagen
import bench
n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 15000000
e = m.Engine(0)
keep = datagen.register_q3_columns(e, n_orders)
e.comm_init_rccl(0, 1, e.comm_unique_id())
for sharded in (False, True):
    p = e.parse(bench.q3_program(n_orders))
    p.set_jit(True)
    p.set_device_outputs(True)
    if sharded:
        p.set_sharded_table("lineitem")
    go = p.execute_sharded if sharded else p.execute
    for _ in range(3): go()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): go()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s: %.3f ms per query, checksums %s" % ("vdl_run_sharded (exchange route, 1 rank)" if sharded else "vdl_run", dt * 1e3, bench.q3_checksums_of_result(p.collect()["results"], "cuda:0")[:2]))
e.close()
