"""Where the time of the sharded Partition route goes for ONE rank of 8 at SF100 (Q3): local phase, packing, tail.
Q3_DEVICE_OUTPUTS=1 leaves the result columns in HBM."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, mplan2vdl_amd as m
from mplan2vdl_amd import datagen, catalog, frontend
n_orders = 150000000
n_li = 4 * n_orders
r0, r1 = m.shard_rows(n_li, 3, 8)
e = m.Engine(0); e.use_torch_stream()
keep = datagen.register_q3_columns(e, n_orders, (r0, r1), device="cuda:0", copartition=True)
meta = "/root/repo/tests/golden/tpch10noorder"
text = frontend.compile_plan(open(meta + "/03.sql.mplan").read(), catalog.tpch_scaled_config(frontend.load_metadata(meta), 10))
plan = e.parse(text)
if os.environ.get("Q3_JIT", "1") == "1":
    plan.set_jit(True)          # (the front specialised, as bench.py runs it)
if os.environ.get("Q3_DEVICE_OUTPUTS"):
    plan.set_device_outputs(True)
ncols = plan.exchange_columns("lineitem")
def sync(): torch.cuda.synchronize()
for it in range(4):
    sync(); t0 = time.perf_counter()
    counts = plan.exchange_begin(1); sync(); t1 = time.perf_counter()
    n_send = sum(counts)
    send = torch.empty((ncols, max(n_send, 1)), dtype=torch.int64, device="cuda:0")[:, :n_send].contiguous(); sync(); t2 = time.perf_counter()
    plan.exchange_pack(send.data_ptr()); sync(); t3 = time.perf_counter()
    out = plan.exchange_finish(send.data_ptr(), n_send, True); sync(); t4 = time.perf_counter()
    print("begin %.2f ms  alloc %.2f  pack %.2f  finish %.2f  (rows sent %d, result rows %d)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, n_send, len(out["results"]["tmp110"][".revenue"])), flush=True)
plan.set_profiling(True)
counts = plan.exchange_begin(1)
t = plan.collect(as_numpy=True)["timings"]
print("begin: sum of statements %.2f ms" % (sum(t.values())/1e3))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:8]: print("   ", k.replace("timeInMicrosecondsForStatement",""), v)
send = torch.empty((ncols, sum(counts)), dtype=torch.int64, device="cuda:0")
plan.exchange_pack(send.data_ptr())
out = plan.exchange_finish(send.data_ptr(), sum(counts), True)
t = out["timings"]
print("finish: sum of statements %.2f ms" % (sum(t.values())/1e3))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:8]: print("   ", k.replace("timeInMicrosecondsForStatement",""), v)
