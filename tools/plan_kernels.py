"""One compiled TPC-H plan over the synthetic catalog under `rocprofv3 --kernel-trace --stats`: which kernels its queries spend
their time in.    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/plan_kernels.py 18 0.3 [clustered]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
n, scale = int(sys.argv[1]), float(sys.argv[2])
cfg = frontend.load_metadata(META)
text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
cols = catalog.synth_columns(META, cfg, text, scale=scale, clustered=("lineitem.lineitem_orders",) if len(sys.argv) > 3 else ())
e = m.Engine(0)
for k, v in cols.items():
    e.upload(k, v)
plan = e.parse(text)
if os.environ.get("PLAN_JIT") == "1":
    plan.set_jit(True)
for _ in range(8):
    plan.run()
e.close()
