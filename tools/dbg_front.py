"""Debug: a compiled TPC-H plan under compiler flags, as planned (fused front on) with tracing: first statement that differs from the oracle."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from mplan2vdl_amd import catalog, frontend
from helpers import engine_with, compare_traced
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
n = int(sys.argv[1]); scale = float(sys.argv[2]); hier = len(sys.argv) > 3
cfg = frontend.load_metadata(META, **({"aggregation_strategy": ("AggHierarchical", 5)} if hier else {}))
text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
cols = catalog.synth_columns(META, cfg, text, scale=scale, seed=1)
orc = oracle.Oracle(); orc.keep_vectors(True)
for k, v in cols.items(): orc.add_column(k, v)
want = orc.run(text)
e = engine_with(cols)
p = e.parse(text)
p.set_trace(True)
os.environ["VDL_TRACE_FORMS"] = "1"
got = p.run()["results"]
print("equal" if got == want["results"] else "DIFFER", file=sys.stderr)
print(json.dumps(compare_traced(p, orc, text)), file=sys.stderr)
