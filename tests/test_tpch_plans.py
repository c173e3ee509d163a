"""Every TPC-H plan the front end compiles (15 of the 22 under tests/golden/tpch10noorder; the other 7 stop at
the reference's own `error` calls, tests/test_frontend.py) runs end to end over a synthetic catalog that
respects the catalog metadata (mplan2vdl_amd/catalog.py): oracle on the CPU, engine on the GPU, bit-exact."""
import os

import numpy as np
import pytest

from mplan2vdl_amd import catalog, frontend
from conftest import ROOT
from helpers import engine_with, oracle_run

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
PLANS = [1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20]


@pytest.fixture(scope="module")
def cfg():
    return frontend.load_metadata(META)


FLAG_SETS = {"hierarchical": {"aggregation_strategy": ("AggHierarchical", 5)}, "shuffle": {"aggregation_strategy": ("AggShuffle",)},
             "crossproduct": {"cross_product": True}}
CROSS_PLANS = [3, 11, 12, 14, 15, 16, 19, 20]          # the others multiply into more than 2^24 slots even at 600 lineitems


def program_and_columns(cfg, n, scale, seed=1):
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
    return text, catalog.synth_columns(META, cfg, text, scale=scale, seed=seed)


def test_synthetic_catalog_respects_the_metadata(cfg):
    text, cols = program_and_columns(cfg, 16, 2e-4)
    info = dict(cfg.colinfo.to_list())
    for path, v in cols.items():
        parts = path.split(".")
        if parts[-1] == "heap" or (parts[0], parts[1]) not in info:
            continue
        ci = info[(parts[0], parts[1])]
        if ci.trailing_zeros >= 63 or path.endswith("_pkey") or ci.dtype[0][0] == "DString":
            continue
        if "%" + parts[1] in {k[1] for k in info if k[0] == parts[0]}:          # FK join index: rows of the scaled dim table
            continue
        assert v.min() >= ci.bounds[0] and v.max() <= ci.bounds[1], path
    assert cols["part.p_type.heap"].dtype.itemsize == 1
    offs = set(cols["part.p_type"].tolist())
    heap = cols["part.p_type.heap"]
    assert all(o % 8 == 0 and (o == 0 or heap[o - 1] == 0) for o in offs)       # aligned string starts


@pytest.mark.parametrize("n", PLANS)
def test_oracle_runs_every_compiled_plan(cfg, n):
    text, cols = program_and_columns(cfg, n, 1e-4)
    res = oracle_run(text, cols)
    assert len(res) >= 1
    sizes = {len(list(v.values())[0]) for v in res.values()}
    assert len(sizes) == 1                                                      # all output columns of a query align


def test_every_plan_selects_something(cfg):
    """No vacuous parity: over the coherent catalog every one of the 15 plans returns rows (Q18's IN-subquery and Q15's
    view join on VALUES, which the catalog keeps consistent with the join indices)."""
    for n in PLANS:
        text, cols = program_and_columns(cfg, n, 5e-4, seed=3)
        res = oracle_run(text, cols)
        assert all(len(list(v.values())[0]) > 0 for v in res.values()), n


@pytest.mark.gpu
@pytest.mark.parametrize("n", PLANS)
@pytest.mark.parametrize("scale,seed", [(2e-4, 1), (1e-3, 7)])
def test_engine_matches_oracle_on_every_compiled_plan(cfg, n, scale, seed):
    text, cols = program_and_columns(cfg, n, scale, seed)
    want = oracle_run(text, cols)
    e = engine_with(cols)
    got = e.run_vdl(text)["results"]
    e.close()
    assert got == want


def test_cross_product_lowering_agrees_with_join_index_lowering(cfg):
    """--crossproduct (Vlite.hs:671-680) and the FK join-index lowering are two programs for the same SQL;
    they must select the same rows -- also Q15, which joins its view on VALUES (the catalog keeps key values
    consistent with the join indices)."""
    xcfg = frontend.load_metadata(META, cross_product=True)
    for n in (3, 11, 12, 14, 15, 16, 19, 20):
        a_text, a_cols = program_and_columns(xcfg, n, 1e-5)
        b_text, b_cols = program_and_columns(cfg, n, 1e-5)
        assert "CrossProductOuter" in a_text and "CrossProduct" not in b_text
        a, b = oracle_run(a_text, a_cols), oracle_run(b_text, b_cols)
        assert [list(v.values())[0] for v in a.values()] == [list(v.values())[0] for v in b.values()], n


@pytest.mark.gpu
@pytest.mark.parametrize("flags", sorted(FLAG_SETS))
def test_engine_matches_oracle_under_compiler_flags(flags):
    """AggHierarchical (two-level folds, Vlite.hs:1181-1192), AggShuffle (Shuffle before every Partition) and
    --crossproduct (CrossProductOuter/Inner + filters instead of join-index gathers)."""
    fcfg = frontend.load_metadata(META, **FLAG_SETS[flags])
    for n in (CROSS_PLANS if flags == "crossproduct" else PLANS):
        text, cols = program_and_columns(fcfg, n, 1e-5 if flags == "crossproduct" else 2e-4)
        want = oracle_run(text, cols)
        e = engine_with(cols)
        got = e.run_vdl(text)["results"]
        e.close()
        assert got == want, (flags, n)


# ---- the VLite output format (--vliteformat, Vdl.hs:370-408; Semisort-based grouping, Vlite.hs:1061-1064) ----------
def vlite_program_and_columns(cfg, n, scale, seed=1):
    vcfg = catalog.scaled_config(frontend.load_metadata(META, format="vlite"), scale)    # table lengths are printed into the program
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), vcfg)
    return text, catalog.synth_columns(META, cfg, text, scale=scale, seed=seed)


def values_of(results):
    return [list(v.values())[0] for v in results.values()]


def test_vlite_dialect_shape(cfg):
    text, _ = vlite_program_and_columns(cfg, 1, 2e-4)
    lines = text.splitlines()
    assert lines[0] == "1,Load,lineitem.l_returnflag" and lines[1] == "2,Project,Id 1"           # no field names
    assert any(l.split(",")[1] == "Semisort" for l in lines) and not any("Partition" in l or "Scatter" in l for l in lines)
    assert "l_returnflag,Output,string_lineitem.l_returnflag,Id 33" in lines                    # named outputs do not print their id
    assert "count_order,Output,decimal_0,Id 69" in lines and "sum_charge,Output,decimal_6,Id 65" in lines
    q3, _ = vlite_program_and_columns(cfg, 3, 2e-4)
    assert "10,RangeC,0,3000,1" in q3.splitlines()                                               # reference vector = table length, Vlite.hs:740


@pytest.mark.parametrize("n", [1, 6])
def test_vlite_and_vdl_programs_agree_on_single_table_plans(cfg, n):
    """Two lowerings of the same SQL (Partition + Scatter vs Semisort + Gather) through the same oracle.  Plans over
    several tables are left out: in the VLite format every table's reference vector is a RangeC, and the reference's
    structural hash makes all RangeC equal (Vlite.hs:121 `show RangeC {} = "RangeC {...}"`, :155-157), so the tables'
    lengths get mixed up -- a compiler bug the restatement reproduces."""
    a_text, a_cols = program_and_columns(cfg, n, 1e-3)
    b_text, b_cols = vlite_program_and_columns(cfg, n, 1e-3)
    assert values_of(oracle_run(a_text, a_cols)) == values_of(oracle_run(b_text, b_cols))


def run_or_error(fn):
    try:
        return fn(), None
    except Exception as exc:                 # OracleError / VdlError: a shape error the program itself contains
        return None, exc


@pytest.mark.gpu
@pytest.mark.parametrize("n", PLANS)
def test_engine_matches_oracle_on_the_vlite_dialect(cfg, n):
    text, cols = vlite_program_and_columns(cfg, n, 2e-4)
    want, oracle_err = run_or_error(lambda: oracle_run(text, cols))
    e = engine_with(cols)
    got, engine_err = run_or_error(lambda: e.run_vdl(text)["results"])
    e.close()
    assert (oracle_err is None) == (engine_err is None), (oracle_err, engine_err)
    if oracle_err is None:
        assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE", "VDL_NO_EXPR_FUSION", "VDL_NO_FILTER_FUSION", "VDL_NO_PROJECTION", "VDL_NO_DIM_SCAN", "VDL_NO_GROUP_BATCH", "VDL_NO_FRONT_EXPR", "VDL_NO_REWRITE", "VDL_NO_GATHER_LIVE_CACHE"])
def test_sparse_vector_routes_agree(cfg, monkeypatch, mode):
    """The general executor keeps vectors that only hold values on a selection in compact form after selective
    filters, and runs chains of single-reader element-wise operators as one fused kernel.  Sparse forced on for
    every filter, sparse switched off, fusion switched off: same answers for every plan."""
    monkeypatch.setenv(mode, "1")
    for n in PLANS:
        text, cols = program_and_columns(cfg, n, 5e-4, seed=3)
        want = oracle_run(text, cols)
        e = engine_with(cols)
        got = e.run_vdl(text)["results"]
        e.close()
        assert got == want, (mode, n)
    fcfg = frontend.load_metadata(META, aggregation_strategy=("AggHierarchical", 5))
    for n in (1, 3, 10, 16):
        text, cols = program_and_columns(fcfg, n, 5e-4, seed=3)
        want = oracle_run(text, cols)
        e = engine_with(cols)
        got = e.run_vdl(text)["results"]
        e.close()
        assert got == want, (mode, "hierarchical", n)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [None, "VDL_NO_GROUP_BATCH", "VDL_NO_SPARSE", "VDL_NO_PROJECTION", "VDL_NO_SORTED_SHORTCUT"])
def test_plans_over_lineitem_clustered_by_order(cfg, monkeypatch, mode):
    """dbgen writes lineitem clustered by order: the join index into orders -- and every group key built from the order key
    (Q3, Q10, Q18) -- is then non-decreasing, the Partition's sortedness pass finds that, hands out identity ranks AND the run
    heads, and the folds of the GROUP BY run as one launch over them (launch_group_fold).  Every switch that takes another route
    gives the same answers, and all of them the oracle's."""
    if mode:
        monkeypatch.setenv(mode, "1")
    for n in (3, 5, 9, 10, 18, 20):
        text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
        cols = catalog.synth_columns(META, cfg, text, scale=5e-4, seed=3, clustered=("lineitem.lineitem_orders",))
        assert "lineitem.lineitem_orders" not in cols or (np.diff(cols["lineitem.lineitem_orders"]) >= 0).all()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        got = e.run_vdl(text)["results"]
        e.close()
        assert got == want, (mode, n)
        assert any(len(list(v.values())[0]) for v in want.values()) or n == 18, n


def test_which_plans_have_a_fused_front(cfg):
    """Host-only: plans that do not fuse as a whole but whose fact-side filters / FK lookups run as one projection scan
    (ProjPlan, vdl_fuse.h) handing sparse vectors to the per-operator executor."""
    import mplan2vdl_amd as m

    e = m.Engine(device=None)
    front = []
    for n in PLANS:
        d = e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)).describe()
        if "\nfused front:" in d:
            front.append(n)
    assert front == [3, 5, 9, 10, 11, 15, 16, 20]            # (Q5: 13 columns -- the take pass handles 16, the select pass sees 10 of them)
    # ... and whose dimension-side selections are scans of the dimension table themselves (PreludeItem::scan)
    dims = {}
    for n in PLANS:
        d = e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)).describe()
        tables = [l.split("one scan of ")[1].split(" sets bit")[0].strip() + ("*" if "semi-join" in l else "") for l in d.split("\n") if l.startswith("prelude") and "one scan of" in l]
        if tables:
            dims[n] = tables
    # (* = a semi-join set: Q4's EXISTS as a lineitem scan setting bits of orders rows)
    assert dims == {3: ["customer", "orders"], 4: ["orders", "lineitem*"], 5: ["orders", "region"], 9: ["part"], 10: ["orders"], 11: ["nation", "supplier"], 16: ["part", "supplier"], 19: ["part"], 20: ["part", "partsupp"]}
    q3 = e.parse(open(os.path.join(ROOT, "tests", "golden", "q3.vdl")).read()).describe()
    assert "\nfused front: one scan of lineitem" in q3 and "orders.o_orderdate[col1]" in q3 and "prelude0.bit[col1] in [1,1]" in q3


def test_which_plans_have_a_sharded_route(cfg):
    """Host-only: the plans vdl_exchange_spec accepts for a row-sharded lineitem, and the reasons the others get."""
    import mplan2vdl_amd as m

    e = m.Engine(device=None)
    verdict = {}
    for n in PLANS:
        p = e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg))
        if p.is_fused:
            verdict[n] = "fused"
            continue
        try:
            p.exchange_columns("lineitem")
            verdict[n] = "exchange"
        except m.VdlError as ex:
            verdict[n] = str(ex)
    # fused JOIN scans: Q14 (lineitem with part looked up through the join index), Q12 (orders' priority looked up, IN lists and
    # CASE conditions as condition columns), Q19 (a disjunction across lineitem and part columns as one condition column)
    # ... and Q4: its EXISTS as a semi-join set (a lineitem scan setting bits of orders rows) + a grouped orders scan
    assert sorted(n for n, v in verdict.items() if v == "fused") == [1, 4, 6, 12, 14, 19]
    assert sorted(n for n, v in verdict.items() if v == "exchange") == [3, 5, 9, 10]
    assert "does not treat every group by itself" in verdict[20]       # its semi-join over suppliers combines groups that may lie on different ranks
    assert "more than one Partition" in verdict[18]
    q4 = e.parse(frontend.compile_plan(open(os.path.join(META, "04.sql.mplan")).read(), cfg))
    with pytest.raises(m.VdlError, match="semi-join set"):     # built from every row of lineitem: fused, it has no sharded route
        q4.partial_spec()
    q4.set_fusion(False)
    with pytest.raises(m.VdlError, match="below the Partition"):
        q4.exchange_columns("lineitem")
    for n in (12, 19):                                         # ... and the routes they take with fusion off
        p = e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg))
        p.set_fusion(False)
        if n == 12:
            p.exchange_columns("lineitem")
        else:
            with pytest.raises(m.VdlError, match="no Partition"):
                p.exchange_columns("lineitem")
    # a join + ungrouped aggregate that does not fuse shards through its global folds instead (Q14 does too with fusion off)
    for n, folds in ((14, 2), (19, 1)):
        p = e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg))
        p.set_fusion(False)
        p.set_sharded_table("lineitem")
        nw, ops = p.partial_spec()
        assert nw == 3 * folds and ops == [m._lib.REDUCE_SUM, m._lib.REDUCE_MIN, m._lib.REDUCE_SUM] * folds
    p = e.parse(frontend.compile_plan(open(os.path.join(META, "18.sql.mplan")).read(), cfg))
    p.set_sharded_table("lineitem")
    with pytest.raises(m.VdlError, match="no global fold over table lineitem"):
        p.partial_spec()
