"""The multi-GPU path behind the C ABI (vdl_comm.cpp: vdl_comm_init*, vdl_run_sharded*) on ONE GPU.

RCCL refuses two ranks on one device, so N > 1 is driven here with the HOST transport: W ranks = W threads of this
process, one context each on device 0, each holding its row range; the all-gather / all-to-all callbacks rendezvous
through a barrier.  Everything else is the production path: vdl_run_sharded picks the route, packs, merges with the
device kernel, exchanges rows, runs the tail.  RCCL itself is exercised at world = 1 (real ncclCommInitRank,
all-gather-free merge, grouped send/receive to self is skipped by construction)."""
import os

import numpy as np
import pytest

import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, datagen, frontend, shard_rows
from conftest import ROOT, golden
from helpers import Rendezvous, engine_with, lineitem, oracle_run, run_ranks

pytestmark = pytest.mark.gpu
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")


def table_shards(cols, world, table):
    n = len(next(v for k, v in cols.items() if k.startswith(table + ".") and not k.endswith(".heap")))
    shards = []
    for r in range(world):
        r0, r1 = shard_rows(n, r, world)
        shards.append((r0, {k: (v[r0:r1] if k.startswith(table + ".") and not k.endswith(".heap") else v) for k, v in cols.items()}))
    return shards


def lineitem_shards(cols, world):
    return table_shards(cols, world, "lineitem")


def sharded_run(text, shards, world, table=None, pipelined=0, fuse=True):
    def work(rank, rv):
        r0, cols = shards[rank]
        e = engine_with(cols)
        e.comm_init_host(rank, world, *rv.transport(rank))
        assert e.comm_info() == (rank, world, "host")
        p = e.parse(text)
        p.set_fusion(fuse)
        if table:
            p.set_sharded_table(table)
        p.set_row_offset(r0)
        if pipelined:
            res = []
            for k in range(pipelined):
                p.run_sharded_begin(k & 1)
                if k:
                    res.append(p.run_sharded_end(1 - (k & 1))["results"])
            res.append(p.run_sharded_end((pipelined - 1) & 1)["results"])
            e.close()
            return res
        res = p.run_sharded()["results"]
        e.close()
        return res

    return run_ranks(world, work)


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("query", ["q6", "q1"])
def test_fused_plans_merge_through_one_all_gather(query, world):
    """Q6 (two SUM words) and Q1 (289 words: SUM, MIN / MAX and FoldChoose pairs): every rank ends with the whole answer."""
    text = golden(query + ".vdl")
    names = datagen.Q6_COLUMNS if query == "q6" else datagen.Q1_COLUMNS
    n = 300007
    whole = lineitem(names, n)
    want = oracle_run(text, whole)
    shards = [(lo, {k: v[lo:hi] for k, v in whole.items()}) for lo, hi in (shard_rows(n, r, world) for r in range(world))]
    for got in sharded_run(text, shards, world):
        assert got == want


def test_pipelined_begin_end_gives_every_query_its_answer():
    text = golden("q6.vdl")
    n = 200003
    whole = lineitem(datagen.Q6_COLUMNS, n)
    want = oracle_run(text, whole)
    shards = [(lo, {k: v[lo:hi] for k, v in whole.items()}) for lo, hi in (shard_rows(n, r, 2) for r in range(2))]
    for got in sharded_run(text, shards, 2, pipelined=5):
        assert got == [want] * 5


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("plan_no,fuse", [(12, True), (14, True), (19, True), (14, False), (19, False)])
def test_join_then_aggregate_merges_its_words(plan_no, fuse, world):
    """Fused join scans (Q12 grouped, Q14, Q19) merge their partial words like any fused plan; statement by statement the
    ungrouped ones merge their fold records.  Every rank ends with the whole answer."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan_no)).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=1e-3, seed=7)
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    for got in sharded_run(text, lineitem_shards(cols, world), world, table="lineitem", fuse=fuse):
        assert got == want


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("plan_no,fuse", [(3, True), (5, True), (9, True), (10, True), (12, False)])
def test_plans_with_a_partition_exchange_rows_and_concatenate(plan_no, fuse, world):
    """ONE all-gather of {status, counts} + ONE all-to-all of every column: the ranks' outputs, in rank order, are the
    unsharded result."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan_no)).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=6e-4, seed=3)
    want = oracle_run(text, cols)
    parts = sharded_run(text, lineitem_shards(cols, world), world, table="lineitem", fuse=fuse)
    got = {k: {name: sum((part[k][name] for part in parts), []) for name in v} for k, v in want.items()}
    assert got == want
    assert any(len(list(v.values())[0]) for v in want.values())


@pytest.mark.parametrize("world", [2, 3])
def test_semi_join_sets_are_merged_across_the_ranks(world):
    """TPC-H Q4 (EXISTS: a semi-join set over orders filled by a scan of lineitem, then a grouped scan of orders): lineitem sharded by
    rows, orders replicated -- every rank builds the set from its rows, ONE all-gather + OR kernel completes it, every rank ends
    with the whole answer.  Also random semi-join programs (the generator of test_random_semijoins, fact table t sharded):
    indices outside the dimension, fact tables shorter than the dimension (the union is clipped at the GLOBAL fact length)."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "04.sql.mplan")).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=1e-3, seed=7)
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    e = m.Engine(device=None)
    p = e.parse(text)
    p.set_sharded_table("lineitem")
    assert p.is_fused and p.sharded_route() == ("set", True)
    p.set_sharded_table("orders")
    with pytest.raises(m.VdlError, match="no sharded route"):
        p.sharded_route()
    for got in sharded_run(text, lineitem_shards(cols, world), world, table="lineitem"):
        assert got == want
    from test_random_semijoins import Gen
    ran = 0
    for seed in range(16):
        for short in (False, True):
            text, cols = Gen(seed, short_fact=short).build()
            p = e.parse(text)
            p.set_sharded_table("t")
            try:
                route = p.sharded_route()
            except m.VdlError:
                continue
            if route != ("set", True):
                continue
            nt = len(cols["t.a"])
            shards = []
            for r in range(world):
                r0, r1 = shard_rows(nt, r, world)
                shards.append((r0, {k: (v[r0:r1] if k.startswith("t.") else v) for k, v in cols.items()}))
            want = oracle_run(text, cols)
            for got in sharded_run(text, shards, world, table="t"):
                assert got == want, (seed, short)
            ran += 1
    assert ran >= 20, ran


def test_a_failed_prelude_on_the_set_route_reaches_every_rank():
    """The set route (Q4): the sets travel in an all-gather issued from a hook behind the prelude.  A rank whose prelude fails -- here
    its replicated orders table lacks the date column the dimension selection reads -- never reaches that hook; it must still meet
    its peers in the status exchange that precedes the sets, or they wait in the collective for ever.  Every rank returns an error,
    the failing one its own; the next query on the same communicator, with the catalog repaired, runs normally."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "04.sql.mplan")).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=1e-3, seed=7)
    want = oracle_run(text, cols)
    shards = lineitem_shards(cols, 2)
    errors, after = [], [None, None]

    def work(rank, rv):
        r0, c = shards[rank]
        e = engine_with(c)
        e.comm_init_host(rank, 2, *rv.transport(rank))
        p = e.parse(text)
        p.set_sharded_table("lineitem")
        p.set_row_offset(r0)
        assert p.sharded_route() == ("set", True)
        if rank == 1:
            e.drop("orders.o_orderdate")
        try:
            p.run_sharded()
        except m.VdlError as exc:
            errors.append((rank, str(exc)))
        if rank == 1:
            e.upload("orders.o_orderdate", c["orders.o_orderdate"])
        after[rank] = p.run_sharded()["results"]
        e.close()

    run_ranks(2, work, timeout=120)
    assert sorted(r for r, _ in errors) == [0, 1], errors
    assert any(r == 0 and "on rank 1" in msg for r, msg in errors), errors
    assert any(r == 1 and "o_orderdate" in msg for r, msg in errors), errors
    assert after == [want, want]


@pytest.mark.parametrize("world", [1, 2, 3])
def test_a_global_fold_beside_the_partition_is_merged_with_the_counts(world):
    """TPC-H Q11: GROUP BY ps_partkey HAVING sum(..) > (select sum(..) * 0.0001 ..) over the same filtered partsupp rows -- a Partition
    AND a global fold over the sharded table, which the tail compares the group sums with.  partsupp sharded by rows: the fold's
    record of every rank travels in the all-gather that carries the row counts, is merged on the host, and the tail of every rank
    reads the merged threshold; the ranks' outputs concatenate to the unsharded result."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "11.sql.mplan")).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=3)
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    e = m.Engine(device=None)
    p = e.parse(text)
    p.set_sharded_table("partsupp")
    assert p.sharded_route() == ("exchange", False)
    parts = sharded_run(text, table_shards(cols, world, "partsupp"), world, table="partsupp")
    got = {k: {name: sum((part[k][name] for part in parts), []) for name in v} for k, v in want.items()}
    assert got == want


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("clustered", [True, False])
def test_a_group_by_over_all_rows_that_feeds_a_position_set_runs_as_a_chain(world, clustered):
    """TPC-H Q18 (/root/reference/tests/tpch10noorder/18.sql.mplan) groups ALL lineitems by order, keeps the orders of the groups that
    pass its HAVING as a position set and scans lineitem a second time against that set.  The "chain" route: the rows travel to the
    owners of their key range (complete groups everywhere, whether the table is clustered by order or not), the positions every owner
    finds are all-gathered into the same set on every rank, the second scan runs on each rank's OWN rows and its survivors are
    gathered for the second GROUP BY.  Every rank ends with the whole answer, twice in a row."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "18.sql.mplan")).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=3, clustered=("lineitem.lineitem_orders",) if clustered else ())
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    shards = table_shards(cols, world, "lineitem")

    def work(rank, rv):
        r0, c = shards[rank]
        e = engine_with(c)
        e.comm_init_host(rank, world, *rv.transport(rank))
        p = e.parse(text)
        p.set_sharded_table("lineitem")
        p.set_row_offset(r0)
        assert p.sharded_route() == ("chain", True)
        first = p.run_sharded()["results"]
        second = p.run_sharded()["results"]
        plain = None
        if rank == 0:                                            # the plan is what it was: unsharded runs of it are untouched
            e2 = engine_with(cols)
            plain = e2.parse(text).run()["results"]
            e2.close()
        e.close()
        return first, second, plain

    got = run_ranks(world, work, timeout=120)
    assert [g[:2] for g in got] == [(want, want)] * world
    assert got[0][2] == want


def chain_program(nd, threshold, second_scan):
    """GROUP BY f.k over ALL rows of f, HAVING sum(f.v) > threshold, the keys of the groups that pass as a position set; then the
    dimension rows in the set (no second scan), or the rows of f whose key is in the set, grouped by f.a (second scan)."""
    from helpers import prog
    lines = ["1,Load,f.k", "2,Project,val,Id 1,k", "3,Load,f.v", "4,Project,val,Id 3,v", "5,Load,d.g", "6,Project,val,Id 5,g",
             "7,RangeC,val,0,%d,1" % nd, "8,Partition,val,Id 2,val,Id 7,val",
             "9,RangeV,val,0,Id 2,1", "10,Scatter,Id 2,Id 9,val,Id 8,val", "11,RangeV,val,0,Id 4,1", "12,Scatter,Id 4,Id 11,val,Id 8,val",
             "13,FoldSum,val,Id 10,val,Id 12,val", "14,RangeV,val,%d,Id 13,0" % threshold, "15,Greater,val,Id 13,val,Id 14,val",
             "16,RangeV,val,0,Id 15,1", "17,FoldSelect,val,Id 16,val,Id 15,val", "18,Gather,Id 10,Id 17,val",      # the keys of the groups that pass
             "19,RangeV,val,1,Id 18,0", "20,RangeV,val,0,Id 19,1", "21,Scatter,Id 19,Id 20,val,Id 18,val"]          # ones at those positions: the set
    if not second_scan:
        lines += ["22,RangeV,val,0,Id 21,1", "23,FoldSelect,val,Id 22,val,Id 21,val", "24,Gather,Id 6,Id 23,val",
                  "25,Project,g,Id 24,val", "26,MaterializeCompact,Id 25", "27,Project,row,Id 23,val", "28,MaterializeCompact,Id 27"]
    else:
        lines += ["22,Load,f.a", "23,Project,val,Id 22,a",
                  "24,Gather,Id 21,Id 2,val",                                                            # per row of f: is its key in the set?
                  "25,RangeV,val,0,Id 24,1", "26,FoldSelect,val,Id 25,val,Id 24,val",
                  "27,Gather,Id 23,Id 26,val", "28,Gather,Id 4,Id 26,val", "29,Gather,Id 6,Id 27,val",         # a, v of the survivors; d.g by a
                  "30,RangeC,val,0,%d,1" % nd, "31,Partition,val,Id 27,val,Id 30,val",
                  "32,RangeV,val,0,Id 27,1", "33,Scatter,Id 27,Id 32,val,Id 31,val", "34,Scatter,Id 28,Id 32,val,Id 31,val", "35,Scatter,Id 29,Id 32,val,Id 31,val",
                  "36,FoldSum,val,Id 33,val,Id 34,val", "37,FoldChoose,val,Id 33,val,Id 35,val", "38,FoldCount,val,Id 33,val,Id 34,val",
                  "39,Project,total,Id 36,val", "40,MaterializeCompact,Id 39", "41,Project,g,Id 37,val", "42,MaterializeCompact,Id 41",
                  "43,Project,rows,Id 38,val", "44,MaterializeCompact,Id 43"]
    return prog(*lines)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("second_scan", [False, True])
def test_the_chain_route_with_and_without_a_second_scan(world, second_scan):
    """The chain route on programs written for it: the groups' keys straddle the ranks' row ranges (random order), one rank's groups
    contribute nothing to the set, and the rest either reads only the replicated table through the set (no second cut: it runs on
    every rank as it stands) or scans the sharded table again and groups the survivors by another column."""
    rng = np.random.default_rng(5 + world + 10 * second_scan)
    n, nd = 12000, 400
    cols = {"f.k": rng.integers(0, nd, n).astype(np.int64), "f.v": rng.integers(0, 100, n).astype(np.int64), "f.a": rng.integers(0, nd, n).astype(np.int64),
            "d.g": rng.integers(0, 1000, nd).astype(np.int64)}
    cols["f.k"][cols["f.k"] < nd // 3] += nd // 3                  # no key in the first third of the domain: the owner of that range finds nothing
    sums = np.bincount(cols["f.k"], weights=cols["f.v"], minlength=nd)
    threshold = int(np.sort(sums)[-40])                               # about forty groups pass
    text = chain_program(nd, threshold, second_scan)
    want = oracle_run(text, cols)
    assert all(len(list(v.values())[0]) > 0 for v in want.values())
    shards = table_shards(cols, world, "f")

    def work(rank, rv):
        r0, c = shards[rank]
        e = engine_with(c)
        e.comm_init_host(rank, world, *rv.transport(rank))
        p = e.parse(text)
        p.set_sharded_table("f")
        p.set_row_offset(r0)
        assert p.sharded_route() == ("chain", True)
        res = p.run_sharded()["results"]
        e.close()
        return res

    assert run_ranks(world, work, timeout=120) == [want] * world


@pytest.mark.parametrize("case", ["empty set", "rank without rows", "one key"])
def test_the_chain_route_at_its_edges(case):
    """No group passes the HAVING (an empty set: nothing to gather, nothing survives the second scan); more ranks than rows (ranks whose
    shard is empty take part in every collective); every row carries the same key (one owner gets everything, the others nothing)."""
    rng = np.random.default_rng(len(case))
    world = 3
    n, nd = (2, 50) if case == "rank without rows" else (5000, 120)
    cols = {"f.k": rng.integers(0, nd, n).astype(np.int64), "f.v": rng.integers(1, 100, n).astype(np.int64), "f.a": rng.integers(0, nd, n).astype(np.int64),
            "d.g": rng.integers(0, 1000, nd).astype(np.int64)}
    if case == "one key":
        cols["f.k"][:] = 77
    threshold = 10 ** 9 if case == "empty set" else 0
    for second_scan in (False, True):
        text = chain_program(nd, threshold, second_scan)
        want = oracle_run(text, cols)
        shards = table_shards(cols, world, "f")

        def work(rank, rv):
            r0, c = shards[rank]
            e = engine_with(c)
            e.comm_init_host(rank, world, *rv.transport(rank))
            p = e.parse(text)
            p.set_sharded_table("f")
            p.set_row_offset(r0)
            assert p.sharded_route() == ("chain", True)
            res = p.run_sharded()["results"]
            e.close()
            return res

        assert run_ranks(world, work, timeout=120) == [want] * world, (case, second_scan)


@pytest.mark.parametrize("world", [2])
def test_a_failure_inside_the_chain_reaches_every_rank(world):
    """One rank lacks a column the second scan reads: it fails in its local phase of stage 2, says so in the status exchange that precedes
    the gather of the survivors, and every rank stops with an error instead of waiting in a collective."""
    rng = np.random.default_rng(77)
    n, nd = 6000, 200
    cols = {"f.k": rng.integers(0, nd, n).astype(np.int64), "f.v": rng.integers(0, 100, n).astype(np.int64), "f.a": rng.integers(0, nd, n).astype(np.int64),
            "d.g": rng.integers(0, 1000, nd).astype(np.int64)}
    text = chain_program(nd, 2000, True)
    shards = table_shards(cols, world, "f")

    def work(rank, rv):
        r0, c = shards[rank]
        if rank == 1:
            c = {k: v for k, v in c.items() if k != "f.a"}
        e = engine_with(c)
        e.comm_init_host(rank, world, *rv.transport(rank))
        p = e.parse(text)
        p.set_sharded_table("f")
        p.set_row_offset(r0)
        try:
            p.run_sharded()
            return "ran"
        except m.VdlError as exc:
            return str(exc)
        finally:
            e.close()

    got = run_ranks(world, work, timeout=120)
    assert "f.a" in got[1] and "rank 1" in got[0], got


@pytest.mark.parametrize("world", [2, 3])
def test_a_plan_no_route_serves_runs_on_the_replicated_table(world, monkeypatch):
    """TPC-H Q18 groups ALL lineitems by order before it filters anything, feeds a semi-join set from the groups and scans lineitem a
    second time: no fold, no exchange, no front -- and, with the chain route switched off (VDL_NO_CHAIN_ROUTE), nothing that scales.
    The last resort gathers the lineitem columns it loads once (rank after rank = row
    order) and runs the whole query on every rank; the second run of the same plan moves nothing."""
    monkeypatch.setenv("VDL_NO_CHAIN_ROUTE", "1")
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "18.sql.mplan")).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=3, clustered=("lineitem.lineitem_orders",))
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    shards = table_shards(cols, world, "lineitem")

    def work(rank, rv):
        r0, c = shards[rank]
        e = engine_with(c)
        e.comm_init_host(rank, world, *rv.transport(rank))
        p = e.parse(text)
        p.set_sharded_table("lineitem")
        p.set_row_offset(r0)
        assert p.sharded_route() == ("replicate", True)
        first = p.run_sharded()["results"]
        second = p.run_sharded()["results"]
        # the catalog of ONE rank moves (the same column uploaded again): whether the table is gathered again is agreed by all ranks
        # on every run, so the others join the collectives instead of leaving this rank waiting in them
        if rank == world - 1:
            e.upload("lineitem.l_quantity", c["lineitem.l_quantity"])
        third = p.run_sharded()["results"]
        e.close()
        return first, second, third

    assert run_ranks(world, work, timeout=120) == [(want, want, want)] * world


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("plan,table", [(16, "partsupp"), (15, "lineitem"), (20, "lineitem")])
def test_two_partitions_run_above_the_gathered_front(plan, table, world):
    """TPC-H Q16 (count(distinct ps_suppkey) under a GROUP BY: two Partitions) has no exchange route; its work on partsupp is a
    fused front, so the survivors' vectors of every rank are all-gathered -- rank after rank = row order -- and every rank runs
    the two Partitions on the complete vectors: the "front" route, whole answer everywhere.  Q15 (a global max over the grouped sums,
    then the suppliers that reach it) goes the same way once the column its FoldChoose'd row ids looked up travels through the
    fold itself (rewrite_program, vdl_fuse.cpp).  So does Q20 (round 4): one Partition, but its tail feeds a semi-join set over suppliers
    from the groups, which the exchange route's concatenation cannot serve (the balanced key cut made that visible: a supplier with
    qualifying groups on two ranks came out twice)."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan)).read(), cfg)
    cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=3)
    want = oracle_run(text, cols)
    assert any(len(list(v.values())[0]) for v in want.values())
    e = m.Engine(device=None)
    p = e.parse(text)
    p.set_sharded_table(table)
    assert p.sharded_route() == ("front", True)
    parts = sharded_run(text, table_shards(cols, world, table), world, table=table)
    assert parts == [want] * world


@pytest.mark.parametrize("world", [2, 3])
def test_the_front_route_serves_any_tail_over_a_filtered_fact_table(world):
    """A filter over the sharded table whose condition names the ROW ID (the reference's anti-join shape), a lookup in a replicated table, then two GROUP BYs one above
    the other (the second over the first's result) and a global fold of the group sums: nothing the exchange route can do, all of
    it above a front.  One rank's share is empty; the row-id condition must count from the table's first row on every rank."""
    from helpers import prog
    rng = np.random.default_rng(11 + world)
    n, nd = 9000, 300
    cols = {"f.k": rng.integers(0, nd, n).astype(np.int64), "f.a": rng.integers(0, 50, n).astype(np.int64), "f.v": rng.integers(-100, 100, n).astype(np.int64),
            "d.g": rng.integers(0, 12, nd).astype(np.int64), "d.pk": np.arange(nd, dtype=np.int64)}
    cols["f.a"][[4000, 8500]] = 0                              # both pass a < 7: row 4000 of the TABLE goes, row 4000 of the second of two shards (8500) stays
    cols["f.v"][[4000, 8500]] = [77, -99]
    text = prog("1,Load,f.k", "2,Project,val,Id 1,k", "3,Load,f.a", "4,Project,val,Id 3,a", "5,Load,f.v", "6,Project,val,Id 5,v",
                "7,Load,d.g", "8,Project,val,Id 7,g",
                "9,RangeV,val,7,Id 4,0", "10,Greater,val,Id 9,val,Id 4,val",                       # a < 7
                "11,RangeV,val,0,Id 4,1", "12,RangeV,val,4000,Id 4,0", "13,Subtract,val,Id 12,val,Id 11,val",  # 4000 - row id as a truth value: every row but row 4000
                "14,LogicalAnd,val,Id 10,val,Id 13,val",
                "15,RangeV,val,0,Id 14,1", "16,FoldSelect,val,Id 15,val,Id 14,val",
                "17,Gather,Id 2,Id 16,val", "18,Gather,Id 6,Id 16,val", "19,Gather,Id 8,Id 17,val",        # key, value, dimension group of the survivors
                "20,RangeC,val,0,%d,1" % nd, "21,Partition,val,Id 17,val,Id 20,val",
                "22,RangeV,val,0,Id 17,1", "23,Scatter,Id 17,Id 22,val,Id 21,val", "24,Scatter,Id 18,Id 22,val,Id 21,val", "25,Scatter,Id 19,Id 22,val,Id 21,val",
                "26,FoldSum,val,Id 23,val,Id 24,val", "27,FoldChoose,val,Id 23,val,Id 25,val",                # per key: sum(v), its group
                "28,RangeV,val,0,Id 27,1", "29,FoldSelect,val,Id 28,val,Id 27,val",                            # the groups' slots (valid where a run starts) ...
                "30,RangeC,val,0,12,1", "31,Partition,val,Id 27,val,Id 30,val",
                "32,RangeV,val,0,Id 27,1", "33,Scatter,Id 27,Id 32,val,Id 31,val", "34,Scatter,Id 26,Id 32,val,Id 31,val",
                "35,FoldSum,val,Id 33,val,Id 34,val", "36,FoldCount,val,Id 33,val,Id 34,val",                 # per dimension group: sum of sums, number of keys
                "37,Project,total,Id 35,val", "38,MaterializeCompact,Id 37", "39,Project,keys,Id 36,val", "40,MaterializeCompact,Id 39",
                "41,RangeV,val,0,Id 26,0", "42,FoldMax,val,Id 41,val,Id 26,val", "43,Project,best,Id 42,val", "44,MaterializeCompact,Id 43")
    want = oracle_run(text, cols)
    assert len(want["tmp38"][".total"]) > 3
    e = m.Engine(device=None)
    p = e.parse(text)
    p.set_sharded_table("f")
    assert p.sharded_route() == ("front", True), p.describe()
    shards = table_shards(cols, world, "f")
    assert sharded_run(text, shards, world, table="f") == [want] * world
    # a rank without rows
    cut = [(0, {k: (v[:0] if k.startswith("f.") else v) for k, v in cols.items()})] + [(0, cols)] + ([(n, {k: (v[:0] if k.startswith("f.") else v) for k, v in cols.items()})] if world == 3 else [])
    assert sharded_run(text, cut, world, table="f") == [want] * world


@pytest.mark.parametrize("world", [2, 3])
def test_a_row_id_condition_counts_from_the_tables_first_row_on_every_rank(world):
    """`k - row id` as a truth value (every row but row k: what the reference's anti-join compiles to) inside a FUSED scan of the
    sharded table: rank r tests the row's number in the table, not in its shard (one row of the whole table goes, not one per rank)."""
    from helpers import prog
    rng = np.random.default_rng(3)
    n = 9000
    cols = {"f.a": rng.integers(0, 50, n).astype(np.int64), "f.v": rng.integers(1, 100, n).astype(np.int64)}
    cols["f.a"][[1000, 4000, 5500, 7000]] = 0
    text = prog("1,Load,f.a", "2,Project,val,Id 1,a", "3,Load,f.v", "4,Project,val,Id 3,v",
                "5,RangeV,val,7,Id 2,0", "6,Greater,val,Id 5,val,Id 2,val",
                "7,RangeV,val,0,Id 2,1", "8,RangeV,val,1000,Id 2,0", "9,Subtract,val,Id 8,val,Id 7,val",
                "10,LogicalAnd,val,Id 6,val,Id 9,val",
                "11,RangeV,val,0,Id 10,1", "12,FoldSelect,val,Id 11,val,Id 10,val", "13,Gather,Id 4,Id 12,val",
                "14,RangeV,val,0,Id 13,0", "15,FoldSum,val,Id 14,val,Id 13,val", "16,MaterializeCompact,Id 15",
                "17,FoldCount,val,Id 14,val,Id 13,val", "18,MaterializeCompact,Id 17")
    want = oracle_run(text, cols)
    e = m.Engine(device=None)
    p = e.parse(text)
    assert p.is_fused, p.describe()
    assert sharded_run(text, table_shards(cols, world, "f"), world, table="f") == [want] * world


@pytest.mark.parametrize("world", [2, 3, 5])
def test_the_key_domain_is_cut_where_the_data_is(world):
    """Keys that fill the lowest eighth of their declared domain (and unevenly: three quarters of the rows in the lowest tenth of the keys
    in use): the ranks' key histograms (4096 slices of the declared domain) travel with the counts, every rank cuts the domain at the
    same data-driven points, and each ends with about 1 / world of the ROWS -- with the declared domain cut evenly every row would go to
    rank 0.  The ranks' outputs concatenate to the oracle's answer (key order), and a tail that treats every group by itself is what the
    analysis insists on."""
    from helpers import prog
    rng = np.random.default_rng(17 + world)
    n = 60000
    k = np.where(rng.random(n) < 0.75, rng.integers(0, 3000, n), rng.integers(3000, 30000, n)).astype(np.int64) + 77
    cols = {"t.k": k, "t.x": rng.integers(-50, 50, n).astype(np.int64), "t.f": (rng.integers(0, 10, n) < 8).astype(np.int64)}
    text = prog("1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.x", "4,Project,val,Id 3,x", "5,Load,t.f", "6,Project,val,Id 5,f",
                "7,RangeV,val,0,Id 6,1", "8,FoldSelect,val,Id 7,val,Id 6,val", "9,Gather,Id 2,Id 8,val", "10,Gather,Id 4,Id 8,val",
                "11,RangeC,val,0,%d,1" % (1 << 18), "12,Partition,val,Id 9,val,Id 11,val",
                "13,RangeV,val,0,Id 9,1", "14,Scatter,Id 9,Id 13,val,Id 12,val", "15,Scatter,Id 10,Id 13,val,Id 12,val",
                "16,FoldSum,val,Id 14,val,Id 15,val", "17,FoldChoose,val,Id 14,val,Id 14,val", "18,FoldCount,val,Id 14,val,Id 15,val",
                "19,Project,s,Id 16,val", "20,MaterializeCompact,Id 19", "21,Project,key,Id 17,val", "22,MaterializeCompact,Id 21",
                "23,Project,c,Id 18,val", "24,MaterializeCompact,Id 23")
    want = oracle_run(text, cols)
    shards = [(lo, {kk: v[lo:hi] for kk, v in cols.items()}) for lo, hi in (shard_rows(n, r, world) for r in range(world))]
    parts = sharded_run(text, shards, world, table="t", fuse=False)
    got = {kk: {name: sum((part[kk][name] for part in parts), []) for name in v} for kk, v in want.items()}
    assert got == want
    rows = [sum(part["tmp24"][".c"]) for part in parts]                   # rows each rank received = the sum of its groups' counts
    assert sum(rows) == int((cols["t.f"] > 0).sum())
    assert min(rows) > 0.7 * sum(rows) / world and max(rows) < 1.3 * sum(rows) / world, rows
    # ... and a tail that folds ACROSS groups (a global maximum over the group sums) is refused by the exchange analysis
    e = m.Engine(device=None)
    p = e.parse(text + prog("25,RangeV,val,0,Id 16,0", "26,FoldMax,val,Id 25,val,Id 16,val", "27,MaterializeCompact,Id 26"))
    p.set_fusion(False)
    with pytest.raises(m.VdlError, match="does not treat every group by itself"):
        p.exchange_columns("t")


def test_a_failure_on_one_rank_is_reported_on_every_rank():
    """A key outside the Partition pivots on ONE rank (the local phase of that rank fails): the status travels with the
    counts, nobody is left waiting in a collective, every rank returns an error."""
    from helpers import prog

    text = prog("1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.x", "4,Project,val,Id 3,x",
                "5,RangeC,val,0,64,1", "6,Partition,val,Id 2,val,Id 5,val",
                "7,RangeV,val,0,Id 2,1", "8,Scatter,Id 2,Id 7,val,Id 6,val", "9,Scatter,Id 4,Id 7,val,Id 6,val",
                "10,FoldSum,val,Id 8,val,Id 9,val", "11,MaterializeCompact,Id 10")
    rng = np.random.default_rng(5)
    cols = {"t.k": rng.integers(0, 64, 4000).astype(np.int64), "t.x": rng.integers(0, 100, 4000).astype(np.int64)}
    shards = [(lo, {k: v[lo:hi] for k, v in cols.items()}) for lo, hi in (shard_rows(4000, r, 2) for r in range(2))]
    want = oracle_run(text, cols)["tmp11"][".val"]
    assert [g["tmp11"][".val"] for g in sharded_run(text, shards, 2, table="t")] == [want, want]       # fused: the fold route, whole answer everywhere
    good = sharded_run(text, shards, 2, table="t", fuse=False)                                         # statement by statement: rows are exchanged
    assert sum((g["tmp11"][".val"] for g in good), []) == want
    shards[1][1]["t.k"] = shards[1][1]["t.k"].copy()
    shards[1][1]["t.k"][7] = 1000                          # outside RangeC 0 64 1
    errors = []

    def work(rank, rv):
        r0, c = shards[rank]
        e = engine_with(c)
        e.comm_init_host(rank, 2, *rv.transport(rank))
        p = e.parse(text)
        p.set_fusion(False)
        p.set_sharded_table("t")
        try:
            p.run_sharded()
        except m.VdlError as exc:
            errors.append((rank, str(exc)))
        e.close()

    run_ranks(2, work)
    assert sorted(r for r, _ in errors) == [0, 1]
    assert any("outside the pivots" in msg for _, msg in errors) and any("failed on rank 1" in msg for _, msg in errors)


@pytest.mark.parametrize("pipelined", [False, True])
def test_a_failed_local_phase_on_the_fold_route_reaches_every_rank(pipelined):
    """The fold route (one all-gather of the partial words): a rank whose local phase fails -- here its catalog lacks a column --
    still takes part in the collective with its status, so its peer is not left waiting; the failing rank reports its own error,
    the other one names the rank.  The next query on the same communicator, with the catalog repaired, runs normally."""
    text = golden("q6.vdl")
    n = 100003
    whole = lineitem(datagen.Q6_COLUMNS, n)
    want = oracle_run(text, whole)
    shards = [(lo, {k: v[lo:hi] for k, v in whole.items()}) for lo, hi in (shard_rows(n, r, 2) for r in range(2))]
    errors, after = [], [None, None]

    def work(rank, rv):
        r0, cols = shards[rank]
        e = engine_with(cols)
        e.comm_init_host(rank, 2, *rv.transport(rank))
        p = e.parse(text)
        p.set_row_offset(r0)
        if rank == 1:
            e.drop("lineitem.l_discount")
        try:
            if pipelined:
                p.run_sharded_begin(0)
                p.run_sharded_end(0)
            else:
                p.run_sharded()
        except m.VdlError as exc:
            errors.append((rank, str(exc)))
        if rank == 1:
            e.upload("lineitem.l_discount", cols["lineitem.l_discount"])
        after[rank] = p.run_sharded()["results"]
        e.close()

    run_ranks(2, work)
    assert sorted(r for r, _ in errors) == [0, 1], errors
    assert any(r == 0 and "failed on rank 1" in msg for r, msg in errors), errors
    assert any(r == 1 and "l_discount" in msg for r, msg in errors), errors
    assert after == [want, want]


def test_rccl_communicator_of_one_rank_runs_both_routes(q6_text):
    """Real RCCL on this box's one GPU: ncclGetUniqueId, ncclCommInitRank, the fold route and the exchange route."""
    n = 100003
    cols = lineitem(datagen.Q6_COLUMNS, n)
    e = engine_with(cols)
    e.comm_init_rccl(0, 1, e.comm_unique_id())
    assert e.comm_info() == (0, 1, "rccl")
    p = e.parse(q6_text)
    assert p.run_sharded()["results"] == oracle_run(q6_text, cols)
    for k in range(4):                                       # pipelined slots on the communication stream
        p.run_sharded_begin(k & 1)
        assert p.run_sharded_end(k & 1)["results"] == oracle_run(q6_text, cols)
    e.close()
    t = datagen.q3_tables(2000)
    e = engine_with(t)
    e.comm_init_rccl(0, 1, e.comm_unique_id())
    p = e.parse(golden("q3.vdl"))
    p.set_sharded_table("lineitem")
    assert p.run_sharded()["results"] == oracle_run(golden("q3.vdl"), t)
    e.close()


def test_rccl_communicator_of_one_rank_runs_the_set_and_front_routes(monkeypatch):
    """The other two routes through real RCCL with one rank: Q4's all-gather of the semi-join set, and -- with
    VDL_FRONT_ROUTE_ALWAYS, which keeps a one-rank run on the collectives -- Q16's and Q15's grouped send / receive of the front's vectors, Q18's chain (its three
    gathers) and, with that switched off, the last resort."""
    cfg = frontend.load_metadata(META)
    monkeypatch.setenv("VDL_FRONT_ROUTE_ALWAYS", "1")
    for plan, table, route in ((4, "lineitem", "set"), (16, "partsupp", "front"), (15, "lineitem", "front"), (18, "lineitem", "chain"), (18, "lineitem", "replicate")):
        text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan)).read(), cfg)
        cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=3)
        if route == "replicate":
            monkeypatch.setenv("VDL_NO_CHAIN_ROUTE", "1")
        e = engine_with(cols)
        e.comm_init_rccl(0, 1, e.comm_unique_id())
        p = e.parse(text)
        p.set_sharded_table(table)
        assert p.sharded_route()[0] == route
        assert p.run_sharded()["results"] == oracle_run(text, cols), plan
        e.close()
