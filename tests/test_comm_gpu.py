"""The multi-GPU path behind the C ABI (vdl_comm.cpp: vdl_comm_init*, vdl_run_sharded*) on ONE GPU.

RCCL refuses two ranks on one device, so N > 1 is driven here with the HOST transport: W ranks = W threads of this
process, one context each on device 0, each holding its row range; the all-gather / all-to-all callbacks rendezvous
through a barrier.  Everything else is the production path: vdl_run_sharded picks the route, packs, merges with the
device kernel, exchanges rows, runs the tail.  RCCL itself is exercised at world = 1 (real ncclCommInitRank,
all-gather-free merge, grouped send/receive to self is skipped by construction)."""
import os
import threading

import numpy as np
import pytest

import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, datagen, frontend, shard_rows
from conftest import ROOT, golden
from helpers import engine_with, lineitem, oracle_run

pytestmark = pytest.mark.gpu
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")


class Rendezvous:
    """In-process stand-in for a host collective library (TEST ONLY)."""

    def __init__(self, world):
        self.world, self.barrier = world, threading.Barrier(world)
        self.slots = [None] * world

    def transport(self, rank):
        def all_gather(send):
            self.slots[rank] = send
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out

        def all_to_all(pieces):
            self.slots[rank] = pieces
            self.barrier.wait()
            out = [self.slots[src][rank] for src in range(self.world)]
            self.barrier.wait()
            return out

        return all_gather, all_to_all


def run_ranks(world, work):
    """work(rank, rendezvous) in `world` threads; returns the per-rank results, re-raising the first failure."""
    rv = Rendezvous(world)
    out, errs = [None] * world, []

    def body(rank):
        try:
            out[rank] = work(rank, rv)
        except BaseException as exc:          # noqa: BLE001
            errs.append(exc)
            rv.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    if errs:
        raise errs[0]
    return out


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
@pytest.mark.parametrize("plan_no,fuse", [(3, True), (5, True), (9, True), (10, True), (20, True), (12, False)])
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
