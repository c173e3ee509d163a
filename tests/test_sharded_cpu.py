"""N > 1 path on CPU: two gloo ranks each fold their own row range, the partial words are merged
with all-reduce exactly as bench.py does with RCCL, and both ranks finalise the full answer."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mplan2vdl_amd as m
from mplan2vdl_amd import _lib, datagen


def test_shard_rows_cover_everything_once():
    for n in (0, 1, 7, 600037902):
        for world in (1, 2, 3, 8):
            r = [m.shard_rows(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


class OracleShardRunner:
    """Stands in for engine.Plan on a CPU-only box (TEST ONLY): computes the partial words of the
    Q6 scan {selected rows, sum(ep*disc), max(ep), min(disc)} for one row range with the oracle."""

    def __init__(self, lo, hi):
        import oracle

        cols = {c: datagen.generate(datagen.LINEITEM[c], lo, hi - lo) for c in datagen.Q6_COLUMNS}
        sd, di, qt, ep = [cols[c] for c in datagen.Q6_COLUMNS]
        self.rev, self.cnt = oracle.sql_q6(sd, di, qt, ep)
        sel = (sd >= 728294) & (sd < 728659) & (di >= 5) & (di <= 7) & (qt < 2400)
        self.mx = int(ep[sel].max()) if sel.any() else np.iinfo(np.int64).min
        self.mn = int(di[sel].min()) if sel.any() else np.iinfo(np.int64).max
        self.buf = None

    def partial_spec(self):
        return 4, [_lib.REDUCE_SUM, _lib.REDUCE_SUM, _lib.REDUCE_MAX, _lib.REDUCE_MIN]

    def run_local(self, ptr):
        self.buf[:4] = torch.tensor([self.cnt, self.rev, self.mx, self.mn], dtype=torch.int64)

    def finalize(self, ptr):
        w = [int(x) for x in self.buf[:4]]
        return {"results": {"tmp42": {".revenue": [w[1]] if w[0] else []}, "tmpX": {".max": [w[2]]}, "tmpY": {".min": [w[3]]}},
                "timings": {}, "count": w[0]}


class PipelinedSumRunner:
    """CPU stand-in (TEST ONLY) for a Q6-shaped plan driven through run_pipelined: two SUM words, every
    query adds its sequence number so that a mixed-up buffer or slot shows."""

    def __init__(self, lo, hi):
        cols = {c: datagen.generate(datagen.LINEITEM[c], lo, hi - lo) for c in datagen.Q6_COLUMNS}
        import oracle

        self.rev, self.cnt = oracle.sql_q6(*[cols[c] for c in datagen.Q6_COLUMNS])
        self.bufs, self.seq, self.slots, self.log = {}, 0, {}, []

    def partial_spec(self):
        return 2, [_lib.REDUCE_SUM, _lib.REDUCE_SUM]

    def run_local(self, ptr):
        self.bufs[ptr][:2] = torch.tensor([self.cnt, self.rev + self.seq], dtype=torch.int64)
        self.seq += 1

    def finalize_begin(self, ptr, slot):
        assert slot not in self.slots, "slot reused before it was read"
        self.slots[slot] = [int(x) for x in self.bufs[ptr][:2]]

    def finalize_end(self, slot):
        return self.slots.pop(slot)


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = m.shard_rows(n, rank, world)
    runner = OracleShardRunner(lo, hi)
    buf = torch.zeros(4, dtype=torch.int64)
    runner.buf = buf
    query = m.ShardedQuery(runner, buf, dist)
    out = None
    for _ in range(2):
        out = query.step()
    # pipelined drive, merge overlapped with the next query and not: every query's own answer, in order
    pr = PipelinedSumRunner(lo, hi)
    bufs = [torch.zeros(2, dtype=torch.int64), torch.zeros(2, dtype=torch.int64)]
    pr.bufs = {b.data_ptr(): b for b in bufs}
    pq = m.ShardedQuery(pr, bufs[0], dist)
    out["pipelined"] = []
    for overlap in (True, False):
        for steps in (1, 2, 5):
            got = []
            pr.seq = 0
            pq.run_pipelined(steps, bufs, got.append, overlap_merge=overlap)
            out["pipelined"].append(got)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_merge_equals_single_rank():
    n = 200001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = OracleShardRunner(0, n)
    want = {"tmp42": {".revenue": [single.rev]}, "tmpX": {".max": [single.mx]}, "tmpY": {".min": [single.mn]}}
    assert outs[0]["results"] == want and outs[1]["results"] == want
    assert outs[0]["count"] == single.cnt
    want = []
    for _ in (True, False):
        for steps in (1, 2, 5):
            want.append([[single.cnt, single.rev + 2 * k] for k in range(steps)])      # both ranks add k
    assert outs[0]["pipelined"] == want and outs[1]["pipelined"] == want


def test_merge_partials_mixed_ops_single_process():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(31000 + (os.getpid() % 2000))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        buf = torch.tensor([3, 10, 7, 2], dtype=torch.int64)
        m.merge_partials(buf, [_lib.REDUCE_SUM, _lib.REDUCE_SUM, _lib.REDUCE_MAX, _lib.REDUCE_MIN], dist)
        assert buf.tolist() == [3, 10, 7, 2]
    finally:
        dist.destroy_process_group()


class GroupedShardRunner:
    """CPU stand-in (TEST ONLY) for a grouped plan: per bucket {count, FIRST(col a), SUM(col a)}."""

    G = 5

    def __init__(self, lo, hi, n):
        rng = np.random.default_rng(123)
        self.key = rng.integers(0, self.G, size=n)[lo:hi]
        self.a = rng.integers(-50, 50, size=n)[lo:hi].astype(np.int64)
        self.lo, self.hi = lo, hi
        self.buf = None

    def partial_spec(self):
        ops = []
        for _ in range(self.G):
            ops += [_lib.REDUCE_SUM, _lib.REDUCE_FIRST, _lib.REDUCE_SUM]
        return 3 * self.G, ops

    def run_local(self, ptr):
        w = []
        for g in range(self.G):
            m_ = self.key == g
            rows = np.nonzero(m_)[0]
            w += [int(m_.sum()), int(self.lo + rows[0]) if rows.size else np.iinfo(np.int64).max, int(self.a[m_].sum())]
        self.buf[: len(w)] = torch.tensor(w, dtype=torch.int64)

    def resolve_first(self, ptr):
        for g in range(self.G):
            r = int(self.buf[3 * g + 1])
            live = int(self.buf[3 * g]) > 0
            self.buf[3 * g + 1] = int(self.a[r - self.lo]) if (live and self.lo <= r < self.hi) else 0

    def finalize(self, ptr):
        return [int(x) for x in self.buf[: 3 * self.G]]


def _grouped_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = m.shard_rows(n, rank, world)
    runner = GroupedShardRunner(lo, hi, n)
    buf = torch.zeros(3 * runner.G, dtype=torch.int64)
    runner.buf = buf
    out = m.ShardedQuery(runner, buf, dist).step()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_merge_of_foldchoose_words():
    n = 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grouped_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = GroupedShardRunner(0, n, n)
    single.buf = torch.zeros(3 * single.G, dtype=torch.int64)
    single.run_local(0)
    single.resolve_first(0)
    assert outs[0] == outs[1] == single.finalize(0)


class ExchangeRunner:
    """CPU stand-in (TEST ONLY) for a plan with a sharded Partition: GROUP BY key, SUM(v) over keys
    0..K-1.  Mirrors the engine's contract: rows grouped by destination rank in row order, columns
    [key, v, validity word]; rank r owns keys [r*K/world, (r+1)*K/world)."""

    K = 97

    def __init__(self, lo, hi, n, fail=False):
        rng = np.random.default_rng(99)
        self.key = rng.integers(0, self.K, size=n)[lo:hi].astype(np.int64)
        self.v = rng.integers(-9, 9, size=n)[lo:hi].astype(np.int64)
        self.fail = fail

    def exchange_columns(self, sharded_table=None):
        return 3

    def exchange_begin(self, world):
        if self.fail:
            raise m.VdlError(_lib.ERR_UNSUPPORTED if hasattr(_lib, "ERR_UNSUPPORTED") else 3, "1 row(s) carry a partition key outside the pivots")
        self.dest = self.key * world // self.K
        self.order = np.argsort(self.dest, kind="stable")
        return [int((self.dest == r).sum()) for r in range(world)]

    def exchange_pack(self, ptr):
        o = self.order
        self.send[0] = torch.from_numpy(self.key[o])
        self.send[1] = torch.from_numpy(self.v[o])
        self.send[2] = 1

    def exchange_finish(self, ptr, n_recv, as_numpy=False):
        recv = self.recv.numpy()
        assert recv.shape[1] == n_recv
        keys = np.unique(recv[0])
        return {"keys": [int(k) for k in keys], "sums": [int(recv[1][recv[0] == k].sum()) for k in keys]}


def _exchange_worker(rank, world, port, n, q, fail_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = m.shard_rows(n, rank, world)
    runner = ExchangeRunner(lo, hi, n, fail=rank == fail_rank)
    # the stand-in has no device memory: hand it the tensors run_exchange allocates
    import mplan2vdl_amd.sharded as sh
    real_rows = sh.exchange_rows

    def spy(send, counts, dist_, group=None):
        runner.recv = real_rows(send, counts, dist_, group)
        return runner.recv

    sh.exchange_rows = spy
    real_empty = torch.empty

    def empty_spy(*a, **k):
        t = real_empty(*a, **k)
        if t.dim() == 2:
            runner.send = t
        return t

    torch.empty = empty_spy
    try:
        out = m.run_exchange(runner, dist, device="cpu")
    except Exception as exc:
        out = "error: %s" % exc
    finally:
        torch.empty = real_empty
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_two_rank_gloo_partition_exchange(fail_rank):
    n = 5003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000) + (7 if fail_rank >= 0 else 0)
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, n, q, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank >= 0:      # both ranks stop before the first data collective, nobody hangs
        assert "outside the pivots" in outs[1] and "another rank" in outs[0]
        return
    whole = ExchangeRunner(0, n, n)
    keys = np.unique(whole.key)
    assert outs[0]["keys"] + outs[1]["keys"] == [int(k) for k in keys]
    assert outs[0]["sums"] + outs[1]["sums"] == [int(whole.v[whole.key == k].sum()) for k in keys]
    assert max(outs[0]["keys"]) < min(outs[1]["keys"])


# ---- the collective formulation behind the C ABI (vdl_comm.cpp): ONE all-gather of the partial words + a per-word merge ----
def _grouped_partials(lo, hi, groups=6):
    """Stand-in (TEST ONLY) for the partial words of a grouped scan over rows [lo, hi): per group {count SUM, sum SUM,
    max MAX, min MIN, FoldChoose FIRST}; returns (words, words-with-FIRST-resolved) as the local phase leaves them."""
    rng = np.random.default_rng(12345)
    n_all = 5000
    key = rng.integers(0, groups, n_all)
    val = rng.integers(-1000, 1000, n_all)
    key[:37] = 0                                          # group 5 may be missing from a shard; group 0 starts early
    words, resolved = [], []
    for g in range(groups):
        rows = np.nonzero(key[lo:hi] == g)[0] + lo
        v = val[rows]
        first = int(rows[0]) if len(rows) else np.iinfo(np.int64).max
        w = [len(rows), int(v.sum()) if len(rows) else 0, int(v.max()) if len(rows) else np.iinfo(np.int64).min,
             int(v.min()) if len(rows) else np.iinfo(np.int64).max, first]
        words += w
        resolved += w[:4] + [int(val[first]) if len(rows) else 0]
    return np.array(words, np.int64), np.array(resolved, np.int64)


GROUP_OPS = [_lib.REDUCE_SUM, _lib.REDUCE_SUM, _lib.REDUCE_MAX, _lib.REDUCE_MIN, _lib.REDUCE_FIRST]


def _merge_host(world, ops, gathered, with_status=False):
    """gathered: per rank the words raw, the words resolved and -- with_status: the block layout vdl_run_sharded gathers -- ONE status word"""
    import ctypes

    L = _lib.load()
    nw = len(ops)
    out = np.zeros(nw, np.int64)
    status = np.zeros(2, np.int64)
    c_ops = (ctypes.c_int32 * nw)(*ops)
    g = np.ascontiguousarray(gathered, np.int64)
    assert len(g) == world * (2 * nw + (1 if with_status else 0))
    i64p = ctypes.POINTER(ctypes.c_int64)
    assert L.vdl_comm_merge_host(world, nw, c_ops, g.ctypes.data_as(i64p), 2 * nw + 1 if with_status else 0, out.ctypes.data_as(i64p),
                                 status.ctypes.data_as(i64p) if with_status else None) == 0
    return (out, status.tolist()) if with_status else out


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = m.shard_rows(5000, rank, world)
    words, resolved = _grouped_partials(lo, hi)
    send = torch.from_numpy(np.concatenate([words, resolved, [0]]))       # the block vdl_run_sharded sends: raw, resolved, status of the local phase
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)                           # the one collective of the fold route
    merged, status = _merge_host(world, GROUP_OPS * 6, torch.cat(recv).numpy(), with_status=True)
    assert status == [0, -1]
    q.put((rank, merged.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_gathered_partial_words_merge_to_the_single_rank_table(world):
    """World-2 / world-3 gloo ranks all-gather {words, resolved words} once and apply the C merge rule
    (vdl_comm_merge_host = what k_merge_words runs on the GPU): SUM / MIN / MAX per word, FoldChoose = the value of the rank
    holding the smallest global row id -- equal to the table of the whole row range on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    _, whole = _grouped_partials(0, 5000)                  # one rank: FIRST words resolved = the values
    for r in range(world):
        assert outs[r] == whole.tolist()


def test_merge_rule_edge_cases():
    big, small = np.iinfo(np.int64).max, np.iinfo(np.int64).min
    ops = [_lib.REDUCE_SUM, _lib.REDUCE_MIN, _lib.REDUCE_MAX, _lib.REDUCE_FIRST]
    # no rank has a row in the group: FIRST stays "no value" (0), MIN / MAX keep their identities, SUM wraps like the engine's
    g = np.array([big, big, small, big, big, big, small, 0] * 3, np.int64)
    assert _merge_host(3, ops, g).tolist() == [np.int64(3 * np.uint64(big)).item() if False else int((3 * big + 2 ** 63) % 2 ** 64 - 2 ** 63), big, small, 0]
    # one rank: the resolved half is the answer
    assert _merge_host(1, ops, np.array([1, 2, 3, 40, 1, 2, 3, 77], np.int64)).tolist() == [1, 2, 3, 77]
    # the blocks vdl_run_sharded gathers end with the status of the rank's local phase: the first failing rank is handed back
    blocks = np.array([1, 2, 3, 40, 1, 2, 3, 77, 0,   5, 1, 9, 41, 5, 1, 9, 88, -7,   2, 0, 4, 39, 2, 0, 4, 99, -3], np.int64)
    merged, status = _merge_host(3, ops, blocks, with_status=True)
    assert merged.tolist() == [8, 0, 9, 99] and status == [-7, 1]
