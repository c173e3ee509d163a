"""Row-range sharded execution: one process per GPU, RCCL all-reduce of the fold partials.

The reference has no distribution at all; this is the MI355X design of
SURVEY.md section 8(e): rank g owns rows [g*n/G, (g+1)*n/G), runs the fused scan over them,
and the per-scan partial words {selected-row count, aggregates...} are merged with
all-reduce (SUM / MIN / MAX) -- 8 to a few hundred bytes, i.e. latency-bound on xGMI --
before every rank finalises the same answer.
"""
from . import _lib


def shard_rows(n_rows, rank, world):
    """Contiguous row range [lo, hi) of `rank`; sizes differ by at most one row."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def merge_partials(buf, ops, dist, group=None, resolve_first=None):
    """All-reduce the words of `buf` (1-D int64 tensor) according to their VDL_REDUCE_* tags.

    FIRST words (FoldChoose of a grouped plan) take two rounds: MIN over the global row ids,
    `resolve_first()` (owner substitutes the value, others 0), then SUM."""
    import torch

    table = {_lib.REDUCE_SUM: dist.ReduceOp.SUM, _lib.REDUCE_MIN: dist.ReduceOp.MIN, _lib.REDUCE_MAX: dist.ReduceOp.MAX,
             _lib.REDUCE_FIRST: dist.ReduceOp.MIN}
    classes = sorted(set(ops))
    if len(classes) == 1 and classes[0] != _lib.REDUCE_FIRST:
        dist.all_reduce(buf, op=table[classes[0]], group=group)
        return buf
    index = {}
    for cls in classes:
        idx = torch.tensor([i for i, o in enumerate(ops) if o == cls], dtype=torch.long, device=buf.device)
        index[cls] = idx
        part = buf.index_select(0, idx)
        dist.all_reduce(part, op=table[cls], group=group)
        buf.index_copy_(0, idx, part)
    if _lib.REDUCE_FIRST in index:
        if resolve_first is None:
            raise ValueError("plan has FoldChoose words: pass resolve_first")
        resolve_first()
        idx = index[_lib.REDUCE_FIRST]
        part = buf.index_select(0, idx)
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
        buf.index_copy_(0, idx, part)
    return buf


class ShardedQuery:
    """Drives local phase -> merge -> finalise for a plan whose outputs are global folds.

    `runner` provides partial_spec() -> (n_words, ops), run_local(dev_ptr), finalize(dev_ptr)
    (mplan2vdl_amd.engine.Plan does); `buf` is a 1-D int64 tensor on the runner's device.
    """

    def __init__(self, runner, buf, dist=None, group=None):
        self.runner = runner
        self.buf = buf
        self.dist = dist
        self.group = group
        self.n_words, self.ops = runner.partial_spec()
        if buf.numel() < self.n_words:
            raise ValueError("partials buffer too small")
        # The collectives run behind torch's current stream; the engine must launch on that same stream, or the
        # all-reduce could read the partial words before the scan wrote them (and finalisation copy them before the
        # reduction landed).  Made part of the API instead of a thing callers have to remember.
        eng = getattr(runner, "_e", None)
        if eng is not None and getattr(buf, "is_cuda", False):
            eng.use_torch_stream()

    def _merge(self, buf):
        has_first = _lib.REDUCE_FIRST in self.ops
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1:
            merge_partials(buf[: self.n_words], self.ops, self.dist, self.group,
                           (lambda: self.runner.resolve_first(buf.data_ptr())) if has_first else None)
        elif has_first:
            self.runner.resolve_first(buf.data_ptr())

    def step(self):
        self.runner.run_local(self.buf.data_ptr())
        self._merge(self.buf)
        return self.runner.finalize(self.buf.data_ptr())

    def _merge_is_one_collective(self):
        classes = set(self.ops)
        return (self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1
                and len(classes) == 1 and _lib.REDUCE_FIRST not in classes)

    def run_pipelined(self, steps, bufs, on_result=None, overlap_merge=True):
        """`steps` queries back to back with the host side of query k overlapped with the kernels of
        query k+1 (two partial buffers / finalisation slots).  Every query still runs in full and
        every result is produced; returns the last one.

        When the merge is a single all-reduce (Q6: two SUM words) and `overlap_merge` is set, it is issued
        asynchronously: RCCL runs it on its own stream behind the scan of query k while the engine's stream
        goes on with the scan of query k+1, and the stream only waits for it just before query k is
        finalised -- the collective's latency (what bounds 8-GPU scaling) is hidden."""
        if len(bufs) != 2:
            raise ValueError("run_pipelined needs two partial buffers")
        out = None

        def emit(slot):
            res = self.runner.finalize_end(slot)
            if on_result is not None:
                on_result(res)
            return res

        if overlap_merge and self._merge_is_one_collective():
            op = {_lib.REDUCE_SUM: self.dist.ReduceOp.SUM, _lib.REDUCE_MIN: self.dist.ReduceOp.MIN,
                  _lib.REDUCE_MAX: self.dist.ReduceOp.MAX}[self.ops[0]]
            work = [None, None]
            for k in range(steps):
                s = k & 1
                self.runner.run_local(bufs[s].data_ptr())
                work[s] = self.dist.all_reduce(bufs[s][: self.n_words], op=op, group=self.group, async_op=True)
                if k >= 1:
                    work[1 - s].wait()                       # stream-side wait; query k's scan is already queued
                    self.runner.finalize_begin(bufs[1 - s].data_ptr(), 1 - s)
                if k >= 2:
                    out = emit(s)                            # query k-2
            if steps >= 2:
                out = emit(steps & 1)                        # query steps-2
            if steps >= 1:
                last = (steps - 1) & 1
                work[last].wait()
                self.runner.finalize_begin(bufs[last].data_ptr(), last)
                out = emit(last)
            return out

        for k in range(steps):
            s = k & 1
            self.runner.run_local(bufs[s].data_ptr())
            self._merge(bufs[s])
            self.runner.finalize_begin(bufs[s].data_ptr(), s)
            if k > 0:
                out = emit(1 - s)
        if steps > 0:
            out = emit((steps - 1) & 1)
        return out


def exchange_rows(send, counts, dist, group=None):
    """All-to-all of the packed rows: `send` is (n_columns, n_send) int64 grouped by destination rank,
    counts[r] rows go to rank r.  Returns (n_columns, n_recv), the pieces in source-rank order."""
    import torch

    if send.is_cuda and dist.get_backend(group) == "gloo":       # rehearsal on one GPU: gloo has no device all-to-all
        return exchange_rows(send.cpu(), counts, dist, group).to(send.device)
    cnt_in = torch.tensor(counts, dtype=torch.int64, device=send.device)
    cnt_out = torch.empty_like(cnt_in)
    dist.all_to_all_single(cnt_out, cnt_in, group=group)
    recv_counts = [int(x) for x in cnt_out.tolist()]
    n_recv = sum(recv_counts)
    ncols = send.shape[0]
    recv = torch.empty((ncols, max(n_recv, 1)), dtype=torch.int64, device=send.device)[:, :n_recv].contiguous()
    for c in range(ncols):
        dist.all_to_all_single(recv[c], send[c], output_split_sizes=recv_counts, input_split_sizes=list(counts), group=group)
    return recv


def run_exchange(plan, dist=None, group=None, device="cuda", sharded_table=None, as_numpy=False):
    """Sharded execution of a plan with a Partition (e.g. TPC-H Q3): local phase, all-to-all of the rows
    by key range over RCCL, local tail.  Returns this rank's slice of the result (the slices of rank 0,
    1, ... concatenate to the unsharded result).  The partitioned table is sharded by rows; dimension
    tables are replicated.  With dist=None (single rank) the exchange degenerates to a local compaction."""
    import torch

    world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
    ncols = plan.exchange_columns(sharded_table)
    failure = None
    try:
        counts = plan.exchange_begin(world)
    except Exception as exc:          # keep the ranks in step: everybody learns about the failure before any collective
        if world == 1:
            raise
        failure, counts = exc, [0] * world
    if world > 1:
        flag = torch.tensor([1 if failure else 0], dtype=torch.int64, device="cpu" if dist.get_backend(group) == "gloo" else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if failure:
            raise failure
        if int(flag.item()):
            raise RuntimeError("sharded Partition exchange failed on another rank")
    n_send = sum(counts)
    send = torch.empty((ncols, max(n_send, 1)), dtype=torch.int64, device=device)[:, :n_send].contiguous()
    plan.exchange_pack(send.data_ptr())
    if world == 1:
        return plan.exchange_finish(send.data_ptr(), n_send, as_numpy)
    recv = exchange_rows(send, counts, dist, group)
    if recv.is_cuda:
        torch.cuda.current_stream(recv.device).synchronize()      # the engine may run on a stream of its own
    return plan.exchange_finish(recv.data_ptr(), recv.shape[1], as_numpy)
