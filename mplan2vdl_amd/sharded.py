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

    def run_pipelined(self, steps, bufs, on_result=None):
        """`steps` queries back to back with the host side of query k overlapped with the kernels of
        query k+1 (two partial buffers / finalisation slots).  Every query still runs in full and
        every result is produced; returns the last one."""
        if len(bufs) != 2:
            raise ValueError("run_pipelined needs two partial buffers")
        out = None
        for k in range(steps):
            s = k & 1
            self.runner.run_local(bufs[s].data_ptr())
            self._merge(bufs[s])
            self.runner.finalize_begin(bufs[s].data_ptr(), s)
            if k > 0:
                out = self.runner.finalize_end(1 - s)
                if on_result is not None:
                    on_result(out)
        if steps > 0:
            out = self.runner.finalize_end((steps - 1) & 1)
            if on_result is not None:
                on_result(out)
        return out
