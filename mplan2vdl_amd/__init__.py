"""mplan2vdl_amd -- MI355X-native execution engine for the VDL text emitted by orm011/mplan2vdl.

The product is libvdl.so (hand-written HIP kernels behind the C ABI of include/vdl.h); this
package is the thin Python host side: ctypes binding, synthetic TPC-H-shaped data, and the
one-process-per-GPU sharding driver.
"""
from . import datagen  # noqa: F401
from .engine import Engine, Plan, VdlError  # noqa: F401
from .sharded import ShardedQuery, merge_partials, run_exchange, shard_rows  # noqa: F401

__all__ = ["Engine", "Plan", "VdlError", "ShardedQuery", "merge_partials", "run_exchange", "shard_rows", "datagen"]
