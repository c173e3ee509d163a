"""Deterministic programs that pin the bookkeeping of the statement-by-statement executor's vector forms
(mplan2vdl_amd/csrc/vdl_genexec.h): each one is a shape where a shortcut's precondition silently stops holding.
Found by reading the executor, not by a failing run; every case compares all statements with the oracle."""
import numpy as np
import pytest

from helpers import check_against_oracle, compare_traced, engine_with, prog

pytestmark = pytest.mark.gpu

MODES = [None, "VDL_NO_SPARSE", "VDL_SPARSE_ALWAYS"]


def run_all_statements(tag, text, cols):
    import oracle

    orc = oracle.Oracle()
    orc.keep_vectors(True)
    for k, v in cols.items():
        orc.add_column(k, v)
    want = orc.run(text)["results"]
    e = engine_with(cols)
    p = e.parse(text)
    p.set_fusion(False)
    p.set_trace(True)
    got = p.run()["results"]
    first = compare_traced(p, orc, text)
    e.close()
    orc.close()
    assert first is None, "%s: %s" % (tag, first)
    check_against_oracle(tag, 0, text, cols, got, want)
    return got


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [1, 64, 777, 5000])
def test_partition_ranks_read_through_a_second_filter_are_no_longer_a_prefix(monkeypatch, mode, n):
    """Partition positions of a filtered vector hold exactly the ranks 0 .. m-1, so a Scatter by them needs no validity
    atomics (the result's validity is a prefix).  Read through ANOTHER filter (an identity Gather: a view) they are a
    subset of those ranks: the view must drop the `ranks` / `perm` marks or the Scatter claims a prefix it does not write."""
    if mode:
        monkeypatch.setenv(mode, "1")
    rng = np.random.default_rng(n)
    cols = {"t.a": rng.integers(-20, 21, n).astype(np.int64), "t.b": rng.integers(0, 8, n).astype(np.int32)}
    text = prog(
        "1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.b", "4,Project,val,Id 3,b",
        "5,RangeV,val,2,Id 2,0", "6,Greater,val,Id 2,val,Id 5,val",
        "7,RangeV,val,0,Id 6,1", "8,FoldSelect,val,Id 7,val,Id 6,val",
        "9,Gather,Id 4,Id 8,val",
        "10,RangeC,val,0,8,1", "11,Partition,val,Id 9,val,Id 10,val",
        "12,RangeV,val,9,Id 2,0", "13,Greater,val,Id 12,val,Id 2,val",
        "14,RangeV,val,0,Id 13,1", "15,FoldSelect,val,Id 14,val,Id 13,val",
        "16,Gather,Id 11,Id 15,val", "17,Gather,Id 4,Id 15,val",
        "18,RangeV,val,0,Id 17,1", "19,Scatter,Id 17,Id 18,val,Id 16,val",
        "20,MaterializeCompact,Id 19",
        "21,RangeV,val,0,Id 19,0", "22,FoldCount,val,Id 21,val,Id 19,val", "23,MaterializeCompact,Id 22")
    run_all_statements("ranks_through_filter_%s_%d" % (mode, n), text, cols)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [2, 300, 4097])
def test_unscattered_and_scattered_keys_sharing_one_buffer_get_their_own_run_heads(monkeypatch, mode, n):
    """A sparse key that is already in order is "scattered" by its Partition ranks without moving a byte: the scattered
    key shares the entry buffer of the original but lives on the prefix selection.  Folding over both must give results
    on two different selections (the cached run heads are per (entries, selection), not per buffer)."""
    if mode:
        monkeypatch.setenv(mode, "1")
    rng = np.random.default_rng(n + 1)
    cols = {"t.c": np.sort(rng.integers(0, 900, n)).astype(np.int64), "t.a": rng.integers(-20, 21, n).astype(np.int64)}
    text = prog(
        "1,Load,t.c", "2,Project,val,Id 1,c", "3,Load,t.a", "4,Project,val,Id 3,a",
        "5,RangeV,val,0,Id 4,0", "6,Greater,val,Id 4,val,Id 5,val",
        "7,RangeV,val,0,Id 6,1", "8,FoldSelect,val,Id 7,val,Id 6,val",
        "9,Gather,Id 2,Id 8,val", "10,Gather,Id 4,Id 8,val",
        "11,RangeC,val,0,1000,1", "12,Partition,val,Id 9,val,Id 11,val",
        "13,RangeV,val,0,Id 9,1",
        "14,Scatter,Id 9,Id 13,val,Id 12,val", "15,Scatter,Id 10,Id 13,val,Id 12,val",
        "16,FoldSum,val,Id 9,val,Id 10,val", "17,FoldSum,val,Id 14,val,Id 15,val",
        "18,Add,val,Id 16,val,Id 17,val",
        "19,MaterializeCompact,Id 18", "20,MaterializeCompact,Id 16", "21,MaterializeCompact,Id 17")
    run_all_statements("shared_key_buffer_%s_%d" % (mode, n), text, cols)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,first_kept", [(8, 3), (700, 65), (5000, 4999), (5000, 0)])
def test_the_first_run_of_a_fold_starts_at_slot_0(monkeypatch, mode, n, first_kept):
    """An ungrouped aggregate over a filter that drops the leading rows is read back at position 0 (the compiler broadcasts
    one-row relations with Gather(result, zeros_ other), Vlite.hs:693-712; TPC-H Q11), and so is the first run of a fold
    over a general control vector that starts with EPS slots; later runs keep their own first member slot."""
    if mode:
        monkeypatch.setenv(mode, "1")
    rng = np.random.default_rng(n + first_kept)
    keep = (rng.random(n) < 0.3).astype(np.int64)
    keep[:first_kept] = 0
    keep[first_kept] = 1
    cols = {"t.a": rng.integers(-50, 50, n).astype(np.int64), "t.b": keep, "t.k": np.sort(rng.integers(0, 6, n)).astype(np.int64)}
    text = prog(
        "1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.b", "4,Project,val,Id 3,b", "5,Load,t.k", "6,Project,val,Id 5,k",
        "7,RangeV,val,0,Id 4,1", "8,FoldSelect,val,Id 7,val,Id 4,val",
        "9,Gather,Id 2,Id 8,val", "10,Gather,Id 6,Id 8,val",
        "11,RangeV,val,0,Id 9,0",
        "12,FoldSum,val,Id 11,val,Id 9,val", "13,FoldMax,val,Id 11,val,Id 9,val", "14,FoldChoose,val,Id 11,val,Id 9,val",
        "15,RangeV,val,0,Id 2,0", "16,Gather,Id 12,Id 15,val", "17,Greater,val,Id 2,val,Id 16,val",
        "18,MaterializeCompact,Id 16", "19,MaterializeCompact,Id 17",
        "20,Subtract,val,Id 13,val,Id 14,val", "21,MaterializeCompact,Id 20",
        "22,FoldSum,val,Id 10,val,Id 9,val", "23,FoldCount,val,Id 10,val,Id 9,val", "24,FoldChoose,val,Id 10,val,Id 9,val",
        "25,Gather,Id 22,Id 15,val", "26,MaterializeCompact,Id 25",
        "27,MaterializeCompact,Id 22", "28,MaterializeCompact,Id 23", "29,MaterializeCompact,Id 24",
        "30,Add,val,Id 22,val,Id 12,val", "31,MaterializeCompact,Id 30")
    got = run_all_statements("first_run_slot0_%s_%d_%d" % (mode, n, first_kept), text, cols)
    total = int(cols["t.a"][keep != 0].sum())
    assert got["tmp18"][".val"] == [total] * n                       # every row reads the aggregate at position 0
