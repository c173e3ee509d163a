"""The GROUP BY tail over keys that arrive in order (vdl_genexec.h: partition_positions / fold_entries, vdl_ops.hip: k_group_fold,
vdl_partition.hip: k_sorted_heads): the Partition's sortedness pass leaves the run heads, every fold of the GROUP BY runs in one
launch with packed results.  Boundary shapes on purpose: vector lengths around the kernel's units (8 entries per lane, 512 per
wave, 2048 per block, 4096 per compaction tile), runs of one entry, runs longer than a wave / a tile, one single run, every fold
kind at once (more folds than one trip of the kernel takes), negative values, constant and row-id data, filtered (entries of a
selection) and unfiltered (all slots) operands -- each against the oracle, with the batch on and off."""
import numpy as np
import pytest

from helpers import check_against_oracle, engine_with, oracle_run, prog

pytestmark = pytest.mark.gpu
SIZES = [1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8193, 100003]
FOLDS = ["FoldSum", "FoldMin", "FoldMax", "FoldCount", "FoldChoose"]


def program(filtered, domain=1 << 40):
    L = ["1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.v", "4,Project,val,Id 3,v", "5,Load,t.w", "6,Project,val,Id 5,w", "7,Load,t.f", "8,Project,val,Id 7,f"]
    nid = [8]

    def emit(body):
        nid[0] += 1
        L.append("%d,%s" % (nid[0], body))
        return nid[0]

    k, v, w = 2, 4, 6
    if filtered:
        zero = emit("RangeV,val,0,Id 8,0")
        pred = emit("Greater,val,Id 8,val,Id %d,val" % zero)
        sel = emit("FoldSelect,val,Id %d,val,Id %d,val" % (emit("RangeV,val,0,Id %d,1" % pred), pred))
        k, v, w = (emit("Gather,Id %d,Id %d,val" % (x, sel)) for x in (k, v, w))
    piv = emit("RangeC,val,0,%d,1" % domain)
    part = emit("Partition,val,Id %d,val,Id %d,val" % (k, piv))
    sk = emit("Scatter,Id %d,Id %d,val,Id %d,val" % (k, emit("RangeV,val,0,Id %d,1" % k), part))
    ones = emit("RangeV,val,1,Id %d,0" % v)
    rowid = emit("RangeV,val,0,Id %d,1" % v)
    vw = emit("Multiply,val,Id %d,val,Id %d,val" % (v, emit("Subtract,val,Id %d,val,Id %d,val" % (w, ones))))
    outs = []
    for data in (v, w, vw, ones, rowid):
        sd = emit("Scatter,Id %d,Id %d,val,Id %d,val" % (data, emit("RangeV,val,0,Id %d,1" % data), part))
        for f in (FOLDS if data in (v, vw) else ["FoldSum", "FoldChoose"]):
            outs.append(emit("%s,val,Id %d,val,Id %d,val" % (f, sk, sd)))
    outs.append(emit("FoldChoose,val,Id %d,val,Id %d,val" % (sk, sk)))
    for o in outs:
        emit("MaterializeCompact,Id %d" % o)
    return prog(*L)


def columns(n, seed, shape):
    r = np.random.default_rng(seed)
    if shape == "short":            # runs of 1 .. 7
        k = np.cumsum(r.integers(0, 2, n) | (r.integers(0, 7, n) == 0))
    elif shape == "mixed":          # runs of 1, runs longer than a wave, longer than a tile
        lens, total = [], 0
        while total < n:
            ln = int(r.choice([1, 1, 2, 3, 70, 600, 5000, 9000]))
            lens.append(ln); total += ln
        k = np.repeat(np.arange(len(lens)) * 3, lens)[:n]
    else:                           # one run
        k = np.zeros(n, np.int64)
    return {"t.k": (k.astype(np.int64) + 5), "t.v": r.integers(-1000, 1000, n).astype(np.int64), "t.w": r.integers(-3, 50, n).astype(np.int32),
            "t.f": (r.integers(0, 10, n) < 7).astype(np.int64)}


@pytest.mark.parametrize("batch", [True, False])
@pytest.mark.parametrize("filtered", [True, False])
def test_folds_over_keys_in_order_match_the_oracle(filtered, batch, monkeypatch):
    if not batch:
        monkeypatch.setenv("VDL_NO_GROUP_BATCH", "1")
    text = program(filtered)
    for n in SIZES:
        for shape in ("short", "mixed", "one"):
            cols = columns(n, n * 7 + len(shape), shape)
            want = oracle_run(text, cols)
            e = engine_with(cols)
            got = e.run_vdl(text)["results"]
            e.close()
            check_against_oracle("group_tail_%s_%s_%s" % ("filtered" if filtered else "all", shape, "batch" if batch else "nobatch"), n, text, cols, got, want)


@pytest.mark.parametrize("domain", [1 << 40, 1 << 12])
def test_folds_over_keys_out_of_order_take_the_radix_route(domain):
    """the same programs over keys in random order: the sortedness pass says no, the Partition sorts (its positions stay in rank order:
    only Scatters read them; the key's own Scatter is what the sort wrote), the folds run over the scattered vectors.  With the small
    domain some keys lie beyond the last pivot and some below the first: buckets are clamped, so the sorted buckets are NOT the
    sorted keys and the key is gathered like every other vector."""
    for filtered in (True, False):
        text = program(filtered, domain)
        for n in (9, 513, 4097, 50021):
            cols = columns(n, n, "mixed")
            cols["t.k"] = np.random.default_rng(n).permutation(cols["t.k"])
            if domain < (1 << 20):
                cols["t.k"] = cols["t.k"] - 40                   # a few below the first pivot; the long tail beyond the last
            want = oracle_run(text, cols)
            e = engine_with(cols)
            got = e.run_vdl(text)["results"]
            e.close()
            check_against_oracle("group_tail_unsorted", n, text, cols, got, want)


@pytest.mark.parametrize("shape", ["packed", "pairs", "holes", "clamped", "narrow"])
def test_partition_over_many_tiles_matches_a_stable_numpy_sort(shape):
    """The one-sweep radix Partition at a size where its look-back matters (vdl_partition.hip: k_part_pass): ~6 M slots = 730 tiles,
    i.e. Fenwick nodes up to level 9 and thousands of status words polled while their producers run.  Checked against numpy's stable
    argsort of the buckets -- positions written out (the last pass scatters ranks), and the GROUP BY idiom whose Scatters read the
    rank-order list instead (lazy positions, the sorted keys handed over by the last pass).
    packed: 2^40 domain, five passes of (bucket << 23 | slot) words; pairs: a domain too wide for one word (keys and slots travel
    side by side); holes: EPS rows in the key (no sortedness pass, the declared domain's pass count); clamped: values outside the
    pivots on both sides; narrow: a 200-bucket domain (one pass)."""
    n = 5_982_721
    r = np.random.default_rng(len(shape))
    pmin = 0
    if shape == "pairs":
        k, domain = r.integers(0, 1 << 52, n), 1 << 52
    elif shape == "clamped":
        k, domain, pmin = r.integers(-50_000, 400_000, n), 300_000, 1000
    elif shape == "narrow":
        k, domain = r.integers(0, 200, n), 200
    else:
        k, domain = (r.integers(1, 1 << 20, n) * r.integers(1, 1 << 19, n)), 1 << 40
    cols = {"t.k": k.astype(np.int64), "t.v": r.integers(-1000, 1000, n).astype(np.int64), "t.f": (r.integers(0, 10, n) < 7).astype(np.int64)}
    L = ["1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.v", "4,Project,val,Id 3,v", "5,Load,t.f", "6,Project,val,Id 5,f"]
    kk, vv = 2, 4
    keep = np.ones(n, bool)
    if shape == "holes":
        L += ["7,RangeV,val,0,Id 6,0", "8,Greater,val,Id 6,val,Id 7,val", "9,RangeV,val,0,Id 8,1", "10,FoldSelect,val,Id 9,val,Id 8,val",
              "11,Gather,Id 2,Id 10,val", "12,Gather,Id 4,Id 10,val"]
        kk, vv = 11, 12
        keep = cols["t.f"] > 0
    L += ["20,RangeC,val,%d,%d,1" % (pmin, domain), "21,Partition,val,Id %d,val,Id 20,val" % kk,
          "22,RangeV,val,0,Id %d,1" % kk, "23,Scatter,Id %d,Id 22,val,Id 21,val" % kk, "24,Scatter,Id %d,Id 22,val,Id 21,val" % vv,
          "25,FoldSum,val,Id 23,val,Id 24,val", "26,Project,s,Id 25,val", "27,MaterializeCompact,Id 26",
          "28,FoldChoose,val,Id 23,val,Id 23,val", "29,Project,key,Id 28,val", "30,MaterializeCompact,Id 29"]
    lazy = prog(*L)
    eager = prog(*(L + ["31,Project,pos,Id 21,val", "32,MaterializeCompact,Id 31"]))
    kept = cols["t.k"][keep]
    bucket = np.where(kept > pmin, np.minimum(kept - pmin, domain), 0)
    order = np.argsort(bucket, kind="stable")
    sk = kept[order]                                        # the folds' control vector: the key VALUES in bucket order (runs = equal neighbours)
    heads = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    want_s = np.add.reduceat(cols["t.v"][keep][order], heads)
    want_key = sk[heads]
    pos = np.empty(len(order), np.int64)
    pos[order] = np.arange(len(order))
    e = engine_with(cols)
    for text in (lazy, eager):
        p = e.parse(text)
        p.set_fusion(False)
        got = p.run(as_numpy=True)["results"]
        assert np.array_equal(got["tmp27"][".s"], want_s), shape
        assert np.array_equal(got["tmp30"][".key"], want_key), shape
        if text is eager:
            assert np.array_equal(got["tmp32"][".pos"], pos), shape
        p.close()
    e.close()
