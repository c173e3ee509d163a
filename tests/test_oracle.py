"""The CPU oracle against (a) committed golden vectors, (b) independent SQL-semantics evaluators,
(c) hand-computed operator cases that fix the EPS-slot vector model (DESIGN.md "Semantics")."""
import json

import numpy as np
import pytest

import oracle
from conftest import golden
from mplan2vdl_amd import datagen
from helpers import lineitem, oracle_run, prog


def test_oracle_reproduces_golden_q6(q6_text):
    g = json.loads(golden("q6_sf001.json"))
    cols = lineitem(datagen.Q6_COLUMNS, g["rows"], seed=g["seed"])
    assert oracle_run(q6_text, cols) == g["results"]


def test_oracle_reproduces_golden_q1(q1_text):
    g = json.loads(golden("q1_sf001.json"))
    cols = lineitem(datagen.Q1_COLUMNS, g["rows"], seed=g["seed"])
    assert oracle_run(q1_text, cols) == g["results"]


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_interpreter_equals_sql_evaluators(q6_text, q1_text, seed):
    n = 30011
    cols = lineitem(datagen.Q1_COLUMNS, n, seed=seed)
    rev, cnt = oracle.sql_q6(*[cols[c] for c in datagen.Q6_COLUMNS])
    assert oracle_run(q6_text, cols) == {"tmp42": {".revenue": [rev]}}
    sd, di, qt, ep = [cols[c] for c in datagen.Q6_COLUMNS]
    m = (sd >= 728294) & (sd < 728659) & (di >= 5) & (di <= 7) & (qt < 2400)       # third evaluation: numpy
    assert (int((ep[m] * di[m]).sum()), int(m.sum())) == (rev, cnt)
    q1 = oracle_run(q1_text, cols)
    tab = oracle.sql_q1(cols["lineitem.l_shipdate"], cols["lineitem.l_returnflag"], cols["lineitem.l_linestatus"],
                        cols["lineitem.l_quantity"], cols["lineitem.l_extendedprice"], cols["lineitem.l_discount"],
                        cols["lineitem.l_tax"])
    assert q1["tmp101"][".count_order"] == [int(x) for x in tab[:, 9]]
    assert q1["tmp82"][".sum_charge"] == [int(x) for x in tab[:, 5]]
    assert q1["tmp40"][".l_returnflag__lineitem__l_returnflag"] == [int(x) for x in tab[:, 0]]


def test_generated_q6_equals_materialised_q6():
    n = 100000
    cols = lineitem(datagen.Q6_COLUMNS, n)
    specs = [(datagen.SEED, datagen.col_id(c), datagen.LINEITEM[c].lo, datagen.LINEITEM[c].hi, datagen.LINEITEM[c].mul,
              datagen.LINEITEM[c].add) for c in datagen.Q6_COLUMNS]
    want = oracle.sql_q6(*[cols[c] for c in datagen.Q6_COLUMNS])
    assert oracle.sql_q6_generated(specs, 0, n, 1) == want
    assert oracle.sql_q6_generated(specs, 0, n, 4) == want
    a = oracle.sql_q6_generated(specs, 0, 40000, 2)
    b = oracle.sql_q6_generated(specs, 40000, 60000, 2)
    assert (a[0] + b[0], a[1] + b[1]) == want


A = np.array([5, -3, 0, 7, 7, 2, 9, 9], dtype=np.int64)
B = np.array([1, 0, 1, 1, 0, 1, 0, 1], dtype=np.int32)
HEAD = ["1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.b", "4,Project,val,Id 3,b"]
SEL = ["5,RangeV,val,0,Id 4,1", "6,FoldSelect,val,Id 5,val,Id 4,val"]          # b != 0 -> slots 0,2,3,5,7


def run(lines, cols=None):
    return oracle_run(prog(*lines), cols or {"t.a": A, "t.b": B})


def out(lines, cols=None):
    r = run(lines, cols)
    return [list(v.values())[0] for v in r.values()]


def test_foldselect_positions_and_compaction():
    assert out(HEAD + SEL + ["7,MaterializeCompact,Id 6"]) == [[0, 2, 3, 5, 7]]


def test_gather_keeps_holes_until_materialize():
    assert out(HEAD + SEL + ["7,Gather,Id 2,Id 6,val", "8,MaterializeCompact,Id 7"]) == [[5, 0, 7, 2, 9]]


def test_rangev_inherits_eps_so_count_star_counts_selected_rows():
    # count(*) lowering: FoldSum(zeros_ refv, ones_ refv), Vlite.hs:636-639,1044-1046,982-983
    lines = HEAD + SEL + ["7,Gather,Id 2,Id 6,val", "8,RangeV,val,0,Id 7,0", "9,RangeV,val,1,Id 7,0",
                          "10,FoldSum,val,Id 8,val,Id 9,val", "11,MaterializeCompact,Id 10"]
    assert out(lines) == [[5]]


def test_global_fold_result_sits_in_first_slot_and_skips_eps():
    lines = HEAD + SEL + ["7,Gather,Id 2,Id 6,val", "8,RangeV,val,0,Id 7,0",
                          "9,FoldSum,val,Id 8,val,Id 7,val", "10,MaterializeCompact,Id 9",
                          "11,FoldMin,val,Id 8,val,Id 7,val", "12,MaterializeCompact,Id 11",
                          "13,FoldMax,val,Id 8,val,Id 7,val", "14,MaterializeCompact,Id 13",
                          "15,FoldChoose,val,Id 8,val,Id 7,val", "16,MaterializeCompact,Id 15"]
    assert out(lines) == [[23], [0], [9], [5]]


def test_ungrouped_aggregate_over_a_filter_that_drops_row_0_is_read_back_at_position_0():
    """The compiler broadcasts a one-row relation with Gather(result, zeros_ other) (Vlite.hs:693-712; Q11's HAVING
    threshold): the result of an ungrouped aggregate must sit at slot 0 even when the filter dropped row 0 -- the first
    run of a vector starts at slot 0, EPS control slots ahead of its first member belong to it."""
    b = np.array([0, 1, 0, 1, 1, 0, 0, 1], dtype=np.int64)                   # filter keeps slots 1, 3, 4, 7: row 0 is gone
    lines = HEAD + SEL + ["7,Gather,Id 2,Id 6,val", "8,RangeV,val,0,Id 7,0", "9,FoldSum,val,Id 8,val,Id 7,val",
                          "10,RangeV,val,0,Id 2,0",                            # positions 0,0,0,... over the unfiltered table
                          "11,Gather,Id 9,Id 10,val", "12,MaterializeCompact,Id 11",
                          "13,Greater,val,Id 2,val,Id 11,val", "14,MaterializeCompact,Id 13"]
    total = int(A[b != 0].sum())
    assert out(lines, {"t.a": A, "t.b": b}) == [[total] * 8, [int(x > total) for x in A]]
    # a second run keeps its own first member slot; only the FIRST run is pulled to slot 0
    ctl = np.array([4, 4, 4, 6, 6, 6, 6, 6], dtype=np.int64)
    lines = HEAD + SEL + ["7,Gather,Id 2,Id 6,val", "20,Load,t.k", "21,Project,val,Id 20,k", "22,Gather,Id 21,Id 6,val",
                          "23,FoldSum,val,Id 22,val,Id 7,val", "24,RangeV,val,0,Id 23,1", "25,FoldSelect,val,Id 24,val,Id 24,val",
                          "26,MaterializeCompact,Id 23"]
    assert out(lines, {"t.a": A, "t.b": b, "t.k": ctl}) == [[int(A[1]), int(A[3] + A[4] + A[7])]]


def test_fold_runs_follow_control_values():
    ctl = np.array([1, 1, 2, 2, 2, 1, 3, 3], dtype=np.int64)
    lines = ["1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.a", "4,Project,val,Id 3,a",
             "5,FoldSum,val,Id 2,val,Id 4,val", "6,MaterializeCompact,Id 5",
             "7,FoldCount,val,Id 2,val,Id 4,val", "8,MaterializeCompact,Id 7"]
    assert out(lines, {"t.k": ctl, "t.a": A}) == [[2, 14, 2, 18], [2, 3, 1, 2]]


def test_scatter_partition_group_by_pattern():
    # Partition + Scatter + Fold = group by (Vlite.hs:1056-1060,1082-1098)
    key = np.array([2, 0, 1, 2, 0, 1, 2, 2], dtype=np.int64)
    lines = ["1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.a", "4,Project,val,Id 3,a",
             "5,RangeC,val,0,3,1", "6,Partition,val,Id 2,val,Id 5,val", "7,MaterializeCompact,Id 6",
             "8,RangeV,val,0,Id 2,1", "9,Scatter,Id 2,Id 8,val,Id 6,val", "10,MaterializeCompact,Id 9",
             "11,RangeV,val,0,Id 4,1", "12,Scatter,Id 4,Id 11,val,Id 6,val",
             "13,FoldSum,val,Id 9,val,Id 12,val", "14,MaterializeCompact,Id 13"]
    r = out(lines, {"t.k": key, "t.a": A})
    assert r[0] == [4, 0, 2, 5, 1, 3, 6, 7]               # stable counting-sort destinations
    assert r[1] == [0, 0, 1, 1, 2, 2, 2, 2]
    assert r[2] == [-3 + 7, 0 + 2, 5 + 7 + 9 + 9]


def test_partition_skips_eps_rows():
    key = np.array([1, 0, 1, 0, 1, 0, 1, 0], dtype=np.int64)
    lines = ["1,Load,t.k", "2,Project,val,Id 1,k", "3,Load,t.b", "4,Project,val,Id 3,b",
             "5,RangeV,val,0,Id 4,1", "6,FoldSelect,val,Id 5,val,Id 4,val", "7,Gather,Id 2,Id 6,val",
             "8,RangeC,val,0,2,1", "9,Partition,val,Id 7,val,Id 8,val", "10,MaterializeCompact,Id 9"]
    assert out(lines, {"t.k": key, "t.b": B}) == [[3, 4, 0, 1, 2]]       # selected rows 0,2,3,5,7 carry keys 1,1,0,0,0


def test_arithmetic_edge_cases():
    a = np.array([7, -7, 7, -2**63, 5, 1, -8, 3], dtype=np.int64)
    b = np.array([2, 2, 0, -1, -1, -3, 1, 64], dtype=np.int64)
    def one(op):
        return out(["1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.b", "4,Project,val,Id 3,b",
                    "5,%s,val,Id 2,val,Id 4,val" % op, "6,MaterializeCompact,Id 5"], {"t.a": a, "t.b": b})[0]
    assert one("Divide") == [3, -3, 0, -2**63, -5, 0, -8, 0]            # C truncation, x/0 := 0, wrap
    assert one("Modulo") == [1, -1, 0, 0, 0, 1, 0, 3]
    assert one("BitShift") == [1, -2, 7, 0, 10, 8, -4, 0]               # b<0 shifts left (Vlite.hs:205-208)
    assert one("Greater") == [1, 0, 1, 0, 1, 1, 0, 0]
    assert one("LogicalAnd") == [1, 1, 0, 1, 1, 1, 1, 1]


def test_metadata_suffix_and_blank_lines_are_ignored():
    lines = ["1,Load,t.a ;; Metadata {databounds = (0,1)}", "", "2,Project,val,Id 1,a", "3,MaterializeCompact,Id 2"]
    assert out(lines) == [list(A)]


@pytest.mark.parametrize("bad", [
    ["1,Load,t.missing", "2,MaterializeCompact,Id 1"],
    ["1,Load,t.a", "2,Project,val,Id 1,wrongfield"],
    ["1,Load,t.a", "2,Frobnicate,val,Id 1,val,Id 1,val"],
    ["1,Load,t.a", "2,Project,val,Id 7,a"],
    ["1,Load,t.a", "1,Load,t.b"],
])
def test_errors(bad):
    with pytest.raises(oracle.OracleError):
        run(bad)


@pytest.mark.parametrize("n_orders", [10, 1500, 20000])
def test_q3_interpreter_equals_sql_evaluation(n_orders):
    """Joins through join-index Gather/Scatter + sparse Partition (2^38 domain): the machine-generated
    Q3 program under the oracle equals Q3 evaluated from its SQL text with numpy."""
    from helpers import sql_q3

    t = datagen.q3_tables(n_orders)
    got = oracle_run(golden("q3.vdl"), t)
    flat = {list(v.keys())[0][1:]: list(v.values())[0] for v in got.values()}
    assert flat == sql_q3(t)


LIKE_WORDS = ["PROMO BRUSHED TIN", "STANDARD POLISHED COPPER", "MEDIUM POLISHED STEEL", "PROMO", "", "a", "ab", "aab", "a%b_c",
              "special requests", "the special packages wake requests", "forest green", "green", "Customer Complaints",
              "Customer xx Complaints yy", "abcabcabd", "x,y,z"]
LIKE_PATTERNS = ["PROMO%", "%special%requests%", "%green%", "forest%", "%Customer%Complaints%", "MEDIUM POLISHED%", "%", "", "_", "a_",
                 "%b", "a%", "%abcabd", "%abd%", "_%_", "%%a%%", "PROMO", "x,y,%", "%,z", "a%b_c", "a_b_c"]


def like_program(pattern):
    return prog("1,Load,t.s", "2,Project,val,Id 1,s", "3,Load,t.s.heap", "4,Project,val,Id 3,s.heap",
                "5,Like,val,Id 2,val,Id 4,val," + pattern, "6,MaterializeCompact,Id 5")


@pytest.mark.parametrize("pattern", LIKE_PATTERNS)
def test_like_matches_regex_statement_of_sql_like(pattern):
    """Like over a string heap (Vdl.hs:444-447): the oracle against a regular-expression restatement."""
    from helpers import make_heap, sql_like

    heap, where = make_heap(LIKE_WORDS)
    rng = np.random.default_rng(3)
    words = [LIKE_WORDS[k] for k in rng.integers(0, len(LIKE_WORDS), 200)]
    cols = {"t.s": np.array([where[w] for w in words], dtype=np.int64), "t.s.heap": heap}
    got = oracle_run(like_program(pattern), cols)
    assert list(got.values())[0][".val"] == [sql_like(w, pattern) for w in words]


def test_like_offsets_outside_the_heap_and_holes():
    from helpers import make_heap

    heap, where = make_heap(["abc", "abd"])
    cols = {"t.s": np.array([where["abc"], -5, len(heap), len(heap) + 7, where["abd"], where["abc"] + 1], dtype=np.int64),
            "t.s.heap": heap, "t.f": np.array([1, 1, 1, 1, 0, 1], dtype=np.int64)}
    text = prog("1,Load,t.s", "2,Project,val,Id 1,s", "3,Load,t.s.heap", "4,Project,val,Id 3,s.heap",
                "5,Load,t.f", "6,Project,val,Id 5,f", "7,RangeV,val,0,Id 6,1", "8,FoldSelect,val,Id 7,val,Id 6,val",
                "9,Gather,Id 2,Id 8,val",                                   # row 4 becomes EPS
                "10,Like,val,Id 9,val,Id 4,val,%c", "11,MaterializeCompact,Id 10")
    got = oracle_run(text, cols)
    assert list(got.values())[0][".val"] == [1, 0, 0, 0, 1]               # "abc", out of heap x3, (hole dropped), "bc"


def test_semisort_gathers_equal_values_together():
    """Semisort (Vdl.hs:42; Vlite.hs:109-111 "a permutation such that when the input is gathered with it, the output
    has all instances of an equal value be contiguous"): stable ascending order of the non-EPS slots, VLite dialect."""
    cols = {"t.k": np.array([5, 3, 5, -2, 3, 9, 5], dtype=np.int64), "t.f": np.array([1, 1, 1, 1, 0, 1, 1], dtype=np.int64)}
    text = prog("1,Load,t.k", "2,Project,Id 1", "3,Load,t.f", "4,Project,Id 3", "5,RangeV,0,Id 4,1", "6,FoldSelect,Id 5,Id 4",
                "7,Gather,Id 2,Id 6",                    # slot 4 becomes EPS
                "8,Semisort,Id 7", "9,Output,Id 8", "10,Gather,Id 7,Id 8", "sorted,Output,decimal_0,Id 10",
                "12,RangeV,1,Id 10,0", "13,FoldSum,Id 10,Id 12", "counts,Output,decimal_0,Id 13")
    got = oracle_run(text, cols)
    assert got == {"tmp9": {".val": [3, 1, 0, 2, 6, 5]}, "tmp11": {".sorted": [-2, 3, 5, 5, 5, 9]}, "tmp14": {".counts": [1, 1, 3, 1]}}
