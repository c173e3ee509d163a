"""The C-ABI library loads and exports every symbol include/vdl.h declares; host-only contexts can
parse, plan and describe; anything that needs a device fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import mplan2vdl_amd as m
from mplan2vdl_amd import _lib
from conftest import ROOT
from helpers import prog


def header_functions():
    text = open(os.path.join(ROOT, "include", "vdl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdl_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.ABI_SYMBOLS) == names


def test_host_only_context_parses_and_describes_q6(q6_text):
    e = m.Engine(device=None)
    p = e.parse(q6_text)
    assert p.is_fused
    d = p.describe()
    assert "lineitem.l_shipdate in [728294,728658]" in d        # >= 1994-01-01 and < 1995-01-01
    assert "lineitem.l_discount in [5,7]" in d                  # between 0.06-0.01 and 0.06+0.01, constants folded
    assert "lineitem.l_quantity in [-inf,2399]" in d            # < 24.00
    assert "sum" in d and "lineitem.l_extendedprice" in d
    assert p.partial_spec() == (2, [_lib.REDUCE_SUM, _lib.REDUCE_SUM])
    with pytest.raises(m.VdlError) as ei:
        p.run()
    assert ei.value.code == _lib.VDL_ERR_DEVICE                 # loud: no CPU fallback
    p.set_fusion(False)
    assert not p.is_fused and "general: 42 statement(s)" in p.describe()


def test_q1_fuses_into_one_grouped_scan(q1_text):
    e = m.Engine(device=None)
    p = e.parse(q1_text)
    assert p.is_fused
    d = p.describe()
    assert "group-scan 0 table=lineitem buckets=[0,31]" in d                 # RangeC 0 32 1 pivots
    assert "lineitem.l_shipdate in [-inf,729999]" in d
    assert "acc=BitwiseAnd(acc,31)" in d                                     # size hint evaluated, Vlite.hs:1111-1115
    assert "key form: composite, ((col1 >> 3) - 2) << 2 | ((col2 >> 3) - 2) << 0, & 31" in d      # makeCompositeKey's shape: no interpreter
    assert d.count(" sum ") == 5 and d.count(" first ") == 2                 # duplicate FoldSums shared, count(*) = the row-count word
    assert "Divide(agg2,count)" in d                                         # avg = sum / count, Vlite.hs:1038-1041
    nw, ops = p.partial_spec()                                               # 32 buckets x (count + 7 aggregates) + out-of-domain count
    assert nw == 32 * 8 + 1 and ops[:8] == [_lib.REDUCE_SUM, _lib.REDUCE_FIRST, _lib.REDUCE_FIRST] + [_lib.REDUCE_SUM] * 5


def test_unsupported_shapes_stay_on_the_general_path():
    e = m.Engine(device=None)
    p = e.parse(prog("1,Load,t.a", "2,Project,val,Id 1,a", "3,RangeV,val,0,Id 2,1", "4,Gather,Id 2,Id 3,val",
                     "5,MaterializeCompact,Id 4"))
    assert not p.is_fused and "not fused" in p.describe()
    with pytest.raises(m.VdlError) as ei:
        p.partial_spec()
    assert ei.value.code == _lib.VDL_ERR_UNSUPPORTED


@pytest.mark.parametrize("text,code", [
    ("", _lib.VDL_ERR_PARSE),
    ("1,Load\n", _lib.VDL_ERR_PARSE),
    ("x,Load,t.a\n", _lib.VDL_ERR_PARSE),
    ("1,Load,t.a\n2,Project,val,Id 9,a\n", _lib.VDL_ERR_PARSE),
    ("1,Load,t.a\n1,Load,t.b\n", _lib.VDL_ERR_PARSE),
    ("1,Load,t.a\n2,Project,val,Id 1,b\n", _lib.VDL_ERR_SHAPE),
    ("1,Load,t.a\n2,Greater,val,Id 1,val,Id 1,val\n", _lib.VDL_ERR_SHAPE),
    ("1,Load,t.a\n2,Frob,val,Id 1,a,Id 1,a\n", _lib.VDL_ERR_PARSE),
    ("1,Load,t.a\n2,Project,val,Id 1,a\n3,Like,val,Id 2,val,Id 2,val," + "x" * 300 + "\n", _lib.VDL_ERR_UNSUPPORTED),
])
def test_parse_errors_carry_codes(text, code):
    e = m.Engine(device=None)
    with pytest.raises(m.VdlError) as ei:
        e.parse(text)
    assert ei.value.code == code


def test_metadata_suffix_is_ignored():
    e = m.Engine(device=None)
    p = e.parse("1,Load,t.a ;; Metadata {databounds = (0,1), name = Just t.a}\n\n2,Project,val,Id 1,a\n3,MaterializeCompact,Id 2\n")
    assert "general: 3 statement(s)" in p.describe()


def test_opening_a_device_without_gpu_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(m.VdlError) as ei:
        m.Engine(device=0)
    assert ei.value.code == _lib.VDL_ERR_DEVICE


HEAD = ["1,Load,t.a", "2,Project,val,Id 1,a", "3,Load,t.b", "4,Project,val,Id 3,b", "5,Load,t.c", "6,Project,val,Id 5,c"]


def plan_of(lines):
    return m.Engine(device=None).parse(prog(*(HEAD + lines)))


def test_fusion_of_range_predicates_and_affine_products():
    p = plan_of(["7,RangeV,val,10,Id 2,0", "8,Greater,val,Id 2,val,Id 7,val",            # a > 10
                 "9,RangeV,val,3,Id 2,0", "10,Equals,val,Id 4,val,Id 9,val",               # b == 3
                 "11,LogicalAnd,val,Id 8,val,Id 10,val",
                 "12,RangeV,val,0,Id 11,1", "13,FoldSelect,val,Id 12,val,Id 11,val",
                 "14,Gather,Id 6,Id 13,val", "15,Gather,Id 2,Id 13,val",
                 "16,RangeV,val,100,Id 14,0", "17,Subtract,val,Id 16,val,Id 14,val",     # 100 - c
                 "18,RangeV,val,2,Id 14,0", "19,Multiply,val,Id 17,val,Id 18,val",       # * 2
                 "20,Multiply,val,Id 19,val,Id 15,val",                                  # * a
                 "21,RangeV,val,0,Id 20,0", "22,FoldSum,val,Id 21,val,Id 20,val",
                 "23,Project,s,Id 22,val", "24,MaterializeCompact,Id 23"])
    assert p.is_fused
    d = p.describe()
    assert "t.a in [11,+inf]" in d and "t.b in [3,3]" in d
    assert "(200 + -2*col" in d                                                       # (100 - c) * 2 folded into one factor


def test_disjunction_across_columns_is_a_condition_column():
    p = plan_of(["7,RangeV,val,10,Id 2,0", "8,Greater,val,Id 2,val,Id 7,val",
                 "9,Greater,val,Id 4,val,Id 7,val", "10,LogicalOr,val,Id 8,val,Id 9,val",  # a > 10 or b > 10
                 "11,RangeV,val,0,Id 10,1", "12,FoldSelect,val,Id 11,val,Id 10,val",
                 "13,Gather,Id 6,Id 12,val", "14,RangeV,val,0,Id 13,0", "15,FoldSum,val,Id 14,val,Id 13,val",
                 "16,MaterializeCompact,Id 15"])
    d = p.describe()                                       # (round 1 refused this shape; TPC-H Q19's predicate has it)
    assert p.is_fused and "cond(col0:[11,+inf] col1:[11,+inf] or) in [1,1]" in d, d


def test_value_that_is_not_a_condition_does_not_fuse_as_one():
    p = plan_of(["7,Multiply,val,Id 2,val,Id 4,val", "8,LogicalOr,val,Id 7,val,Id 6,val",          # (a * b) or c
                 "11,RangeV,val,0,Id 8,1", "12,FoldSelect,val,Id 11,val,Id 8,val",
                 "13,Gather,Id 6,Id 12,val", "14,RangeV,val,0,Id 13,0", "15,FoldSum,val,Id 14,val,Id 13,val",
                 "16,MaterializeCompact,Id 15"])
    assert not p.is_fused and "not a conjunction of per-column ranges" in p.describe()


def test_contradictory_predicate_is_detected():
    p = plan_of(["7,RangeV,val,10,Id 2,0", "8,Greater,val,Id 2,val,Id 7,val", "9,Greater,val,Id 7,val,Id 2,val",
                 "10,LogicalAnd,val,Id 8,val,Id 9,val",                                   # a > 10 and a < 10
                 "11,RangeV,val,0,Id 10,1", "12,FoldSelect,val,Id 11,val,Id 10,val",
                 "13,Gather,Id 6,Id 12,val", "14,RangeV,val,0,Id 13,0", "15,FoldSum,val,Id 14,val,Id 13,val",
                 "16,MaterializeCompact,Id 15"])
    assert p.is_fused and "[never]" in p.describe()


def test_unfiltered_aggregate_and_min_max_fuse():
    p = plan_of(["7,RangeV,val,0,Id 2,0", "8,FoldMax,val,Id 7,val,Id 2,val", "9,MaterializeCompact,Id 8",
                 "10,FoldMin,val,Id 7,val,Id 4,val", "11,MaterializeCompact,Id 10"])
    assert p.is_fused
    assert p.partial_spec() == (3, [_lib.REDUCE_SUM, _lib.REDUCE_MAX, _lib.REDUCE_MIN])


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """The boundary is a C ABI: include/vdl.h compiles as C11 and a C program links against libvdl.so (host-only
    context: parse, describe, and a loud VDL_ERR_DEVICE from vdl_run -- no CPU fallback)."""
    import subprocess

    hdr = os.path.join(ROOT, "include", "vdl.h")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c", hdr], check=True)
    exe = str(tmp_path / "host_only")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host_only.c"),
                    "-o", exe, "-L" + libdir, "-lvdl", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "general: 3 statement(s)" in out and "vdl_run without a device -> 4" in out


def test_every_compiled_plan_has_a_sharded_route():
    """vdl_plan_sharded_route for the 15 TPC-H plans the front end compiles (no GPU needed: the analysis is host code): partial words
    merged (fold), rows exchanged by key range, a semi-join set merged, the front's survivors gathered -- and for Q18 the chain of three
    of them (exchange up to its position set, the set's positions gathered, the second scan's survivors gathered), with the last
    resort behind it when that is switched off: the table's columns gathered once, the whole query on every rank.  Q20's tail feeds a semi-join set over suppliers from the groups: the
    ranks' outputs would not concatenate (a supplier with qualifying groups on two ranks would come out twice), so the exchange
    analysis refuses it and its fused front carries it (round 4)."""
    from mplan2vdl_amd import frontend
    meta = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
    cfg = frontend.load_metadata(meta)
    want = {1: "fold", 3: "exchange", 4: "set", 5: "exchange", 6: "fold", 9: "exchange", 10: "exchange", 11: "exchange", 12: "fold", 14: "fold",
            15: "front", 16: "front", 18: "chain", 19: "fold", 20: "front"}
    for q, route in want.items():
        text = frontend.compile_plan(open(os.path.join(meta, "%02d.sql.mplan" % q)).read(), cfg)
        e = m.Engine(device=None)
        p = e.parse(text)
        p.set_sharded_table("partsupp" if q in (11, 16) else "lineitem")
        assert p.sharded_route() == (route, route != "exchange"), q
        if route == "chain":                                # without it the last resort; that can be switched off too: then the reasons are the answer
            os.environ["VDL_NO_CHAIN_ROUTE"] = "1"
            try:
                assert p.sharded_route() == ("replicate", True)
                os.environ["VDL_NO_REPLICATE_ROUTE"] = "1"
                with pytest.raises(m.VdlError) as err:
                    p.sharded_route()
                assert "more than one Partition" in str(err.value) and "no fused front" in str(err.value) and "VDL_NO_CHAIN_ROUTE" in str(err.value)
            finally:
                del os.environ["VDL_NO_CHAIN_ROUTE"]
                os.environ.pop("VDL_NO_REPLICATE_ROUTE", None)
