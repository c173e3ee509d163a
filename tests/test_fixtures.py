"""Pins the VDL fixtures and constants to what the reference itself states."""
import datetime

from conftest import golden

# verbatim from /root/reference/README.md:40-52 (the only golden output in the reference)
README_HEAD = [
    "1,Load,lineitem.l_quantity",
    "2,Project,val,Id 1,l_quantity",
    "3,Load,lineitem.l_shipdate",
    "4,Project,val,Id 3,l_shipdate",
    "5,RangeV,val,728294,Id 2,0",
    "6,Greater,val,Id 4,val,Id 5,val",
    "7,Equals,val,Id 5,val,Id 4,val",
    "8,LogicalOr,val,Id 6,val,Id 7,val",
    "9,RangeV,val,728659,Id 2,0",
]
README_TAIL = [
    "40,FoldSum,val,Id 34,val,Id 39,val",
    "41,Project,revenue,Id 40,val",
    "42,MaterializeCompact,Id 41",
]


def test_q6_fixture_matches_readme_golden_lines():
    lines = golden("q6.vdl").strip().split("\n")
    assert len(lines) == 42                       # README.md:39-53: statements 1..42
    assert lines[:9] == README_HEAD
    assert lines[-3:] == README_TAIL
    assert [int(l.split(",")[0]) for l in lines] == list(range(1, 43))


def test_q1_fixture_shape():
    lines = golden("q1.vdl").strip().split("\n")
    assert len(lines) == 101
    assert sum(1 for l in lines if ",MaterializeCompact," in l) == 10     # 10 output columns (01.sql.mplan:1-18)
    assert "33,RangeC,val,0,32,1" in lines                                 # dense 5-bit key domain (Vlite.hs:1076-1098)


def day_number(y, m, d):
    """Mplan.hs:51-57: proleptic ordinal + 365 (year 0 counted as a 366-day year ahead of 0001-01-01... = ordinal + 365)."""
    return datetime.date(y, m, d).toordinal() + 365


def test_date_constants():
    assert day_number(1994, 1, 1) == 728294       # README.md:44
    assert day_number(1995, 1, 1) == 728659       # README.md:48
    assert day_number(1998, 9, 2) == 729999       # Q1: 1998-12-01 - 90 days (01.sql.mplan)
    assert day_number(1995, 3, 15) == 728732      # Q3
    text = golden("q6.vdl")
    assert ",728294," in text and ",728659," in text
    assert ",729999," in golden("q1.vdl")
