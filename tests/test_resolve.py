"""Result decoding (the reference's resolve.py contract) on the example reply quoted in its own comments."""
import os

from conftest import ROOT
from mplan2vdl_amd import resolve

DICT = os.path.join(ROOT, "tests", "golden", "tpch10noorder", "dictionary.csv")

# /root/reference/resolve.py:8-32
EXAMPLE = {"results": {"tmp66": {".o_orderpriority__orders__o_orderpriority": [16, 40, 72, 104, 128]},
                       "tmp75": {".order_count": [311, 263, 266, 274, 36783]}},
           "timings": {"timeInMicrosecondsForFragment12": 215, "timeInMicrosecondsForFragment13": 565}}


def test_reference_example_reply_decodes():
    r = resolve.load_dictionary(DICT)
    names, rows = resolve.decode(EXAMPLE, r)
    assert names == [".o_orderpriority", ".order_count"]
    assert [row[1] for row in rows] == [311, 263, 266, 274, 36783]
    # codes present in dictionary.csv:78-79 are decoded, unknown codes pass through (resolve.py:92-96)
    assert [row[0] for row in rows] == [16, "1-URGENT", 72, "2-HIGH", 128]


def test_q1_reply_decodes_flags_and_pads():
    r = resolve.load_dictionary(DICT)
    reply = {"results": {"tmp40": {".l_returnflag__lineitem__l_returnflag": [16, 40, 64]},
                         "tmp51": {".sum_qty": [1, 2]}, "tmp52": {".none": None}}}
    warnings = []
    names, rows = resolve.decode(reply, r, warnings.append)
    assert names == [".l_returnflag", ".sum_qty", ".none"]
    assert [row[0] for row in rows] == ["N", "R", "A"]            # dictionary.csv:80-82
    assert rows[2][1] == "-" and rows[0][2] == "-"                # padded to the longest column
    assert any("null" in w for w in warnings)
