#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

The reference has no executor and no result goldens, so result vectors are
produced here by TWO independent CPU evaluations that must agree before anything is written:
the scalar VDL interpreter (oracle/vdl_oracle.c) running the VDL fixtures, and the fused
SQL-semantics loops written from the SQL text in the plan headers
(/root/reference/tests/tpch10noorder/06.sql.mplan:1-9, 01.sql.mplan:1-18).  Inputs are the
synthetic SF0.01 lineitem columns (60175 rows, /root/reference/tests/tpchnoorder/bounds.csv:59).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from mplan2vdl_amd import datagen  # noqa: E402


def main():
    n = datagen.LINEITEM_ROWS["sf0.01"]
    cols = datagen.generate_table(datagen.Q1_COLUMNS, 0, n)
    o = oracle.Oracle()
    for k, v in cols.items():
        o.add_column(k, v)
    q6 = o.run(open(os.path.join(HERE, "q6.vdl")).read())["results"]
    rev, cnt = oracle.sql_q6(*[cols[c] for c in datagen.Q6_COLUMNS])
    assert q6 == {"tmp42": {".revenue": [rev]}}, (q6, rev)
    q1 = o.run(open(os.path.join(HERE, "q1.vdl")).read())["results"]
    tab = oracle.sql_q1(cols["lineitem.l_shipdate"], cols["lineitem.l_returnflag"], cols["lineitem.l_linestatus"],
                        cols["lineitem.l_quantity"], cols["lineitem.l_extendedprice"], cols["lineitem.l_discount"],
                        cols["lineitem.l_tax"])
    order = ["l_returnflag__lineitem__l_returnflag", "l_linestatus__lineitem__l_linestatus", "sum_qty", "sum_base_price",
             "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc", "count_order"]
    flat = {list(v.keys())[0][1:]: list(v.values())[0] for v in q1.values()}
    for j, name in enumerate(order):
        assert flat[name] == [int(x) for x in tab[:, j]], name
    json.dump({"rows": n, "seed": datagen.SEED, "selected_rows": cnt, "results": q6},
              open(os.path.join(HERE, "q6_sf001.json"), "w"), indent=1)
    json.dump({"rows": n, "seed": datagen.SEED, "results": q1}, open(os.path.join(HERE, "q1_sf001.json"), "w"), indent=1)
    print("golden vectors written")


if __name__ == "__main__":
    main()
