#!/usr/bin/env python3
"""Writes tests/golden/plans/qNN[_variant].json: input recipe + expected outputs for every TPC-H plan the front end compiles.

The reference holds no result vectors, so each fixture is produced by the CPU oracle (oracle/vdl_oracle.c) running the
compiled VDL program over the coherent synthetic catalog (mplan2vdl_amd/catalog.py, scale / seed recorded in the file),
and is only written when an INDEPENDENT evaluation of the query's SQL over the same columns gives the same rows:
tests/sql_eval.py for Q4,5,9,10,11,12,14,15,16,18,19,20; for Q1, Q3 and Q6 the evaluators that already pin them on the
generator's lineitem (oracle.sql_q1 / sql_q6, helpers.sql_q3) are restated here over the catalog's columns.
Two reference compiler bugs keep the program of a plan from meaning its SQL (tests/sql_eval.py, q16 / vlite.DISTINCT_RANGEC):
Q16 and Q18 are therefore written twice -- as the reference compiles them ("as_compiled": pinned by the oracle alone for
Q18, by the as-compiled SQL reading for Q16) and with distinct RangeC identities ("distinct_rangec": pinned by SQL).

    python tests/golden/make_plan_goldens.py        (run from the repo root; needs no GPU)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import sql_eval  # noqa: E402
from helpers import oracle_run  # noqa: E402
from mplan2vdl_amd import frontend  # noqa: E402
from mplan2vdl_amd.frontend.mplan import day_count  # noqa: E402

META = os.path.join(HERE, "tpch10noorder")
PLANS = [1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20]
SCALE, SEED = 5e-4, 3
NEEDS_DISTINCT_RANGEC = (16, 18)


def q1(db):
    """01.sql.mplan:1-21."""
    l = np.nonzero(db["lineitem.l_shipdate"] <= day_count("1998-12-01") - 90)[0]
    qty, ep, disc, tax = (db["lineitem." + c][l] for c in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"))
    rows = sql_eval.group_rows((db["lineitem.l_returnflag"][l], db["lineitem.l_linestatus"][l]),
                               [(qty, "sum"), (ep, "sum"), (ep * (100 - disc), "sum"), (ep * (100 - disc) * (100 + tax), "sum"), (disc, "sum"), (None, "count")])
    return sorted((rf, ls, sq, sp, sd, sc, sq // n, sp // n, sdisc // n, n) for rf, ls, sq, sp, sd, sc, sdisc, n in rows)


def q3(db):
    """03.sql.mplan:1-19."""
    o = sql_eval.where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"])
    c = sql_eval.where_key(db["customer.c_custkey"], db["orders.o_custkey"][o])
    keep = (db["customer.c_mktsegment"][c] == db.code("customer", "c_mktsegment", "BUILDING")) & \
        (db["orders.o_orderdate"][o] < day_count("1995-03-15")) & (db["lineitem.l_shipdate"] > day_count("1995-03-15"))
    l = np.nonzero(keep)[0]
    rows = sql_eval.group_rows((db["lineitem.l_orderkey"][l], db["orders.o_orderdate"][o[l]], db["orders.o_shippriority"][o[l]]), [(sql_eval.revenue(db, l), "sum")])
    return sorted((k, rev, d, p) for k, d, p, rev in rows)


def q6(db):
    """06.sql.mplan:1-9."""
    l = np.nonzero(sql_eval.date_range(db["lineitem.l_shipdate"], "1994-01-01", 12) & (db["lineitem.l_discount"] >= 5) & (db["lineitem.l_discount"] <= 7) &
                   (db["lineitem.l_quantity"] < 2400))[0]
    return [(int((db["lineitem.l_extendedprice"][l] * db["lineitem.l_discount"][l]).sum()),)] if len(l) else []


SQL = dict(sql_eval.EVALUATORS)
SQL.update({1: q1, 3: q3, 6: q6})


def build(n, distinct_rangec, cfg):
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg, distinct_rangec=distinct_rangec)
    db = sql_eval.Db(META, cfg, text, SCALE, SEED)
    loads = {ln.split(",")[2].split(";;")[0].strip() for ln in text.splitlines() if ln.split(",")[1] == "Load"}
    cols = {k: v for k, v in db.cols.items() if k in loads}
    return text, db, cols


def main():
    cfg = frontend.load_metadata(META)
    out_dir = os.path.join(HERE, "plans")
    os.makedirs(out_dir, exist_ok=True)
    for n in PLANS:
        for variant in (["as_compiled", "distinct_rangec"] if n in NEEDS_DISTINCT_RANGEC else ["as_compiled"]):
            text, db, cols = build(n, variant == "distinct_rangec", cfg)
            results = oracle_run(text, cols)
            rows = sql_eval.rows_of(results)
            sql_rows = SQL[n](db)
            pinned_by_sql = rows == sql_rows
            if n == 18 and variant == "as_compiled":
                assert not pinned_by_sql, "Q18 as compiled partitions its second key over the first key's pivots: it cannot equal its SQL"
            else:
                assert pinned_by_sql, "Q%d (%s): oracle and SQL differ\n oracle %r\n sql    %r" % (n, variant, rows[:4], sql_rows[:4])
                assert rows, "Q%d selects nothing at this scale / seed: pick another" % n
            name = "q%02d.json" % n if variant == "as_compiled" else "q%02d_%s.json" % (n, variant)
            json.dump({"plan": "%02d.sql.mplan" % n, "variant": variant, "scale": SCALE, "seed": SEED,
                       "program_sha256": hashlib.sha256(text.encode()).hexdigest(), "pinned_by_sql": pinned_by_sql,
                       "rows": len(rows), "results": results}, open(os.path.join(out_dir, name), "w"), separators=(",", ":"))
            print("Q%02d %-16s %5d rows  sql=%s" % (n, variant, len(rows), pinned_by_sql))


if __name__ == "__main__":
    main()
