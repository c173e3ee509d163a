"""TPC-H queries evaluated from their SQL with numpy, over the coherent synthetic catalog (mplan2vdl_amd/catalog.py).

TEST INFRASTRUCTURE.  One function per plan the front end compiles and that has no evaluator elsewhere (Q1 / Q6:
oracle/vdl_oracle.c orc_sql_*; Q3: tests/helpers.sql_q3).  Each follows the query text in the header of
/root/reference/tests/tpch10noorder/NN.sql.mplan (cited per function; Q11's header carries no SQL and Q15 selects from
the view revenue0, so those two follow the TPC-H specification's text for the query and the view) -- relational
semantics only: filters, joins, grouping, aggregation.  Nothing here knows about VDL, Partition, Scatter or folds; that
independence is the point: oracle/vdl_oracle.c running the compiled program must produce the same rows.

Numbers are the scaled integers MonetDB types them as in the plan body under the header (decimal(15,2) = value * 100;
`1 - l_discount` = 100 - l_discount at scale 2; a product adds scales; a quotient subtracts them and truncates; a cast
to a scale multiplies / divides by the power of ten, /root/reference/src/Vlite.hs:937-957), strings are dictionary codes
(dictionary.csv) or heap offsets, dates are day numbers (Mplan.hs:46-57).  Equality joins between a foreign key and
the key it references are evaluated on the VALUES (l_orderkey = o_orderkey), which the coherent catalog keeps
consistent with the join indices the compiled programs use.

Every function returns the result as a sorted list of row tuples in the column order of the SELECT list; the queries
carry no ORDER BY (directory "noorder"), so row order is not part of the answer.
"""
import calendar
import datetime

import numpy as np

from mplan2vdl_amd import catalog
from mplan2vdl_amd.frontend.mplan import day_count
from helpers import sql_like

# value columns the SQL texts name that a compiled program may reach through a join index instead
EXTRA = ["lineitem.l_orderkey", "lineitem.l_partkey", "lineitem.l_suppkey", "orders.o_orderkey", "orders.o_custkey", "customer.c_custkey",
         "customer.c_nationkey", "supplier.s_suppkey", "supplier.s_nationkey", "part.p_partkey", "partsupp.ps_partkey", "partsupp.ps_suppkey",
         "nation.n_nationkey", "nation.n_regionkey", "region.r_regionkey"]


def add_months(datestr, months):
    y, m, d = (int(x) for x in datestr.split("-"))
    m0 = m - 1 + months
    y, m = y + m0 // 12, m0 % 12 + 1
    return "%04d-%02d-%02d" % (y, m, min(d, calendar.monthrange(y, m)[1]))


class Db:
    def __init__(self, meta_dir, cfg, vdl_text, scale, seed):
        self.cols = catalog.synth_columns(meta_dir, cfg, vdl_text, scale=scale, seed=seed, extra=EXTRA)
        self.codes = catalog._per_column_codes(meta_dir)
        self._more = (meta_dir, cfg, scale, seed)

    def __getitem__(self, path):
        if path not in self.cols:
            meta_dir, cfg, scale, seed = self._more
            self.cols.update(catalog.synth_columns(meta_dir, cfg, "", scale=scale, seed=seed, extra=[path]))
        return self.cols[path].astype(np.int64) if not path.endswith(".heap") else self.cols[path]

    def code(self, table, col, string):
        return self.codes[(table, col)][string]

    def strings(self, table, col):
        """The column's values as Python strings (read out of its heap: NUL-terminated at the stored offsets)."""
        heap = bytes(self["%s.%s.heap" % (table, col)].astype(np.int8).tobytes())
        memo = {}
        out = []
        for off in self["%s.%s" % (table, col)].tolist():
            if off not in memo:
                memo[off] = heap[off:heap.index(b"\0", off)].decode()
            out.append(memo[off])
        return out

    def like(self, table, col, pattern):
        return np.array([bool(sql_like(s, pattern)) for s in self.strings(table, col)], dtype=bool)


def where_key(keys, wanted):
    """row of `keys` (unique) holding each value of `wanted`; every wanted value must be present (foreign keys are)."""
    order = np.argsort(keys, kind="stable")
    pos = np.searchsorted(keys[order], wanted)
    assert np.array_equal(keys[order][pos], wanted)
    return order[pos]


def group_rows(keys, aggs):
    """GROUP BY keys (tuple of equal-length arrays): [(key..., agg...)] with aggs = [(array, 'sum' | 'count' | 'max')]."""
    n = len(keys[0])
    if n == 0:
        return []
    stacked = np.stack([np.asarray(k, dtype=np.int64) for k in keys], axis=1)
    uniq, inv = np.unique(stacked, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    cols = []
    for arr, how in aggs:
        if how == "count":
            cols.append(np.bincount(inv, minlength=len(uniq)).astype(np.int64))
        elif how == "sum":
            acc = np.zeros(len(uniq), np.int64)
            np.add.at(acc, inv, np.asarray(arr, dtype=np.int64))
            cols.append(acc)
        else:
            acc = np.full(len(uniq), np.iinfo(np.int64).min, np.int64)
            np.maximum.at(acc, inv, np.asarray(arr, dtype=np.int64))
            cols.append(acc)
    return [tuple(int(x) for x in uniq[g]) + tuple(int(c[g]) for c in cols) for g in range(len(uniq))]


def revenue(db, rows):
    """l_extendedprice * (1 - l_discount): scale 2 x scale 2 = scale 4."""
    return db["lineitem.l_extendedprice"][rows] * (100 - db["lineitem.l_discount"][rows])


def date_range(col, start, months):
    return (col >= day_count(start)) & (col < day_count(add_months(start, months)))


def q4(db):
    """04.sql.mplan:1-20: orders of 1993-Q3 with at least one lineitem received after its commit date, counted per priority."""
    late = db["lineitem.l_commitdate"] < db["lineitem.l_receiptdate"]
    has_late = np.isin(db["orders.o_orderkey"], db["lineitem.l_orderkey"][late])
    o = np.nonzero(date_range(db["orders.o_orderdate"], "1993-07-01", 3) & has_late)[0]
    return sorted(group_rows((db["orders.o_orderpriority"][o],), [(None, "count")]))


def q5(db):
    """05.sql.mplan:1-22: revenue per nation of region ASIA, customer and supplier from the same nation, orders of 1994."""
    l = np.arange(len(db["lineitem.l_orderkey"]))
    o = where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"])
    c = where_key(db["customer.c_custkey"], db["orders.o_custkey"][o])
    s = where_key(db["supplier.s_suppkey"], db["lineitem.l_suppkey"])
    n = where_key(db["nation.n_nationkey"], db["supplier.s_nationkey"][s])
    r = where_key(db["region.r_regionkey"], db["nation.n_regionkey"][n])
    keep = (db["customer.c_nationkey"][c] == db["supplier.s_nationkey"][s]) & (db["region.r_name"][r] == db.code("region", "r_name", "ASIA")) & \
        date_range(db["orders.o_orderdate"][o], "1994-01-01", 12)
    l = l[keep]
    return sorted(group_rows((db["nation.n_name"][n[keep]],), [(revenue(db, l), "sum")]))


def q9(db):
    """09.sql.mplan:1-30: profit per nation and order year over the parts named like '%green%'."""
    p = where_key(db["part.p_partkey"], db["lineitem.l_partkey"])
    keep = db.like("part", "p_name", "%green%")[p]
    l = np.nonzero(keep)[0]
    s = where_key(db["supplier.s_suppkey"], db["lineitem.l_suppkey"][l])
    n = where_key(db["nation.n_nationkey"], db["supplier.s_nationkey"][s])
    o = where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"][l])
    # partsupp: the row with (ps_partkey, ps_suppkey) = (l_partkey, l_suppkey); pairs are unique
    pair = db["partsupp.ps_partkey"] * (1 << 32) + db["partsupp.ps_suppkey"]
    ps = where_key(pair, db["lineitem.l_partkey"][l] * (1 << 32) + db["lineitem.l_suppkey"][l])
    year = np.array([(datetime.date.fromordinal(int(d) - 365)).year for d in db["orders.o_orderdate"][o]], dtype=np.int64)
    amount = revenue(db, l) - db["partsupp.ps_supplycost"][ps] * db["lineitem.l_quantity"][l]
    return sorted(group_rows((db["nation.n_name"][n], year), [(amount, "sum")]))


def q10(db):
    """10.sql.mplan:1-31: revenue lost to returned items per customer, orders of 1993-Q4."""
    l = np.nonzero(db["lineitem.l_returnflag"] == db.code("lineitem", "l_returnflag", "R"))[0]
    o = where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"][l])
    keep = date_range(db["orders.o_orderdate"][o], "1993-10-01", 3)
    l, o = l[keep], o[keep]
    c = where_key(db["customer.c_custkey"], db["orders.o_custkey"][o])
    n = where_key(db["nation.n_nationkey"], db["customer.c_nationkey"][c])
    out = []
    for cust, name, rev in group_rows((c, db["nation.n_name"][n]), [(revenue(db, l), "sum")]):     # c_custkey is unique per customer row
        out.append((int(db["customer.c_custkey"][cust]), int(db["customer.c_name"][cust]), rev, int(db["customer.c_acctbal"][cust]), name,
                    int(db["customer.c_address"][cust]), int(db["customer.c_phone"][cust]), int(db["customer.c_comment"][cust])))
    return sorted(out)


def q11(db):
    """TPC-H Q11 (the plan file carries no SQL header; text of the specification, fraction 0.0001 / SF with SF = 10, the
    literal `decimal(7,6) "10"` of 11.sql.mplan:33): parts whose stock value in GERMANY exceeds that fraction of the total."""
    s = where_key(db["supplier.s_suppkey"], db["partsupp.ps_suppkey"])
    n = where_key(db["nation.n_nationkey"], db["supplier.s_nationkey"][s])
    ps = np.nonzero(db["nation.n_name"][n] == db.code("nation", "n_name", "GERMANY"))[0]
    value = db["partsupp.ps_supplycost"][ps] * db["partsupp.ps_availqty"][ps]                       # scale 2
    threshold = int(value.sum()) * 10 // 10 ** 6                                                   # x 0.000010 (scale 6), back to scale 2
    return sorted(r for r in group_rows((db["partsupp.ps_partkey"][ps],), [(value, "sum")]) if r[1] > threshold)


def q12(db):
    """12.sql.mplan:1-30: late lineitems of 1994 shipped by MAIL / SHIP, counted by ship mode and order priority class."""
    ship, commit, receipt = db["lineitem.l_shipdate"], db["lineitem.l_commitdate"], db["lineitem.l_receiptdate"]
    modes = [db.code("lineitem", "l_shipmode", m) for m in ("MAIL", "SHIP")]
    l = np.nonzero(np.isin(db["lineitem.l_shipmode"], modes) & (commit < receipt) & (ship < commit) & date_range(receipt, "1994-01-01", 12))[0]
    o = where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"][l])
    high = np.isin(db["orders.o_orderpriority"][o], [db.code("orders", "o_orderpriority", p) for p in ("1-URGENT", "2-HIGH")])
    return sorted(group_rows((db["lineitem.l_shipmode"][l],), [(high.astype(np.int64), "sum"), ((~high).astype(np.int64), "sum")]))


def q14(db):
    """14.sql.mplan:1-13: share of promotion parts in the revenue of September 1995.  100.00 * sum(promo) / sum(all) as the
    plan types it: decimal(4,1) "1" * (decimal(19,8)[sum promo] / sum all), i.e. the truncated quotient at scale 4."""
    l = np.nonzero(date_range(db["lineitem.l_shipdate"], "1995-09-01", 1))[0]
    if len(l) == 0:
        return []
    p = where_key(db["part.p_partkey"], db["lineitem.l_partkey"][l])
    promo = db.like("part", "p_type", "PROMO%")[p]
    rev = revenue(db, l)
    num, den = int(rev[promo].sum()) * 10 ** 4, int(rev.sum())
    q = abs(num) // abs(den) * (1 if (num >= 0) == (den >= 0) else -1) if den else 0                # C truncation, x / 0 = 0
    return [(1 * q,)]


def q15(db):
    """15.sql.mplan:1-14 over the view revenue0 of the TPC-H specification (supplier_no = l_suppkey, total_revenue =
    sum(l_extendedprice * (1 - l_discount)) for shipments of 1996-Q1): the supplier(s) with the largest total."""
    l = np.nonzero(date_range(db["lineitem.l_shipdate"], "1996-01-01", 3))[0]
    view = group_rows((db["lineitem.l_suppkey"][l],), [(revenue(db, l), "sum")])
    if not view:
        return []
    top = max(r[1] for r in view)
    out = []
    for supp, total in view:
        if total == top:
            s = int(where_key(db["supplier.s_suppkey"], np.array([supp]))[0])
            out.append((supp, int(db["supplier.s_name"][s]), int(db["supplier.s_address"][s]), int(db["supplier.s_phone"][s]), total))
    return sorted(out)


def q16(db, as_compiled=True):
    """16.sql.mplan:1-30: suppliers per (brand, type, size) for the wanted parts, without the suppliers complained about.

    as_compiled: the reference does not compile the NOT IN of this query into an anti-join.  In handleGatherJoin
    (/root/reference/src/Vlite.hs:1199-1229) the name `selectmask` is re-bound to the FoldSelect over the join's boolean --
    a vector of POSITIONS, EPS where the row found no partner -- before the LeftAnti case computes
    `antiboolean = ones_ selectmask -. selectmask`: 1 - position, EPS for the rows without partner.  The rows that
    survive are therefore the partsupp rows that DO have a complained-about supplier, except the one at position 1
    (1 - 1 = 0).  The front-end restatement reproduces this, oracle and engine execute it faithfully, and this flag
    states the same rows in SQL terms, so that everything else of the plan (two nested GROUP BYs, LIKE, IN lists,
    count distinct) is still pinned.  as_compiled=False is the query as written."""
    bad = db["supplier.s_suppkey"][db.like("supplier", "s_comment", "%Customer%Complaints%")]
    p = where_key(db["part.p_partkey"], db["partsupp.ps_partkey"])
    part_ok = (db["part.p_brand"] != db.code("part", "p_brand", "Brand#45")) & ~db.like("part", "p_type", "MEDIUM POLISHED%") & \
        np.isin(db["part.p_size"], [49, 14, 23, 45, 19, 3, 36, 9])
    supplier_ok = ~np.isin(db["partsupp.ps_suppkey"], bad)
    if as_compiled:
        supplier_ok = ~supplier_ok & (np.arange(len(supplier_ok)) != 1)
    ps = np.nonzero(part_ok[p] & supplier_ok)[0]
    distinct = group_rows((db["part.p_brand"][p[ps]], db["part.p_type"][p[ps]], db["part.p_size"][p[ps]], db["partsupp.ps_suppkey"][ps]), [])
    if not distinct:
        return []
    d = np.array(distinct, dtype=np.int64)
    return sorted(group_rows((d[:, 0], d[:, 1], d[:, 2]), [(None, "count")]))


def q18(db):
    """18.sql.mplan:1-31: orders of more than 300 units, with their customer."""
    per_order = group_rows((db["lineitem.l_orderkey"],), [(db["lineitem.l_quantity"], "sum")])
    big = np.array([k for k, q in per_order if q > 300 * 100], dtype=np.int64)
    l = np.nonzero(np.isin(db["lineitem.l_orderkey"], big))[0]
    o = where_key(db["orders.o_orderkey"], db["lineitem.l_orderkey"][l])
    c = where_key(db["customer.c_custkey"], db["orders.o_custkey"][o])
    out = []
    for cust, order, qty in group_rows((c, o), [(db["lineitem.l_quantity"][l], "sum")]):
        out.append((int(db["customer.c_name"][cust]), int(db["customer.c_custkey"][cust]), int(db["orders.o_orderkey"][order]),
                    int(db["orders.o_orderdate"][order]), int(db["orders.o_totalprice"][order]), qty))
    return sorted(out)


def q19(db):
    """19.sql.mplan:1-36 (the header is cut off after the second branch; the third is the specification's: Brand#34, LG
    containers, quantity 20..30, size 1..15): discounted revenue of three brand / container / quantity / size combinations."""
    p = where_key(db["part.p_partkey"], db["lineitem.l_partkey"])
    brand, cont, size, qty = db["part.p_brand"][p], db["part.p_container"][p], db["part.p_size"][p], db["lineitem.l_quantity"]

    def branch(b, containers, q_lo, size_hi):
        return (brand == db.code("part", "p_brand", b)) & np.isin(cont, [db.code("part", "p_container", x) for x in containers]) & \
            (qty >= q_lo * 100) & (qty <= (q_lo + 10) * 100) & (size >= 1) & (size <= size_hi)

    common = np.isin(db["lineitem.l_shipmode"], [db.code("lineitem", "l_shipmode", m) for m in ("AIR", "AIR REG")]) & \
        (db["lineitem.l_shipinstruct"] == db.code("lineitem", "l_shipinstruct", "DELIVER IN PERSON"))
    keep = common & (branch("Brand#12", ("SM CASE", "SM BOX", "SM PACK", "SM PKG"), 1, 5) |
                     branch("Brand#23", ("MED BAG", "MED BOX", "MED PKG", "MED PACK"), 10, 10) |
                     branch("Brand#34", ("LG CASE", "LG BOX", "LG PACK", "LG PKG"), 20, 15))
    l = np.nonzero(keep)[0]
    return [(int(revenue(db, l).sum()),)] if len(l) else []


def q20(db):
    """20.sql.mplan:1-37: suppliers of CANADA holding more of a 'forest%' part than half of what was shipped of it in 1994."""
    forest = db["part.p_partkey"][db.like("part", "p_name", "forest%")]
    ps = np.nonzero(np.isin(db["partsupp.ps_partkey"], forest))[0]
    l = np.nonzero(date_range(db["lineitem.l_shipdate"], "1994-01-01", 12))[0]
    pair_l = db["lineitem.l_partkey"][l] * (1 << 32) + db["lineitem.l_suppkey"][l]
    shipped = dict(group_rows((pair_l,), [(db["lineitem.l_quantity"][l], "sum")]))
    good = set()
    for row in ps:
        total = shipped.get(int(db["partsupp.ps_partkey"][row]) * (1 << 32) + int(db["partsupp.ps_suppkey"][row]))
        if total is not None and int(db["partsupp.ps_availqty"][row]) > 5 * total // 1000:          # 0.5 (scale 1) x sum (scale 2), cast to int
            good.add(int(db["partsupp.ps_suppkey"][row]))
    n = where_key(db["nation.n_nationkey"], db["supplier.s_nationkey"])
    s = np.nonzero(np.isin(db["supplier.s_suppkey"], sorted(good)) & (db["nation.n_name"][n] == db.code("nation", "n_name", "CANADA")))[0]
    return sorted((int(db["supplier.s_name"][k]), int(db["supplier.s_address"][k])) for k in s)


EVALUATORS = {4: q4, 5: q5, 9: q9, 10: q10, 11: q11, 12: q12, 14: q14, 15: q15, 16: q16, 18: q18, 19: q19, 20: q20}


def rows_of(results):
    """The reply dict of a run as sorted row tuples (output columns in program order = SELECT-list order)."""
    cols = [list(v.values())[0] for v in results.values()]
    return sorted(zip(*cols)) if cols and len(cols[0]) else []
