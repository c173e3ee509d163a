"""Shared helpers for the parity tests."""
import numpy as np

from mplan2vdl_amd import datagen


def lineitem(names, n, seed=datagen.SEED, row0=0):
    return {name: datagen.generate(datagen.LINEITEM[name], row0, n, seed) for name in names}


def oracle_run(text, cols):
    import oracle

    o = oracle.Oracle()
    for k, v in cols.items():
        o.add_column(k, v)
    try:
        return o.run(text)["results"]
    finally:
        o.close()


def engine_with(cols, device=0):
    import mplan2vdl_amd as m

    e = m.Engine(device=device)
    for k, v in cols.items():
        e.upload(k, v)
    return e


def prog(*lines):
    return "\n".join(lines) + "\n"


def rand_cols(rng, n, spec):
    """spec: {name: (dtype, lo, hi)} -> uniform random integer columns."""
    return {k: rng.integers(lo, hi + 1, size=n, dtype=np.int64).astype(dt) for k, (dt, lo, hi) in spec.items()}


def sql_q3(t):
    """TPC-H Q3 evaluated from its SQL text (/root/reference/tests/tpch10noorder/03.sql.mplan:1-19) with numpy,
    over the join-index catalog of datagen.q3_tables.  Returns the four output columns in the order the
    VDL program produces them: ascending composite key (l_orderkey, o_orderdate, o_shippriority)."""
    seg = t["customer.c_mktsegment"]
    o_date, o_prio, o_cust = t["orders.o_orderdate"], t["orders.o_shippriority"], t["orders.orders_customer"]
    l_ord, l_key = t["lineitem.lineitem_orders"], t["lineitem.l_orderkey"].astype(np.int64)
    l_ship, l_ep, l_disc = t["lineitem.l_shipdate"], t["lineitem.l_extendedprice"], t["lineitem.l_discount"]
    order_ok = (o_date < 728732) & (seg[o_cust] == 16)            # date '1995-03-15', 'BUILDING'
    li_ok = (l_ship > 728732) & order_ok[l_ord]
    rows = np.nonzero(li_ok)[0]
    key = ((l_key[rows] - 1) << 12) | (o_date[l_ord[rows]].astype(np.int64) - 727563)
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rev = np.zeros(len(uniq), np.int64)
    np.add.at(rev, inv, l_ep[rows] * (100 - l_disc[rows]))
    fr = rows[first]
    return {"l_orderkey__lineitem__l_orderkey": [int(x) for x in l_key[fr]], "revenue": [int(x) for x in rev],
            "o_orderdate__orders__o_orderdate": [int(x) for x in o_date[l_ord[fr]]],
            "o_shippriority__orders__o_shippriority": [int(x) for x in o_prio[l_ord[fr]]]}


def make_heap(strings, base=16, align=8):
    """MonetDB-style string heap: NUL-terminated strings at `align`-byte aligned offsets starting at
    `base` (the codes of the reference's dictionary.csv are such offsets: 'BUILDING' -> 16).
    Returns (heap as int8 array, {string: offset})."""
    buf = bytearray(base)
    where = {}
    for s in strings:
        if s in where:
            continue
        while len(buf) % align:
            buf.append(0)
        where[s] = len(buf)
        buf += s.encode() + b"\0"
    return np.frombuffer(bytes(buf), dtype=np.int8).copy(), where


def sql_like(s, pattern):
    """SQL LIKE ('%', '_', no escape) through a regular expression: an independent statement of the
    semantics oracle/vdl_oracle.c:like_match implements."""
    import re

    rx = "".join(".*" if ch == "%" else "." if ch == "_" else re.escape(ch) for ch in pattern)
    return 1 if re.fullmatch(rx, s, flags=re.S) else 0
