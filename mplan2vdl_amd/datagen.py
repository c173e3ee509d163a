"""Synthetic TPC-H-shaped columns.

The reference ships no data, only catalog metadata: value ranges in
/root/reference/tests/tpch10noorder/bounds.csv:59-79 and storage widths in
storage.csv:188-208.  Columns are produced by a counter-based hash so that any GPU can
materialise any row range in place:

    v(row) = add + mul * (lo + splitmix64(seed ^ col_id * PHI ^ row) mod (hi - lo + 1))

``col_id`` is the FNV-1a-64 hash of the column key path ("lineitem.l_shipdate").
The device generator (csrc/vdl_kernels.hip: k_gen_column) and this numpy one are
bit-identical (tests/test_datagen.py, tests/test_gpu_parity.py).
"""
from collections import namedtuple

import numpy as np

SEED = 0x5EED0006
PHI = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1

ColumnSpec = namedtuple("ColumnSpec", "name dtype lo hi mul add")

# lineitem row counts: SF0.01 tests/tpchnoorder/bounds.csv:59, SF10 tests/tpch10noorder/bounds.csv:59,
# SF1/SF100 from the TPC-H specification.
LINEITEM_ROWS = {"sf0.01": 60175, "sf1": 6001215, "sf10": 59986052, "sf100": 600037902}

LINEITEM = {
    "lineitem.l_shipdate":      ColumnSpec("lineitem.l_shipdate", np.int32, 727564, 730089, 1, 0),
    "lineitem.l_discount":      ColumnSpec("lineitem.l_discount", np.int64, 0, 10, 1, 0),
    "lineitem.l_quantity":      ColumnSpec("lineitem.l_quantity", np.int64, 1, 50, 100, 0),
    "lineitem.l_extendedprice": ColumnSpec("lineitem.l_extendedprice", np.int64, 90091, 10494950, 1, 0),
    "lineitem.l_tax":           ColumnSpec("lineitem.l_tax", np.int64, 0, 8, 1, 0),
    # dictionary codes {16,40,64} / {16,40}: dictionary.csv:80-82, bounds.csv:67-68 (3 trailing zero bits)
    "lineitem.l_returnflag":    ColumnSpec("lineitem.l_returnflag", np.int32, 0, 2, 24, 16),
    "lineitem.l_linestatus":    ColumnSpec("lineitem.l_linestatus", np.int32, 0, 1, 24, 16),
}

Q6_COLUMNS = ["lineitem.l_shipdate", "lineitem.l_discount", "lineitem.l_quantity", "lineitem.l_extendedprice"]
Q1_COLUMNS = Q6_COLUMNS + ["lineitem.l_tax", "lineitem.l_returnflag", "lineitem.l_linestatus"]
Q6_BYTES_PER_ROW = 28   # 4 + 8 + 8 + 8, SURVEY.md section 8(d)
Q1_BYTES_PER_ROW = 44


def col_id(name):
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _splitmix64(x):
    x = x + np.uint64(PHI)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def generate(spec, row0, n, seed=SEED):
    """Host (numpy) generation of rows [row0, row0+n) of one column."""
    with np.errstate(over="ignore"):
        rows = np.arange(row0, row0 + n, dtype=np.uint64)
        key = np.uint64(seed) ^ np.uint64((col_id(spec.name) * PHI) & _M64)
        h = _splitmix64(key ^ rows)
        span = np.uint64(spec.hi - spec.lo + 1)
        v = (np.uint64(spec.lo & _M64) + h % span) * np.uint64(spec.mul & _M64) + np.uint64(spec.add & _M64)
    return v.astype(np.int64).astype(spec.dtype)


def generate_table(names, row0, n, seed=SEED):
    return {name: generate(LINEITEM[name], row0, n, seed) for name in names}


# ---- TPC-H Q3 shape: customer / orders / lineitem with FK join-index columns -----------------------
# (value ranges /root/reference/tests/tpch10noorder/bounds.csv:38-79).  Join indices are row numbers of
# the referenced table ("oid = row number in dim table"); every order has
# exactly 4 lineitems here, so lineitem_orders = row // 4 is non-decreasing and any row range of
# lineitem can be produced independently.  Order keys follow the TPC-H pattern (8 of every 32 used).
CUSTOMER = {
    "customer.c_mktsegment": ColumnSpec("customer.c_mktsegment", np.int32, 1, 5, 16, 0),     # 'BUILDING' = 16 (dictionary.csv:74)
}
ORDERS = {
    "orders.o_orderdate": ColumnSpec("orders.o_orderdate", np.int32, 727563, 729968, 1, 0),
    "orders.o_shippriority": ColumnSpec("orders.o_shippriority", np.int32, 0, 0, 1, 0),
}
Q3_COLUMNS = ["lineitem.l_orderkey", "lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount",
              "lineitem.lineitem_orders", "lineitem.lineitem_l_orderkey_l_linenumber_pkey",
              "orders.orders_o_orderkey_pkey", "orders.o_orderdate", "orders.o_shippriority", "orders.orders_customer",
              "customer.customer_c_custkey_pkey", "customer.c_mktsegment"]


def orderkey_of_row(j):
    j = np.asarray(j, dtype=np.int64)
    return 1 + (j // 8) * 32 + (j % 8)


def q3_tables(n_orders, seed=SEED):
    """Host (numpy) Q3 catalog: n_orders orders, 4 lineitems each, n_orders // 10 customers."""
    n_cust = max(n_orders // 10, 1)
    n_li = 4 * n_orders
    t = {}
    t["customer.customer_c_custkey_pkey"] = np.zeros(n_cust, np.int64)                     # ref vector: length only
    t["customer.c_mktsegment"] = generate(CUSTOMER["customer.c_mktsegment"], 0, n_cust, seed)
    t["orders.orders_o_orderkey_pkey"] = np.zeros(n_orders, np.int64)
    for name in ORDERS:
        t[name] = generate(ORDERS[name], 0, n_orders, seed)
    t["orders.orders_customer"] = generate(ColumnSpec("orders.orders_customer", np.int64, 0, n_cust - 1, 1, 0), 0, n_orders, seed)
    li_orders = np.arange(n_li, dtype=np.int64) // 4
    t["lineitem.lineitem_orders"] = li_orders
    t["lineitem.lineitem_l_orderkey_l_linenumber_pkey"] = np.zeros(n_li, np.int64)
    t["lineitem.l_orderkey"] = orderkey_of_row(li_orders).astype(np.int32)
    for name in ("lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"):
        t[name] = generate(LINEITEM[name], 0, n_li, seed)
    return t


def register_q3_columns(engine, n_orders, li_rows=None, device="cuda", seed=SEED, copartition=False):
    """The Q3 catalog of `q3_tables` built in place on the GPU (counter-based generator + arithmetic join indices):
    customer / orders in full, lineitem rows [li_rows[0], li_rows[1]) (default: all 4 * n_orders).
    copartition=True keeps only the orders rows the lineitem shard references (lineitem is clustered by order, so
    that is one contiguous range) and rebases the join index to it: the co-located placement of a sharded star
    schema, under which no rank repeats the orders-side work of another.
    Returns the tensors that back the registered columns (keep them alive while the engine uses them)."""
    import torch

    n_cust, n_li = max(n_orders // 10, 1), 4 * n_orders
    r0, r1 = li_rows if li_rows is not None else (0, n_li)
    o0, o1 = (r0 // 4, max((r1 - 1) // 4 + 1, r0 // 4)) if copartition and r1 > r0 else (0, n_orders)
    keep = {}

    def reg(name, t):
        keep[name] = t
        engine.register_tensor(name, t)

    engine.generate(CUSTOMER["customer.c_mktsegment"], 0, n_cust, seed)
    for name in ORDERS:
        engine.generate(ORDERS[name], o0, o1 - o0, seed)
    engine.generate(ColumnSpec("orders.orders_customer", np.int64, 0, n_cust - 1, 1, 0), o0, o1 - o0, seed)
    for name in ("lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"):
        engine.generate(LINEITEM[name], r0, r1 - r0, seed)
    reg("customer.customer_c_custkey_pkey", torch.zeros(n_cust, dtype=torch.int64, device=device))
    reg("orders.orders_o_orderkey_pkey", torch.zeros(o1 - o0, dtype=torch.int64, device=device))
    reg("lineitem.lineitem_l_orderkey_l_linenumber_pkey", torch.zeros(r1 - r0, dtype=torch.int64, device=device))
    lo = torch.arange(r0, r1, dtype=torch.int64, device=device) // 4
    reg("lineitem.lineitem_orders", lo - o0)
    reg("lineitem.l_orderkey", (1 + (lo // 8) * 32 + (lo % 8)).to(torch.int32))
    torch.cuda.synchronize()
    return keep


# ---- TPC-H Q14 shape: lineitem with a join index into part, part.p_type as offsets into a string heap ---------------
# (p_type: 150 strings "<size> <finish> <metal>" in TPC-H, a fifth of them starting with PROMO; here 25 strings, 5 of them
# PROMO ..., at 8-byte aligned offsets of a MonetDB-style heap.)  lineitem_part is drawn with the counter-based generator,
# so any row range of lineitem can be produced independently.
Q14_COLUMNS = ["lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount", "lineitem.lineitem_part",
               "lineitem.lineitem_l_orderkey_l_linenumber_pkey", "part.part_p_partkey_pkey", "part.p_type", "part.p_type.heap"]
Q14_BYTES_PER_ROW = 4 + 8 + 8 + 8            # what the fused join scan reads of every lineitem row (the part side is looked up)
_P_TYPES = ["%s %s %s" % (a, b, c) for a in ("PROMO", "STANDARD", "SMALL", "MEDIUM", "ECONOMY") for b in ("ANODIZED", "BRUSHED", "PLATED", "POLISHED", "BURNISHED")
            for c in ("TIN",)]


def q14_part(n_part, seed=SEED):
    heap = bytearray(8)
    offs = []
    for s_ in _P_TYPES:
        while len(heap) % 8:
            heap.append(0)
        offs.append(len(heap))
        heap += s_.encode() + b"\0"
    pick = generate(ColumnSpec("part.p_type", np.int64, 0, len(offs) - 1, 1, 0), 0, n_part, seed)
    return {"part.part_p_partkey_pkey": np.zeros(n_part, np.int64), "part.p_type": np.asarray(offs, np.int64)[pick],
            "part.p_type.heap": np.frombuffer(bytes(heap), dtype=np.int8).copy()}


def q14_tables(n_li, seed=SEED):
    """Host (numpy) Q14 catalog: n_li lineitems, n_li // 30 parts."""
    n_part = max(n_li // 30, 1)
    t = q14_part(n_part, seed)
    for name in ("lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"):
        t[name] = generate(LINEITEM[name], 0, n_li, seed)
    t["lineitem.lineitem_part"] = generate(ColumnSpec("lineitem.lineitem_part", np.int64, 0, n_part - 1, 1, 0), 0, n_li, seed)
    t["lineitem.lineitem_l_orderkey_l_linenumber_pkey"] = np.zeros(n_li, np.int64)
    return t


def register_q14_columns(engine, n_li, li_rows=None, device="cuda", seed=SEED):
    """The Q14 catalog of `q14_tables` built on the GPU: part in full (uploaded: 1/30 of lineitem), lineitem rows
    [li_rows[0], li_rows[1]) generated in place.  Returns the tensors backing registered columns."""
    import torch

    n_part = max(n_li // 30, 1)
    r0, r1 = li_rows if li_rows is not None else (0, n_li)
    for name, v in q14_part(n_part, seed).items():
        engine.upload(name, v)
    for name in ("lineitem.l_shipdate", "lineitem.l_extendedprice", "lineitem.l_discount"):
        engine.generate(LINEITEM[name], r0, r1 - r0, seed)
    engine.generate(ColumnSpec("lineitem.lineitem_part", np.int64, 0, n_part - 1, 1, 0), r0, r1 - r0, seed)
    keep = {"lineitem.lineitem_l_orderkey_l_linenumber_pkey": torch.zeros(r1 - r0, dtype=torch.int64, device=device)}
    engine.register_tensor("lineitem.lineitem_l_orderkey_l_linenumber_pkey", keep["lineitem.lineitem_l_orderkey_l_linenumber_pkey"])
    torch.cuda.synchronize()
    return keep
