"""Synthetic catalogs for machine-generated VDL programs.

The reference ships catalog *metadata* only (bounds.csv, storage.csv, dictionary.csv,
schema.msqldump under /root/reference/tests/tpch10noorder; readers in frontend/__init__.py after
Config.hs:57-79).  `synth_columns` produces, for every column a VDL program Loads, data that respects
that metadata at a reduced row count, so that any of the TPC-H plans the front end compiles can be run
end to end:

* ordinary columns: uniform in the catalog's [min, max] at the column's storage width and trailing-zero
  alignment (Config.hs:114-120); dictionary-coded columns draw from the codes dictionary.csv lists for
  them (so equality predicates on string literals select something);
* primary-key reference vectors (`table.<pk constraint>`, Vlite.hs:734-741): zeros, length only;
* foreign-key join indices (`table.<fk constraint>`, Vlite.hs:1250-1258): row numbers of the scaled dim table;
* the database is COHERENT: primary-key value columns are unique (ascending, spread over the catalog's range), every
  foreign-key value column holds the key of the row its join index points at (l_orderkey = o_orderkey[lineitem_orders]),
  lineitem's part / supplier indices go through its partsupp index, and (ps_partkey, ps_suppkey) pairs are unique --
  so plans that join on VALUES (Q15's view, Q18's IN-subquery, Q20's correlated subquery) select the same rows as
  plans that join through the indices, and the SQL text of a query can be evaluated over the columns directly
  (tests/sql_eval.py);
* string heaps (`table.col.heap`, Vdl.hs:246): NUL-terminated strings at 8-byte aligned offsets; the
  vocabulary contains a match and a near-miss for every LIKE pattern the program applies to that heap.
"""
import csv
import os
import re
import zlib

import numpy as np

from .frontend.config import FrontendError

_WORDS = ("almond antique aquamarine azure beige bisque black blanched blue blush brown burlywood burnished chartreuse chiffon "
          "chocolate coral cornflower cornsilk cream cyan dark deep dim dodger drab firebrick floral frosted gainsboro ghost "
          "goldenrod honeydew hot indian ivory khaki lace lavender lawn lemon light lime linen magenta maroon medium").split()


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def scaled_rows(count, scale):
    return int(count) if count <= 64 else max(int(count * scale), 16)


def _per_column_codes(meta_dir):
    codes = {}
    with open(os.path.join(meta_dir, "dictionary.csv"), newline="") as fh:
        for row in csv.reader(fh):
            if len(row) >= 4:
                codes.setdefault((row[0], row[1]), {})[row[2]] = int(row[3])
    return codes


def _like_patterns(vdl_text):
    """{heap column key path: [patterns]} from the Like statements of a program."""
    loads, alias, out = {}, {}, {}
    for line in vdl_text.splitlines():
        line = line.split(";;")[0].strip()
        f = line.split(",")
        if len(f) < 3:
            continue
        if f[1] == "Load":
            loads[f[0]] = f[2]
        elif f[1] == "Project" and len(f) >= 4:
            alias[f[0]] = f[3].replace("Id ", "").strip()
        elif f[1] == "Like" and len(f) >= 8:
            h = f[5].replace("Id ", "").strip()
            while h in alias:
                h = alias[h]
            if h in loads:
                out.setdefault(loads[h], []).append(",".join(f[7:]))
    return out


def _vocabulary(patterns, rng, n_fill):
    vocab = []
    for p in patterns:
        vocab.append(p.replace("%", " xx ").replace("_", "q").strip() or "xx")        # matches
        vocab.append(p.replace("%", "").replace("_", "q")[:-1] + "#")                  # near miss
    for _ in range(n_fill):
        vocab.append(" ".join(_WORDS[k] for k in rng.integers(0, len(_WORDS), int(rng.integers(1, 5)))))
    seen, out = set(), []
    for s in vocab:
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def _build_heap(placed, free, base, limit=None):
    """placed: {string: fixed offset}; free: strings to put at the next 8-byte aligned free offsets >= base,
    as long as the offset stays <= limit (the column's upper bound in bounds.csv)."""
    where = dict(placed)
    end = max([o + len(s.encode()) + 1 for s, o in placed.items()] + [base])
    buf = bytearray(end)
    spans = sorted((o, o + len(s.encode()) + 1) for s, o in placed.items())
    for (a0, a1), (b0, _) in zip(spans, spans[1:]):
        if a1 > b0:
            raise FrontendError("dictionary codes overlap in the heap")
    for s, o in placed.items():
        buf[o:o + len(s.encode())] = s.encode()
    for s in free:
        if s in where:
            continue
        while len(buf) % 8:
            buf.append(0)
        if limit is not None and len(buf) > limit and where:
            break
        where[s] = len(buf)
        buf += s.encode() + b"\0"
    return np.frombuffer(bytes(buf), dtype=np.int8).copy(), where


def _fk_graph(cfg):
    """{(fact table, index column): (dim table, [(fact value column, dim key column), ...])} from the catalog's foreign keys."""
    out = {}
    for fk in cfg.fkrefs.values():
        if fk.fkjoinorder != "FactDim" or fk.cols[0][1][1] == "%TID%":
            continue
        out[(fk.fact[0], fk.idxname[1])] = (fk.dim[0], [(a[1], b[1]) for a, b in fk.cols])
    return out


def synth_columns(meta_dir, cfg, vdl_text, scale=2e-4, seed=1, extra=(), clustered=()):
    """{column key path: numpy array} for every Load of `vdl_text` (and every path in `extra`).
    clustered: join-index columns (e.g. "lineitem.lineitem_orders") whose rows come in the dimension's order, as dbgen writes
    lineitem clustered by order -- the same random draw, sorted: keys derived through it (l_orderkey) are then non-decreasing."""
    codes = _per_column_codes(meta_dir)
    patterns = _like_patterns(vdl_text)
    info = {name: ci for name, ci in cfg.colinfo.to_list()}
    rows = {}
    for (table, _), ci in info.items():
        rows[table] = scaled_rows(ci.count, scale)
    fk_dim = {fk.idxname: fk.dim[0] for fk in cfg.fkrefs.values()}
    pk_names = set(cfg.table_pkeys.values())
    wanted = [ln.split(",")[2].split(";;")[0].strip() for ln in vdl_text.splitlines() if len(ln.split(",")) >= 3 and ln.split(",")[1] == "Load"]
    wanted += list(extra)
    heaps, out = {}, {}
    fks = _fk_graph(cfg)
    single = {k: v for k, v in fks.items() if len(v[1]) == 1}
    composite = {k: v for k, v in fks.items() if len(v[1]) > 1}
    pk_value = set()                               # (table, column): single-column primary keys some foreign key refers to
    fk_value = {}                                  # (fact table, value column) -> (index column, dim table, dim key column)
    for (fact, idx), (dim, pairs) in single.items():
        pk_value.add((dim, pairs[0][1]))
        fk_value[(fact, pairs[0][0])] = (idx, dim, pairs[0][1])
    # a fact table with a composite foreign key (lineitem -> partsupp on (partkey, suppkey)) reaches part / supplier THROUGH
    # that row: {(fact, single index): (composite index, middle table, the middle table's own index to the same dim)}
    via = {}
    pair_tables = set()                            # tables whose rows are unique pairs of two foreign keys (partsupp)
    for (fact, cidx), (mid, pairs) in composite.items():
        pair_tables.add(mid)
        for fcol, mcol in pairs:
            for (f2, i2), (d2, p2) in single.items():
                if f2 == fact and p2[0][0] == fcol:
                    hop = [i3 for (f3, i3), (d3, p3) in single.items() if f3 == mid and p3[0][0] == mcol and d3 == d2]
                    if hop:
                        via[(fact, i2)] = (cidx, mid, hop[0])
    memo = {}

    def join_index(table, idx):
        key = (table, idx)
        if key in memo:
            return memo[key]
        n = rows[table]
        dim = fks[key][0]
        if key in via:                                                  # lineitem_part = partsupp_part[lineitem_partsupp]
            cidx, mid, hop = via[key]
            v = join_index(mid, hop)[join_index(table, cidx)]
        elif table in pair_tables and key in single:                    # unique (part, supplier) pairs
            mine = sorted(i for (f, i) in single if f == table)
            i = np.arange(n, dtype=np.int64)
            n0, n1 = rows[single[(table, mine[0])][0]], rows[single[(table, mine[1])][0]]
            a = i % n0
            v = a if idx == mine[0] else (i // n0 + a * 7) % n1
        else:
            v = _rng(seed, "%s.%s" % key).integers(0, rows[dim], n)
            if "%s.%s" % key in clustered:
                v = np.sort(v)
        memo[key] = np.asarray(v, dtype=np.int64)
        return memo[key]

    def key_values(table, col):
        ci = info[(table, col)]
        lo, hi, n = int(ci.bounds[0]), int(ci.bounds[1]), rows[table]
        return lo + np.arange(n, dtype=np.int64) * max((hi - lo) // max(n, 1), 1)

    def string_column(table, col):
        key = (table, col)
        if key not in heaps:
            rng = _rng(seed, "%s.%s.heap" % key)
            ci = info[key]
            fixed = codes.get(key, {})
            free = [] if fixed and not patterns.get("%s.%s.heap" % key) else _vocabulary(patterns.get("%s.%s.heap" % key, []), rng, 24)
            if not fixed and int(ci.bounds[1]) - int(ci.bounds[0]) < 4096:        # a narrow code range: short strings so several fit
                free = [w[:6] for w in free]
            heaps[key] = _build_heap(fixed, free, max(int(ci.bounds[0]), 8) if not fixed else 8, int(ci.bounds[1]))
        return heaps[key]

    for path in wanted:
        if path in out:
            continue
        parts = path.split(".")
        table = parts[0]
        if table not in rows:
            raise FrontendError("no table %s in the catalog" % table)
        n = rows[table]
        if len(parts) == 3 and parts[2] == "heap":
            out[path] = string_column(table, parts[1])[0]
            continue
        key = (table, parts[1])
        if key in pk_names:
            out[path] = np.zeros(n, np.int64)
            continue
        if key in fk_dim:
            out[path] = join_index(table, parts[1])
            continue
        if key not in info:
            raise FrontendError("no column %s in the catalog" % path)
        ci = info[key]
        rng = _rng(seed, path)
        if key in fk_value:                                              # the key of the row the join index points at
            idx, dim, dcol = fk_value[key]
            out[path] = key_values(dim, dcol)[join_index(table, idx)].astype(np.int32 if ci.stype == ("SInt32",) else np.int64)
            continue
        if key in pk_value:
            out[path] = key_values(table, parts[1]).astype(np.int32 if ci.stype == ("SInt32",) else np.int64)
            continue
        if ci.dtype[0][0] == "DString":
            _, where = string_column(table, parts[1])
            offs = np.array(sorted(where.values()), dtype=np.int64)
            if n <= len(offs) and codes.get(key):                        # nation / region: every name once, as in the real tables
                out[path] = offs[rng.permutation(len(offs))[:n]]
            else:
                out[path] = offs[rng.integers(0, len(offs), n)]
            continue
        lo, hi, tz = int(ci.bounds[0]), int(ci.bounds[1]), int(ci.trailing_zeros)
        if tz >= 63 or hi < lo:
            vals = np.zeros(n, np.int64)
        else:
            a, b = lo >> tz, hi >> tz
            vals = (rng.integers(a, b + 1, n, dtype=np.int64)) << tz
        out[path] = vals.astype(np.int32 if ci.stype == ("SInt32",) else np.int64)
    return out


def export_columns(cols, directory):
    """Writes {key path: array} as raw little-endian `<key path>.bin` files plus `columns.csv`
    (name, bytes per element, rows): the layout `vdlrun --data DIR` and `load_columns` read."""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "columns.csv"), "w") as fh:
        for name in sorted(cols):
            v = np.ascontiguousarray(cols[name])
            v.astype(v.dtype.newbyteorder("<"), copy=False).tofile(os.path.join(directory, name + ".bin"))
            fh.write("%s,%d,%d\n" % (name, v.dtype.itemsize, v.shape[0]))


def load_columns(directory, names=None):
    """{key path: array} from a directory written by export_columns (memory-mapped)."""
    out = {}
    with open(os.path.join(directory, "columns.csv")) as fh:
        for line in fh:
            name, width, rows = line.strip().split(",")
            if names is not None and name not in names:
                continue
            dt = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[int(width)]
            out[name] = np.memmap(os.path.join(directory, name + ".bin"), dtype=dt, mode="r", shape=(int(rows),)) if int(rows) else np.zeros(0, dt)
    return out


def scaled_config(cfg, scale):
    """A copy of the catalog whose row counts are those of the scaled tables `synth_columns` generates.
    Needed for the VLite output format, which prints table lengths into the program (reference vectors
    become `RangeC 0 <rows> 1`, Vlite.hs:740) instead of loading a primary-key column for its length."""
    import copy

    from .frontend.config import NameTable

    out = copy.copy(cfg)
    out.colinfo = NameTable.from_list([(name, ci._replace(count=scaled_rows(ci.count, scale))) for name, ci in cfg.colinfo.to_list()])
    return out


def tpch_scaled_config(cfg, factor):
    """The catalog at `factor` times its scale factor (the metadata under tests/golden/tpch10noorder describes SF10;
    factor 10 = SF100): row counts of every table but nation / region, and the upper bounds of surrogate keys and
    join indices, grow with the scale factor (TPC-H specification 4.2.5); everything else keeps its range.  The
    compiled program depends on these bounds (group-key widths, Partition pivots), so a program compiled for SF10
    must not be run over SF100 keys."""
    import copy

    from .frontend.config import NameTable

    def grow(name, ci):
        table, col = name[0], name[-1]
        count = ci.count if ci.count <= 64 else int(ci.count * factor)
        lo, hi = ci.bounds
        is_key = col.endswith("key") or col.startswith("%")
        if is_key and ci.trailing_zeros < 63 and hi > lo and ci.count > 64:
            hi = int(hi * factor)
        return ci._replace(count=count, bounds=(lo, hi))

    out = copy.copy(cfg)
    out.colinfo = NameTable.from_list([(name, grow(name, ci)) for name, ci in cfg.colinfo.to_list()])
    return out
