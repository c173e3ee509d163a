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


def synth_columns(meta_dir, cfg, vdl_text, scale=2e-4, seed=1):
    """{column key path: numpy array} for every Load of `vdl_text`."""
    codes = _per_column_codes(meta_dir)
    patterns = _like_patterns(vdl_text)
    info = {name: ci for name, ci in cfg.colinfo.to_list()}
    rows = {}
    for (table, _), ci in info.items():
        rows[table] = scaled_rows(ci.count, scale)
    fk_dim = {fk.idxname: fk.dim[0] for fk in cfg.fkrefs.values()}
    pk_names = set(cfg.table_pkeys.values())
    wanted = [ln.split(",")[2].split(";;")[0].strip() for ln in vdl_text.splitlines() if len(ln.split(",")) >= 3 and ln.split(",")[1] == "Load"]
    heaps, out = {}, {}

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
            out[path] = _rng(seed, path).integers(0, rows[fk_dim[key]], n).astype(np.int64)
            continue
        if key not in info:
            raise FrontendError("no column %s in the catalog" % path)
        ci = info[key]
        rng = _rng(seed, path)
        if ci.dtype[0][0] == "DString":
            _, where = string_column(table, parts[1])
            offs = np.array(sorted(where.values()), dtype=np.int64)
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
