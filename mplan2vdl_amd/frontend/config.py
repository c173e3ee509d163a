"""Catalog model of the mplan -> VDL front end: names, types, column info, PK/FK maps.

Python restatement of the reference's Name.hs (suffix-matching name table), Types.hs (storage /
display / Monet types) and Config.hs (CSV records, ColInfo, key constraints).  Citations are
/root/reference/src/<file>:<lines>.
"""
import csv
from collections import namedtuple

INT64_MIN, INT64_MAX = -(1 << 63), (1 << 63) - 1
INT32_MIN, INT32_MAX = -(1 << 31), (1 << 31) - 1


class FrontendError(Exception):
    pass


def show_name(name):
    """Name.hs:53-54: dotted form."""
    return ".".join(name)


class NameTable:
    """Name.hs:65-126.  Fully qualified names are inserted; lookups may use any unambiguous suffix."""

    def __init__(self):
        self.m = {}            # reversed tuple -> value

    def insert(self, name, val):
        k = tuple(reversed(name))
        if k in self.m:
            raise FrontendError("Scope already has %s" % show_name(name))
        self.m[k] = val

    def insert_weak(self, name, val):
        self.m[tuple(reversed(name))] = val

    def lookup(self, name):
        """Returns (full_name, value) or raises; Name.hs:96-113."""
        r = tuple(reversed(name))
        hits = [k for k in self.m if k[: len(r)] == r]
        if not hits:
            raise FrontendError("no name: %s in scope: %s" % (show_name(name), sorted(show_name(tuple(reversed(k))) for k in self.m)))
        if len(hits) > 1:
            raise FrontendError("Ambiguous name resolution for %s: %s" % (show_name(name), [show_name(tuple(reversed(k))) for k in sorted(hits)]))
        k = hits[0]
        return tuple(reversed(k)), self.m[k]

    def contains(self, name):
        try:
            self.lookup(name)
            return True
        except FrontendError:
            return False

    def to_list(self):
        """Map.toList order = ascending reversed key (Name.hs:79-82)."""
        return [(tuple(reversed(k)), self.m[k]) for k in sorted(self.m)]

    @staticmethod
    def from_list(pairs):
        t = NameTable()
        for n, v in pairs:
            t.insert_weak(n, v)
        return t


# ---- Types.hs ------------------------------------------------------------------------------
# SType: ("SDecimal", precision, scale) | ("SInt32",) | ("SInt64",)
# DType: ("DDecimal", point) | ("DString", decoder_name) | ("DDate",)
# MType: ("MInt",), ("MChar", n), ("MDecimal", p, s), ...
S_INT32, S_INT64 = ("SInt32",), ("SInt64",)
D_DATE = ("DDate",)


def d_decimal(point):
    return ("DDecimal", point)


def resolve_typespec(tname, tparams):
    """Types.hs:154-173 (case-insensitive type names from plans and from the schema dump)."""
    t = tname.lower()
    p = list(tparams)
    if t in ("int", "integer") and not p: return ("MInt",)
    if t == "tinyint" and not p: return ("MTinyint",)
    if t == "smallint" and not p: return ("MSmallint",)
    if t == "bigint" and not p: return ("MBigInt",)
    if t == "date" and not p: return ("MDate",)
    if t == "char" and len(p) == 1: return ("MChar", p[0])
    if t == "char" and not p: return ("MChar", -1)
    if t == "varchar" and len(p) == 1: return ("MVarchar", p[0])
    if t == "decimal" and len(p) == 2: return ("MDecimal", p[0], p[1])
    if t == "sec_interval" and len(p) == 1: return ("MMillisec",)
    if t == "month_interval" and not p: return ("MMonth",)
    if t == "double" and not p: return ("MDouble",)
    if t == "boolean" and not p: return ("MBoolean",)
    if t == "oid" and not p: return ("MOid",)
    raise FrontendError("unsupported typespec: %s%s" % (tname, p))


def stype_of_mtype(m):
    """Types.hs:127-138."""
    k = m[0]
    if k in ("MInt", "MDate", "MSmallint", "MTinyint"): return S_INT32
    if k in ("MOid", "MChar", "MVarchar", "MBigInt"): return S_INT64
    if k == "MDecimal": return ("SDecimal", m[1], m[2])
    raise FrontendError("we don't expect reading this type from the monet columns/queries: %s" % (m,))


def dtype_of_mtype(m, name):
    """Types.hs:140-151."""
    k = m[0]
    if k in ("MInt", "MSmallint", "MTinyint", "MBigInt", "MOid"): return d_decimal(0)
    if k == "MDecimal": return d_decimal(m[2])
    if k == "MDate": return D_DATE
    if k in ("MChar", "MVarchar"): return ("DString", name)
    raise FrontendError("not handling this type on its own: %s" % (m,))


def bounds_of(stype):
    return (INT32_MIN, INT32_MAX) if stype == S_INT32 else (INT64_MIN, INT64_MAX)


# ---- Config.hs -----------------------------------------------------------------------------
# ColInfo (Config.hs:114-120); dtype is the pair (DType, note)
ColInfo = namedtuple("ColInfo", "bounds trailing_zeros count stype dtype")

FKInstance = namedtuple("FKInstance", "cols fkjoinorder fact dim idxname")     # Config.hs:198
Table = namedtuple("Table", "name columns pkey fkeys")                        # SchemaParser.y:131-139
PKey = namedtuple("PKey", "pkcols pkconstraint")
FKey = namedtuple("FKey", "references colmap fkconstraint")


class Config:
    """Config.hs:223-238 plus the flags of MainFuns.hs:34-75 that shape the emitted program."""

    def __init__(self):
        self.cross_product = False
        self.sparsity_threshold = 1.0
        self.aggregation_strategy = ("AggSerial",)      # | ("AggHierarchical", lg_grain) | ("AggShuffle",)
        self.show_metadata = False
        self.gboffset = 0
        self.format = "vdl"                              # "vdl" | "vlite"
        self.dictionary = {}
        self.colinfo = NameTable()
        self.fkrefs = {}          # sorted tuple of (fact col, dim col) pairs -> FKInstance
        self.pkeys = {}           # sorted tuple of pk column names -> pk constraint name
        self.table_pkeys = {}     # table name -> its pk constraint (qualified)
        self.partialfks = {}      # (col a, col b) -> (join order, full sorted pair list)
        self.partialpks = {}      # pk column -> full sorted pk column tuple

    def is_pkey(self, cols):
        return self.pkeys.get(tuple(sorted(cols)))

    def lookup_pkey(self, table):
        if table not in self.table_pkeys:
            raise FrontendError("every table has a pkey. no info for table loaded?")
        return self.table_pkeys[table]

    def is_fkref(self, cols):
        return self.fkrefs.get(tuple(sorted(cols)))


def _read_csv(path):
    with open(path, newline="") as f:
        return [row for row in csv.reader(f) if row]


def make_config(bounds_rows, storage_rows, tables, dict_rows, **flags):
    """Config.hs:149-170."""
    cfg = Config()
    for k, v in flags.items():
        if not hasattr(cfg, k):
            raise FrontendError("unknown option %s" % k)
        setattr(cfg, k, v)
    for row in dict_rows:                                   # makeDictionary, Config.hs:82-85: keyed by the string only
        cfg.dictionary[row[2]] = int(row[3])
    constraints = []
    tspecs = NameTable()
    for t in tables:
        constraints.append(t.name + t.pkey.pkconstraint)
        constraints += [t.name + fk.fkconstraint for fk in t.fkeys]
        for cname, ts in t.columns:
            tspecs.insert_weak(t.name + cname, ts)
    # storage: (mtype, storagesize) per table.column, Config.hs:88-105
    storagemap = NameTable()
    for r in storage_rows:
        tab, col, typstring, count, colsize = r[1], r[2], r[3], int(r[5]), int(r[7])
        name = (tab, col)
        if typstring != "oid":
            if not tspecs.contains(name):
                continue
            ts = tspecs.lookup(name)[1]
        else:
            ts = ("oid", ())
        storagemap.insert_weak(name, resolve_typespec(ts[0], ts[1]))
    for r in bounds_rows:                                   # addEntry, Config.hs:136-147
        tab, col = r[0], r[1]
        lo, hi, count, tz = int(r[2]), int(r[3]), int(r[4]), int(r[5])
        name = (tab, col)
        mtype = storagemap.lookup(name)[1]
        info = ColInfo((lo, hi), tz, count, stype_of_mtype(mtype), (dtype_of_mtype(mtype, name), "from storage file"))
        cfg.colinfo.insert(name, info)
        if name in constraints:
            cfg.colinfo.insert((tab, "%" + col), info)      # constraints get marked with % as well
    allrefs = []
    for t in tables:                                        # makeFKEntries, Config.hs:200-218
        for fk in t.fkeys:
            local = [t.name + a for a, _ in fk.colmap]
            remote = [fk.references + b for _, b in fk.colmap]
            joinidx = t.name + fk.fkconstraint
            implicit = tuple(sorted(zip(local, remote)))
            implicit_back = tuple(sorted(zip(remote, local)))
            tid = fk.references + ("%TID%",)
            allrefs += [FKInstance(implicit, "FactDim", t.name, fk.references, joinidx),
                        FKInstance(implicit_back, "DimFact", t.name, fk.references, joinidx),
                        FKInstance(((joinidx, tid),), "FactDim", t.name, fk.references, joinidx),
                        FKInstance(((tid, joinidx),), "DimFact", t.name, fk.references, joinidx)]
    for inst in allrefs:
        cfg.fkrefs[inst.cols] = inst
        for pair in inst.cols:                              # make_partials, Config.hs:161-162
            straight = inst.cols if inst.fkjoinorder == "FactDim" else tuple((b, a) for a, b in inst.cols)
            cfg.partialfks[pair] = (inst.fkjoinorder, straight)
    for t in tables:
        pkl = tuple(sorted(t.name + c for c in t.pkey.pkcols))
        cons = t.name + t.pkey.pkconstraint
        cfg.pkeys[pkl] = cons
        for c in pkl:
            cfg.partialpks[c] = pkl
        cfg.table_pkeys[t.name] = cons
    return cfg


def load_config(bounds_csv, storage_csv, schema_dump, dictionary_csv, **flags):
    """What tpchrun wires up: -b bounds.csv -s schema.msqldump -t storage.csv --dictionary dictionary.csv
    (/root/reference/tpchrun:4)."""
    from .parse import parse_schema, read_commented

    tables = parse_schema(read_commented(schema_dump))
    return make_config(_read_csv(bounds_csv), _read_csv(storage_csv), tables, _read_csv(dictionary_csv), **flags)
