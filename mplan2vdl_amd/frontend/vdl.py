"""VDL emitter of the front end: Vexp DAG -> Voodoo operators -> numbered SSA text.

Restates /root/reference/src/Vdl.hs: operator lowering incl. the `<`, `<=`, `>=`, `!=`, min/max and
`?:` sugar (:136-157,209-231), the Vexp -> Voodoo memo (:171-179), output renaming and the
MaterializeCompact wrap (:271-292), post-order numbering with a CSE table whose key includes the
printed metadata (:294-369 -- which is why structurally equal folds can appear twice), and the line
printers (:410-481).
"""
from .config import FrontendError, show_name


class _Interner:
    """Voodoo nodes are identified by operator + operand identities + metadata (Vdl.hs:56-70)."""

    def __init__(self):
        self.ids, self.nodes = {}, []

    def get(self, vd, meta):
        key = (vd, meta)
        if key not in self.ids:
            self.ids[key] = len(self.nodes)
            self.nodes.append(key)
        return self.ids[key]


def _meta_of(v):
    """Vdl.hs:82-92."""
    origin = v.lineage[0] if v.lineage is not None else None
    return (v.info.bounds, v.info.count, v.name, v.info.dtype[0], origin, v.info.dtype[1] + " " + v.comment)


class Emitter:
    def __init__(self, config):
        self.config = config
        self.I = _Interner()
        self.memo = {}            # Vexp structural key -> voodoo node id

    # -- convenience constructors (Vdl.hs:104-157); all carry no metadata
    def n(self, *vd): return self.I.get(vd, None)
    def const_(self, k, v): return self.n("RangeV", k, 0, v)
    def pos_(self, v): return self.n("RangeV", 0, 1, v)
    def bin(self, op, a, b): return self.n("Binary", op, a, b)
    def gt(self, a, b): return self.bin("Greater", a, b)
    def lt(self, a, b): return self.bin("Greater", b, a)          # notice the argument swap
    def eq(self, a, b): return self.bin("Equals", a, b)
    def leq(self, a, b): return self.bin("LogicalOr", self.lt(a, b), self.eq(a, b))
    def geq(self, a, b): return self.bin("LogicalOr", self.gt(a, b), self.eq(a, b))

    def cond(self, c, a, b):
        negcond = self.eq(c, self.const_(0, a))
        return self.bin("Add", self.bin("Multiply", self.bin("Subtract", self.const_(1, a), negcond), a),
                        self.bin("Multiply", negcond, b))

    def makeload(self, name):
        if len(name) < 2:
            raise FrontendError("need longer keypath to be consistent with ./Driver keypaths")
        return ("Project", ("val",), name[1:], self.I.get(("Load", name), None))

    def from_vexp(self, v):
        """Vdl.hs:171-179: memoised on the Vexp's structure; the FIRST Vexp seen supplies the metadata."""
        if v.key in self.memo:
            return self.memo[v.key]
        nid = self.I.get(self._lower(v.vx), _meta_of(v))
        self.memo[v.key] = nid
        return nid

    def _lower(self, vx):
        k = vx[0]
        if k == "Load":
            return self.makeload(vx[1])
        if k == "RangeV":
            return ("RangeV", vx[1], vx[2], self.from_vexp(vx[3]))
        if k == "RangeC":
            return ("RangeC", vx[1], vx[2], vx[3])
        if k == "CrossProduct":
            l = self.from_vexp(vx[1])
            return ("CrossProduct", l, self.from_vexp(vx[2]), vx[3])
        if k == "Binop":
            l = self.from_vexp(vx[2])
            r = self.from_vexp(vx[3])
            op = vx[1]
            table = {"Gt": self.gt, "Eq": self.eq, "Lt": self.lt, "Leq": self.leq, "Geq": self.geq}
            simple = {"Mul": "Multiply", "Sub": "Subtract", "Add": "Add", "LogAnd": "LogicalAnd", "LogOr": "LogicalOr",
                      "Div": "Divide", "BitShift": "BitShift", "BitOr": "BitwiseOr", "BitAnd": "BitwiseAnd", "Mod": "Modulo"}
            if op in table: nid = table[op](l, r)
            elif op in simple: nid = self.bin(simple[op], l, r)
            elif op == "Min": nid = self.cond(self.leq(l, r), l, r)
            elif op == "Max": nid = self.cond(self.geq(l, r), l, r)
            elif op == "Neq": nid = self.bin("Subtract", self.const_(1, l), self.eq(l, r))
            else: raise FrontendError("binop %s" % op)
            return self.I.nodes[nid][0]                   # the operator part; metadata is attached by from_vexp
        if k == "Shuffle":
            src = self.from_vexp(vx[2])
            pos = self.from_vexp(vx[3])
            if vx[1] == "Gather":
                return ("Binary", "Gather", src, pos)
            svd = self.I.nodes[src][0]
            fold = src if (svd[0] == "RangeV" and svd[1] == 0 and svd[2] == 1) else self.pos_(src)
            return ("Scatter", src, fold, pos)
        if k == "Like":
            d = self.from_vexp(vx[1])
            ldict = self.I.get(self.makeload(tuple(vx[3]) + ("heap",)), None)
            return ("Like", d, ldict, vx[2])
        if k == "VShuffle":
            return ("VShuffle", self.from_vexp(vx[1]))
        if k == "Semisort":                               # Vdl.hs:205-207
            return ("Semisort", self.from_vexp(vx[1]))
        if k == "Fold":
            g = self.from_vexp(vx[2])
            d = self.from_vexp(vx[3])
            op = {"FChoose": "FoldChoose", "FSum": "FoldSum", "FMax": "FoldMax", "FMin": "FoldMin", "FSel": "FoldSelect"}[vx[1]]
            return ("Binary", op, g, d)
        if k == "Partition":
            data = self.from_vexp(vx[2])
            piv = self.from_vexp(vx[1])
            return ("Binary", "Partition", data, piv)
        raise FrontendError("lowering of %s" % k)

    def outputs(self, vexps):
        """Vdl.hs:271-292.  The fold conses, so outputs come out in reverse list order."""
        ans = []
        for v in vexps:
            meta = _meta_of(v)
            nid_memo = self.from_vexp(v)
            nid = self.I.get(self.I.nodes[nid_memo][0], meta)            # the output keeps ITS metadata
            ans.insert(0, (nid, meta, v))
        outs = []
        for nid, meta, v in ans:
            if self.config.format == "vdl":
                name, origin = meta[2], meta[4]
                if name is not None and origin is not None: catted = (name[-1],) + tuple(origin)
                elif name is not None: catted = (name[-1],)
                elif origin is not None: catted = ("val",) + tuple(origin)
                else: catted = ("val",)
                newname = show_name(catted).replace(".", "__")
                nid = self.I.get(("Project", (newname,), ("val",), nid), meta[:5] + ("rename for output",))
            outs.append(self.I.get(("MaterializeCompact", nid), meta))
        return outs

    # -- numbering + printing (Vdl.hs:294-481)
    def children(self, vd):
        k = vd[0]
        if k in ("Load", "RangeC"): return []
        if k == "Project": return [vd[3]]
        if k == "RangeV": return [vd[3]]
        if k == "Binary": return [vd[2], vd[3]]
        if k == "Scatter": return [vd[1], vd[2], vd[3]]
        if k in ("CrossProduct", "Like"): return [vd[1], vd[2]]
        if k in ("VShuffle", "MaterializeCompact", "Semisort"): return [vd[1]]
        raise FrontendError("children of %s" % k)

    def number(self, outs):
        ids, log = {}, []

        def visit(nid):
            if nid in ids:
                return ids[nid]
            vd, meta = self.I.nodes[nid]
            refs = [visit(c) for c in self.children(vd)]
            ids[nid] = len(log) + 1
            log.append((ids[nid], vd, refs, meta))
            return ids[nid]

        for o in outs:
            visit(o)
        return log

    def fields(self, vd, refs):
        k = vd[0]
        r = ["Id %d" % x for x in refs]
        if k == "Load":
            nm = vd[1][1:] if vd[1][0] == "sys" else vd[1]
            return ["Load", show_name(nm)]
        if k == "Project": return ["Project", show_name(vd[1]), r[0], show_name(vd[2])]
        if k == "RangeV": return ["RangeV", "val", str(vd[1]), r[0], str(vd[2])]
        if k == "RangeC": return ["RangeC", "val", str(vd[1]), str(vd[3]), str(vd[2])]
        if k == "Binary":
            if vd[1] == "Gather": return ["Gather", r[0], r[1], "val"]
            return [vd[1], "val", r[0], "val", r[1], "val"]
        if k == "Scatter": return ["Scatter", r[0], r[1], "val", r[2], "val"]
        if k == "Like": return ["Like", "val", r[0], "val", r[1], "val", vd[3]]
        if k == "VShuffle": return ["Shuffle", r[0]]
        if k == "MaterializeCompact": return ["MaterializeCompact", r[0]]
        if k == "CrossProduct": return ["CrossProductOuter" if vd[3] == "COuter" else "CrossProductInner", r[0], r[1]]
        if k == "Semisort": return ["Semisort", r[0]]
        raise FrontendError("printing of %s" % k)

    def vlite_fields(self, vd, refs):
        """Vdl.hs:370-408 (toVList): "lighter syntax (one value per vector)", MainFuns.hs:70 -- no field names."""
        k = vd[0]
        r = ["Id %d" % x for x in refs]
        if k == "Load": return ["Load", show_name(vd[1])]
        if k == "Project": return ["Project", r[0]]
        if k == "RangeV": return ["RangeV", str(vd[1]), r[0], str(vd[2])]
        if k == "RangeC": return ["RangeC", str(vd[1]), str(vd[3]), str(vd[2])]
        if k == "Semisort": return ["Semisort", r[0]]
        if k == "Binary": return [vd[1], r[0], r[1]]
        if k == "Scatter": return ["Scatter", r[0], r[1], r[2]]
        if k == "Like": return ["Like", r[0], r[1], vd[3]]
        if k == "VShuffle": return ["Shuffle", r[0]]
        if k == "MaterializeCompact": return ["Output", r[0]]
        if k == "CrossProduct": return ["CrossProductOuter" if vd[3] == "COuter" else "CrossProductInner", r[0], r[1]]
        raise FrontendError("printing of %s" % k)


def _show_meta(meta):
    if meta is None:
        return ""
    bounds, count, name, dtype, origin, comment = meta
    return " ;; Metadata {databounds = (%d,%d), sizebound = %d, name = %s, displaytype = %s, origin = %s, comment = %r}" % (
        bounds[0], bounds[1], count, "Just " + show_name(name) if name is not None else "Nothing", dtype,
        "Just " + show_name(origin) if origin is not None else "Nothing", comment)


def vdl_from_vexps(vexps, config):
    """Vdl.hs:490-495: the program text, one statement per line."""
    em = Emitter(config)
    log = em.number(em.outputs(vexps))
    lines = []
    for ident, vd, refs, meta in log:
        if config.format == "vlite":
            # Vdl.hs:455-475 (printLine): named outputs print as "<name>,Output,<display type>,Id x"
            strs = em.vlite_fields(vd, refs)
            if vd[0] == "MaterializeCompact" and meta is not None and meta[2] is not None:
                dt = meta[3]
                typ = "decimal_%d" % dt[1] if dt[0] == "DDecimal" else "string_%s" % show_name(dt[1]) if dt[0] == "DString" else "date"
                fstrs = [meta[2][-1], "Output", typ] + strs[1:]
            else:
                fstrs = [str(ident)] + strs
            s = ",".join(fstrs)
        else:
            s = ",".join([str(ident)] + em.fields(vd, refs))
        if config.show_metadata:
            s += _show_meta(meta)
        lines.append(s)
    return "\n".join(lines)
