"""Vector IR of the front end: RelExpr -> DAG of vector expressions (Vexp) + the three cleanup passes.

Restates /root/reference/src/Vlite.hs.  A Vexp carries the operator (vx), inferred column metadata
(bounds, count, storage/display type, trailing zero bits), lineage (which base column the values come
from, and through which position mask), uniqueness, an optional name and a comment.  Two Vexps are
"equal" iff their operators are structurally equal (Vlite.hs:152-157: equality on the memoised
structural hash), which is what every memo table below keys on.  Bug-compatible detail: every RangeC
hashes alike (Vlite.hs:75,121 `show RangeC{} = "RangeC {...}"`).
"""
from .config import (D_DATE, INT64_MAX, INT64_MIN, S_INT32, S_INT64, ColInfo, FrontendError, NameTable, d_decimal,
                     dtype_of_mtype, show_name, stype_of_mtype)


class Vexp:
    __slots__ = ("vx", "info", "lineage", "name", "key", "quant", "comment")

    def __init__(self, vx, info, lineage, name, quant, comment):
        self.vx, self.info, self.lineage, self.name, self.quant, self.comment = vx, info, lineage, name, quant, comment
        self.key = vx_key(vx)

    def replace(self, **kw):
        v = Vexp.__new__(Vexp)
        for s in Vexp.__slots__:
            setattr(v, s, kw.get(s, getattr(self, s)))
        return v


# vx tuples:  ("Load", name) ("RangeV", rmin, rstep, rref) ("RangeC", rmin, rstep, rcount) ("Binop", op, l, r)
#             ("Shuffle", "Gather"|"Scatter", source, pos) ("Fold", op, groups, data) ("Partition", pivots, data)
#             ("VShuffle", arg) ("Like", data, pattern, col) ("CrossProduct", l, r, variant) ("Semisort", data)
_INTERN = {}
# compile_plan(distinct_rangec=True): give every RangeC its own identity.  NOT the reference's behaviour -- there all
# RangeC hash alike (Vlite.hs:72-75,121), so a program with two GROUP BYs partitions the second key over the FIRST key's
# pivots (TPC-H Q16, Q18: groups fall apart).  The tests use it to show that this is the only thing standing between
# the compiled programs of those two plans and their SQL (tests/test_sql_pins.py).
DISTINCT_RANGEC = False


def vx_key(vx):
    """Structural identity as a small integer (hash-consing): the operator with its operands replaced by
    THEIR identities.  Nested tuples would re-expand shared sub-DAGs into trees on every hash."""
    k = vx[0]
    flat = ("RangeC",) if k == "RangeC" and not DISTINCT_RANGEC else tuple(x.key if isinstance(x, Vexp) else x for x in vx)
    return _INTERN.setdefault(flat, len(_INTERN))


# lineage: None | (col_name, mask_vexp)


def complete(vx):
    """Vlite.hs:247-257."""
    name = vx[2].name if vx[0] == "Shuffle" else None
    lin = infer_lineage(vx)
    if lin is not None and (lin[1].lineage is not None or lin[1].name is not None):
        raise FrontendError("lineage vector should not itself have lineage or name")
    return Vexp(vx, infer_metadata(vx), lin, name, infer_uniqueness(vx), "")


def pos_(v): return complete(("RangeV", 0, 1, v))
def const_(k, v): return complete(("RangeV", k, 0, v))
def zeros_(v): return const_(0, v)
def ones_(v): return const_(1, v)


def typedconst_(k, v, dt):
    """Vlite.hs:183-186: literals keep their display type."""
    p = const_(k, v)
    return p.replace(info=p.info._replace(stype=S_INT32, dtype=(dt, "literal")))


def binop(op, l, r): return complete(("Binop", op, l, r))
def gather(values, positions): return complete(("Shuffle", "Gather", values, positions))
def scattered_to(values, positions): return complete(("Shuffle", "Scatter", values, positions))
def shl(a, b): return binop("BitShift", a, binop("Sub", zeros_(b), b))      # sign encodes direction, Vlite.hs:205-208


def cond(c, a, b):
    """c ? a : b, Vlite.hs:237-245."""
    negcond = binop("Eq", c, zeros_(c))
    poscond = binop("Sub", ones_(c), negcond)
    return binop("Add", binop("Mul", poscond, a), binop("Mul", negcond, b))


def _show_dtype(d):
    if d[0] == "DDecimal": return "DDecimal {point = %d}" % d[1]
    if d[0] == "DString": return "DString {decoder = %s}" % show_name(d[1])
    return "DDate"


def _shift(a, b):
    return a << (-b) if b < 0 else a >> b


def bitsize(num):
    if num < 0:
        raise FrontendError("bitwidth only allowed for non-negative numbers (num=%d)" % num)
    if num >= INT64_MAX:
        raise FrontendError("number %d is larger than maxInt int64" % num)
    return num.bit_length()


def get_bit_width(v):
    return max(bitsize(v.info.bounds[0]), bitsize(v.info.bounds[1]))


def max_for_width(v):
    return (1 << get_bit_width(v)) - 1


def infer_bounds(op, left, right):
    """Vlite.hs:417-467."""
    (l1, u1), (l2, u2) = left.info.bounds, right.info.bounds
    if op in ("Gt", "Lt", "Eq", "Neq", "Geq", "Leq", "LogAnd", "LogOr"): return (0, 1)
    if op == "Add": return (l1 + l2, u1 + u2)
    if op == "Sub": return (l1 - u2, u1 - l2)
    if op == "Mul":
        p = [l1 * l2, l1 * u2, u1 * l2, u1 * u2]
        return (min(p), max(p))
    if op == "Div":
        p = [x // y for x, y in ((l1, l2), (l1, u2), (u1, l2), (u1, u2))]      # Haskell `div` floors, as Python //
        return (min(p), max(p))
    if op == "Min": return (min(l1, l2), min(u1, u2))
    if op == "Max": return (max(l1, l2), max(u1, u2))
    if op == "Mod": return (0, u2 - 1)
    if op == "BitAnd":
        return (0, min(max_for_width(left), max_for_width(right))) if (l1 >= 0 and l2 >= 0) else (INT64_MIN, INT64_MAX)
    if op == "BitOr":
        return (0, max(max_for_width(left), max_for_width(right))) if (l1 >= 0 and l2 >= 0) else (INT64_MIN, INT64_MAX)
    if op == "BitShift":
        e = [_shift(a, b) for a, b in ((l1, l2), (l1, u2), (u1, l2), (u1, u2))]
        return (min(e), max(e))
    raise FrontendError("bounds of %s" % op)


_CMP = ("Gt", "Lt", "Leq", "Geq", "Eq", "Neq")


def infer_metadata(vx):
    """Vlite.hs:269-414."""
    k = vx[0]
    dec0 = (d_decimal(0), "")
    if k == "CrossProduct":
        lp, rp = pos_(vx[1]), pos_(vx[2])
        b = lp.info.bounds if vx[3] == "COuter" else rp.info.bounds
        return ColInfo(b, 0, lp.info.count * rp.info.count, S_INT32, dec0)
    if k == "Load":
        raise FrontendError("at the moment, should not be called with Load")
    if k in ("VShuffle", "Semisort"):        # Vlite.hs:294,322: keeps all current metadata the same
        return vx[1].info
    if k == "Like":
        return ColInfo((0, 1), 0, vx[1].info.count, S_INT32, dec0)
    if k == "RangeV":
        rstart, rstep, count = vx[1], vx[2], vx[3].info.count
        ex = [rstart, rstart + count * rstep]
        return ColInfo((min(ex), max(ex)), 0, count, S_INT64, dec0)
    if k == "RangeC":
        rstart, rstep, rcount = vx[1], vx[2], vx[3]
        ex = [rstart + rcount * rstep, rstart]
        return ColInfo((min(ex), max(ex)), 0, rcount, S_INT64, dec0)
    if k == "Shuffle":
        src, pos = vx[2].info, vx[3].info
        if vx[1] == "Scatter":
            return ColInfo(src.bounds, src.trailing_zeros, pos.bounds[1], src.stype, src.dtype)
        return ColInfo(src.bounds, src.trailing_zeros, pos.count, src.stype, src.dtype)
    if k == "Fold":
        op, g, d = vx[1], vx[2].info, vx[3].info
        if op == "FSel":
            return ColInfo((0, d.count - 1), 0, d.count, S_INT64, dec0)
        count_bound = min(g.bounds[1] - g.bounds[0] + 1, g.count)
        dl, du = d.bounds
        dt = d.dtype[0]
        if op == "FSum":
            ex = [dl, dl * d.count, du, du * d.count]
            out_dt = dt if dt[0] == "DDecimal" else d_decimal(0)
            return ColInfo((min(ex), max(ex)), d.trailing_zeros, count_bound, d.stype, (out_dt, ""))
        return ColInfo((dl, du), d.trailing_zeros, count_bound, d.stype, (dt, ""))
    if k == "Partition":
        return ColInfo((0, vx[1].info.count - 1), 0, vx[2].info.count, S_INT64, dec0)
    if k == "Binop":
        op, left, right = vx[1], vx[2], vx[3]
        li, ri = left.info, right.info
        tz = li.trailing_zeros - ri.bounds[1] if op == "BitShift" else 0
        lt, rt = li.stype, ri.stype
        if op == "Mul" and lt[0] == "SDecimal" and rt[0] == "SDecimal": stype = ("SDecimal", lt[1] + rt[1], lt[2] + rt[2])
        elif op == "Mul" and lt[0] == "SDecimal" and rt in (S_INT32, S_INT64): stype = lt
        elif op == "Mul" and rt[0] == "SDecimal": stype = rt
        elif op == "Div" and lt[0] == "SDecimal" and rt in (S_INT32, S_INT64): stype = lt
        elif op == "Div" and lt[0] == "SDecimal" and rt[0] == "SDecimal":
            if lt[2] - rt[2] < 0:
                raise FrontendError("implement division where numerator dec. point is less than denominator")
            stype = ("SDecimal", max(lt[1], rt[1]), lt[2] - rt[2])
        else: stype = lt
        ld, rd = li.dtype[0], ri.dtype[0]
        cs = "(%s,%s,%s)" % (op, _show_dtype(ld), _show_dtype(rd))
        if op == "Mul" and ld[0] == "DDecimal" and rd[0] == "DDecimal": dtype = (d_decimal(ld[1] + rd[1]), "")
        elif op == "Div" and ld[0] == "DDecimal" and rd[0] == "DDecimal":
            if ld[1] - rd[1] < 0:
                raise FrontendError("need to implement conversion for this division")
            dtype = (d_decimal(ld[1] - rd[1]), "")
        elif op in _CMP and ld == rd: dtype = (d_decimal(0), "")
        elif op in _CMP: dtype = (ld, "ERROR comparing across types without conversion " + cs)
        elif op in ("Sub", "Add") and ld[0] == "DDecimal" and rd[0] == "DDecimal":
            dtype = (ld, "") if ld[1] == rd[1] else (ld, "ERROR addition across different types without conversion " + cs)
        else: dtype = (ld, "WARNING case not implemented: " + cs)
        return ColInfo(infer_bounds(op, left, right), tz, min(li.count, ri.count), stype, dtype)
    raise FrontendError("metadata of %s" % k)


def infer_lineage(vx):
    """Vlite.hs:469-494: gather/scatter and min/max/choose folds keep the base-column lineage."""
    if vx[0] == "Shuffle" and vx[2].lineage is not None:
        col, lv = vx[2].lineage
        return (col, complete(("Shuffle", vx[1], lv, vx[3])))
    if vx[0] == "Fold" and vx[1] in ("FMin", "FMax", "FChoose") and vx[3].lineage is not None:
        col, lv = vx[3].lineage
        return (col, complete(("Fold", vx[1], vx[2], lv)))
    return None


def infer_uniqueness(vx):
    """Vlite.hs:496-520."""
    k = vx[0]
    if k == "Shuffle" and vx[1] == "Scatter": return vx[2].quant
    if k == "Shuffle" and vx[1] == "Gather" and vx[3].quant == "Unique": return vx[2].quant
    if k == "Partition": return "Unique"
    if k in ("RangeV", "RangeC") and vx[2] != 0: return "Unique"
    if k == "Fold" and vx[1] == "FSel": return "Unique"
    return "Any"


# ---- environments (Vlite.hs:530-548) ---------------------------------------------------------------
class Env:
    def __init__(self, lst, weak):
        self.list = lst
        self.table = NameTable()
        for v in lst:
            if v.name is not None:
                (self.table.insert_weak if weak else self.table.insert)(v.name, v)


def add_comment(s, vexps):
    return [v.replace(comment=v.comment + " " + s) for v in vexps]


# ---- scalar expressions (Vlite.hs:924-1019) ---------------------------------------------------------
def sc(env, e):
    k = e[0]
    if k == "Ref":
        return env.table.lookup(e[1])[1]
    if k == "Cast":
        if e[1] == ("MDouble",):
            return sc(env, e[2])                          # only used ahead of averages: dropped
        v = sc(env, e[2])
        in_dt = v.info.dtype[0]
        out_st = stype_of_mtype(e[1])
        nm = in_dt[1] if in_dt[0] == "DString" else None
        if e[1][0] in ("MChar", "MVarchar") and nm is None:
            raise FrontendError("cannot cast non-strings to strings")
        out_dt = dtype_of_mtype(e[1], nm)
        out = v
        if in_dt != out_dt and in_dt[0] == "DDecimal" and out_dt[0] == "DDecimal" and in_dt[1] != out_dt[1]:
            factor = 10 ** abs(out_dt[1] - in_dt[1])          # 1 -> 1.00 is represented as 100
            out = binop("Mul" if out_dt[1] > in_dt[1] else "Div", v, const_(factor, v))
        return out.replace(info=out.info._replace(stype=out_st, dtype=(out_dt, "")))
    if k == "Binop":
        return binop(e[1], sc(env, e[2]), sc(env, e[3]))
    if k == "In":
        left = sc(env, e[1])
        eqs = [binop("Eq", sc(env, x), left) for x in e[2]]
        if not eqs:
            raise FrontendError("list is empty here")
        out = eqs[0]
        for x in eqs[1:]:
            out = binop("LogOr", out, x)
        return out
    if k == "Literal":
        return typedconst_(e[2], env.list[0], e[1])
    if k == "Identity":
        return pos_(env.list[0])
    if k == "Unary" and e[1] == "Year":
        # ((days * 1000) + 1100) / 365243, Vlite.hs:988-994.  The source line reads `(d *. 1000) +. 1100 /. v365243`
        # and the module declares no fixities, so all three operators are infixl 9: the sum is divided, not the 1100
        d = sc(env, e[2])
        return binop("Div", binop("Add", binop("Mul", d, const_(1000, d)), const_(1100, d)), const_(365243, d))
    if k == "IfThenElse":
        c, t, el = e[1], e[2], e[3]
        if c[0] == "Unary" and c[1] == "IsNull" and t[0] == "Literal" and t[2] == 0 and c[2] == el:
            return sc(env, c[2])
        return cond(sc(env, c), sc(env, t), sc(env, el))
    if k == "Like":
        d = sc(env, e[1])
        if d.lineage is None:
            raise FrontendError("cannot apply like expressions without knowing lineage")
        return complete(("Like", d, e[2], d.lineage[0]))
    if k == "Unary" and e[1] == "Neg":
        a = sc(env, e[2])
        return binop("Sub", ones_(a), a)
    raise FrontendError("unhandled mplan scalar expression: %r" % (e,))


# ---- relational operators (Vlite.hs:562-732) ----------------------------------------------------------
def get_ref_vector(config, table):
    """Vlite.hs:734-741: the table's primary-key constraint column, used only for its length."""
    pk = config.lookup_pkey(table)
    info = config.colinfo.lookup(pk)[1]
    if config.format == "vlite":
        return complete(("RangeC", 0, 1, info.count))
    return Vexp(("Load", pk), info, None, None, "Unique", "ref vector")


def load_as(config, table, col, alias):
    """Vlite.hs:743-755."""
    mask = pos_(get_ref_vector(config, table))
    outname = alias if alias is not None else col
    if len(col) == 2 and col[1] == "%TID%":
        return mask.replace(lineage=(col, mask), name=outname)
    if len(col) != 2:
        raise FrontendError("unexpected name")
    info = config.colinfo.lookup(col)[1]
    quant = "Unique" if config.is_pkey([col]) is not None else "Any"
    return Vexp(("Load", col), info, (col, mask), outname, quant, "")


def solve(config, rel):
    return Env(solve_list(config, rel), weak=False)


def output_name(expr, alias):
    if alias is not None: return alias
    if expr[0] == "Ref": return expr[1]
    return None


def solve_list(config, rel):
    k = rel[0]
    if k == "Table":
        return [load_as(config, rel[1], c, a) for c, a in rel[2]]
    if k == "Project":
        if rel[3]:
            raise FrontendError("ordered queries are not supported")
        l0 = solve(config, rel[1]).list
        acc = []
        for expr, alias in rel[2]:
            env = Env(l0 + acc, weak=True)
            acc = [sc(env, expr).replace(name=output_name(expr, alias))] + acc
        return acc
    if k == "GroupBy":
        return solve_groupby(config, rel)
    if k == "CartesianProduct":
        lv, rv = solve(config, rel[1]).list, solve(config, rel[2]).list
        outer = complete(("CrossProduct", lv[0], rv[0], "COuter"))
        inner = complete(("CrossProduct", lv[0], rv[0], "CInner"))
        return gather_all(lv, outer) + gather_all(rv, inner)
    if k == "Join":
        return solve_join(config, rel)
    if k == "Select":
        child = solve(config, rel[1])
        fdata = sc(child, rel[2])
        idx = complete(("Fold", "FSel", pos_(fdata), fdata))
        return [gather(c, idx).replace(name=c.name) for c in child.list]
    raise FrontendError("unsupported M.rel: %s" % k)


def gather_all(cols, pos):
    return [gather(c, pos) for c in cols]


# ---- aggregation (Vlite.hs:624-669, 1033-1194) ----------------------------------------------------------
def shift_to_zero(v):
    vmin, tz = v.info.bounds[0], v.info.trailing_zeros
    if vmin == 0 and tz == 0:
        return v
    norm = binop("BitShift", v, const_(tz, v))
    return binop("Sub", norm, const_(norm.info.bounds[0], norm))


def compose_keys(l, r):
    sl, sr = shift_to_zero(l), shift_to_zero(r)
    if get_bit_width(sl) + get_bit_width(sr) >= 65:
        raise FrontendError("composite group key wider than 64 bits")
    return binop("BitOr", shl(sl, const_(get_bit_width(sr), sl)), sr)


def make_composite_key(config, keys):
    out = shift_to_zero(keys[0])
    for r in keys[1:]:
        out = compose_keys(out, r)
    if config.gboffset > 0:
        out = binop("Add", out, const_(config.gboffset, out)).replace(comment="offset added by goffset")
    out = out.replace(info=out.info._replace(bounds=(0, out.info.bounds[1])))      # override the lower bound
    if config.format == "vlite":
        return out
    hint = const_(max_for_width(out), out).replace(comment="size hint for voodoo backend")
    return binop("BitAnd", out, hint)


def get_scatter_mask(config, predata):
    """Vlite.hs:1082-1098."""
    lo, hi = predata.info.bounds
    if lo == hi:
        return pos_(predata), "Dense"
    pivots = complete(("RangeC", lo, 1, hi - lo + 1))
    sparse = "Sparse" if (hi - lo + 1) > 32000 else "Dense"
    strat = config.aggregation_strategy[0]
    if strat == "AggSerial": pdata = predata
    elif strat == "AggShuffle" or sparse == "Sparse": pdata = complete(("VShuffle", predata))
    else: pdata = predata
    return complete(("Partition", pivots, pdata)), sparse


def make_2level_fold(sparsity, config, op, fgroups, fdata):
    plain = ("Fold", op, fgroups, fdata)
    strat = config.aggregation_strategy
    if sparsity == "Dense" and strat[0] == "AggHierarchical":
        pos = pos_(fgroups)
        level1par = binop("BitAnd", binop("BitShift", pos, const_(strat[1], fgroups)), ones_(fgroups))
        level1 = complete(("Fold", op, compose_keys(fgroups, level1par), fdata))
        return complete(("Fold", op, fgroups, level1))
    return complete(plain)


def solve_agg(config, env, after, gkey, agg):
    if agg[0] == "GAvg":
        return binop("Div", solve_agg(config, env, after, gkey, ("GFold", "FSum", agg[1])),
                     solve_agg(config, env, after, gkey, ("GCount",)))
    if agg[0] == "GCount":
        return solve_agg(config, env, after, gkey, ("GFold", "FSum", ("Literal", d_decimal(0), 1)))
    op, expr = agg[1], agg[2]

    def default():
        gdata = sc(env, expr)
        if config.format == "vlite":             # Vlite.hs:1061-1064: sort-based grouping, no Partition / Scatter
            gmask = complete(("Semisort", gkey))
            return complete(("Fold", op, gather(gkey, gmask), gather(gdata, gmask)))
        mask, sparsity = get_scatter_mask(config, gkey)
        return make_2level_fold(sparsity, config, op, scattered_to(gkey, mask), scattered_to(gdata, mask))

    if op == "FChoose" and expr[0] == "Ref" and after.table.contains(expr[1]):
        return after.table.lookup(expr[1])[1]      # already a grouped column
    return default()


def solve_groupby(config, rel):
    child = solve(config, rel[1])
    if not child.list:
        raise FrontendError("empty env")
    refv = child.list[0]
    keys = rel[2]
    keyvecs = [child.table.lookup(n)[1] for n, _ in keys]
    keyaliases = [v.replace(name=a) for v, (_, a) in zip(keyvecs, keys) if a is not None]
    list1 = child.list + keyaliases
    gbkeys = keyvecs if keyvecs else [zeros_(refv)]
    gkey = make_composite_key(config, gbkeys).replace(comment="groupBy key")
    acc = []
    for agg, alias in rel[3]:
        env = Env(list1 + acc, weak=True)
        after = Env(acc, weak=False)
        anon = solve_agg(config, env, after, gkey, agg)
        is_choose_ref = agg[0] == "GFold" and agg[1] == "FChoose" and agg[2][0] == "Ref"
        outalias = agg[2][1] if (is_choose_ref and alias is None) else alias
        quant = anon.quant
        if len(keys) == 1 and is_choose_ref and agg[2][1] == keys[0][0]:
            quant = "Unique"
        lin = anon.lineage
        if lin is not None:
            lin = (lin[0], lin[1].replace(quant="Unique" if quant == "Unique" else lin[1].quant))
        acc = [anon.replace(name=outalias, quant=quant, lineage=lin)] + acc
    return add_comment("groupBy output", acc)


# ---- joins (Vlite.hs:682-719, 764-903, 1197-1282) ---------------------------------------------------------
def solve_join(config, rel):
    sleft, sright = solve(config, rel[1]), solve(config, rel[2])
    conds, variant = rel[3], rel[4]
    specs, rest = separate_fk_joinable(config, conds, sleft, sright)
    if len(specs) == 1 and not rest:
        spec = specs[0]
        if spec["kind"] == "fk" and spec["joinorder"] == "DimFact":
            return handle_gather_join(config, sright, sleft, variant, spec)
        return handle_gather_join(config, sleft, sright, variant, spec)
    if not specs and len(rest) == 1 and rest[0][0] == "Binop":
        op = rest[0][1]
        kl, kr = sc(sleft, rest[0][2]), sc(sright, rest[0][3])
        if kl.info.count == 1 and len(sleft.list) == 1:
            boolean = binop(op, gather(kl, zeros_(kr)), kr)
            mask = complete(("Fold", "FSel", pos_(boolean), boolean))
            return gather_all(sright.list, mask)
        if kr.info.count == 1 and len(sright.list) == 1:
            boolean = binop(op, kl, gather(kr, zeros_(kl)))
            mask = complete(("Fold", "FSel", pos_(boolean), boolean))
            return add_comment("join output", gather_all(sleft.list, mask))
    if len(specs) == 1 and len(rest) == 1:
        if variant != "Plain":
            raise FrontendError("can only do this rewrite for plain joins")
        inner = ("Join", rel[1], rel[2], [c for c in conds if c != rest[0]], variant)
        return solve_list(config, ("Select", inner, rest[0]))
    raise FrontendError("not handling this join case right now")


def separate_fk_joinable(config, conds, left, right):
    joinenv = NameTable()
    for n, v in left.table.to_list(): joinenv.insert_weak(n, ("L", v))
    for n, v in right.table.to_list(): joinenv.insert_weak(n, ("R", v))
    partials = {}            # key -> [partial spec, acc cols, unique, exprs]; insertion order irrelevant (one expected)
    non = []
    for expr in conds:
        added = False
        if expr[0] == "Binop" and expr[1] == "Eq" and expr[2][0] == "Ref" and expr[3][0] == "Ref":
            (s1, v1), (s2, v2) = joinenv.lookup(expr[2][1])[1], joinenv.lookup(expr[3][1])[1]
            if {s1, s2} == {"L", "R"} and v1.lineage is not None and v2.lineage is not None:
                lv, rv = (v1, v2) if s1 == "L" else (v2, v1)
                added = _process_partial(config, partials, lv, rv, expr)
        if not added:
            non = [expr] + non
    specs, more = [], []
    for key in sorted(partials, key=repr):
        p = partials[key]
        if p["kind"] == "fk":
            if tuple(sorted(p["acc"])) == tuple(p["pcols"]):
                inst = config.is_fkref(p["acc"])
                if inst is None:
                    raise FrontendError("kp was result of lookup earlier")
                specs.append({"kind": "fk", "factmask": p["factmask"].replace(comment="factmask"),
                              "dimmask": p["dimmask"].replace(comment="dimmmask"), "factunique": p["unique"],
                              "joinorder": p["joinorder"], "joinidx": inst.idxname,
                              "dimref": get_ref_vector(config, inst.dim)})
            else:
                more += p["exprs"]
        else:
            if tuple(sorted(p["acc"])) == tuple(p["pkcols"]):
                specs.append({"kind": "self", "leftmask": p["leftmask"], "rightmask": p["rightmask"]})
            else:
                more += p["exprs"]
    return specs, more + non


def _process_partial(config, partials, lv, rv, expr):
    """Vlite.hs:877-903."""
    (lcol, lmask), (rcol, rmask) = lv.lineage, rv.lineage
    if lcol == rcol:
        pks = config.partialpks.get(lcol)
        if pks is None or not (lmask.quant == "Unique" or rmask.quant == "Unique"):
            return False
        key = ("self", lmask.key, rmask.key, pks)
        p = partials.setdefault(key, {"kind": "self", "leftmask": lmask, "rightmask": rmask, "pkcols": pks, "acc": [], "exprs": []})
        p["acc"].append(lcol)
        p["exprs"].append(expr)
        return True
    hit = config.partialfks.get((lcol, rcol))
    if hit is None:
        return False
    order, kp = hit
    if order == "FactDim":
        fm, dm, pair, uq = lmask, rmask, (lcol, rcol), lv.quant
    else:
        fm, dm, pair, uq = rmask, lmask, (rcol, lcol), rv.quant
    key = ("fk", fm.key, dm.key, kp, order)
    p = partials.setdefault(key, {"kind": "fk", "factmask": fm, "dimmask": dm, "pcols": kp, "joinorder": order,
                                  "acc": [], "unique": "Any", "exprs": []})
    p["acc"].append(pair)
    if uq == "Unique":
        p["unique"] = "Unique"
    p["exprs"].append(expr)
    return True


def deduce_masks(config, spec):
    """Vlite.hs:1248-1282: positional (join-index) gather join."""
    idx_name = spec["joinidx"]
    fact_dim_idx = Vexp(("Load", idx_name), config.colinfo.lookup(idx_name)[1], None, None, "Any", "")
    fprime_dim_idx = gather(fact_dim_idx, spec["factmask"]).replace(quant=spec["factunique"])
    dimprime = spec["dimmask"]
    if dimprime.quant != "Unique":
        raise FrontendError("the dimension column is not known to be unique")
    # Vlite.hs:1270-1275: the VLite format scatters with an explicit shape (tfScatterTo); the shape is dropped
    # again when the Scatter is lowered (Vdl.hs:234-242 ignores shshape), so both formats build the same vector
    valid = scattered_to(ones_(dimprime), dimprime)
    didx = scattered_to(pos_(dimprime), dimprime)
    return gather(valid, fprime_dim_idx), gather(didx, fprime_dim_idx)


def handle_gather_join(config, fact, dim, variant, spec):
    """Vlite.hs:1199-1246."""
    if spec["kind"] == "self":
        lm, rm = spec["leftmask"], spec["rightmask"]
        if rm.vx[0] == "RangeV" and rm.vx[1] == 0 and rm.vx[2] == 1: factcols, dimcols, gm = fact.list, dim.list, lm
        elif lm.vx[0] == "RangeV" and lm.vx[1] == 0 and lm.vx[2] == 1: factcols, dimcols, gm = dim.list, fact.list, rm
        else: raise FrontendError("TODO: both children of this self join have been modified")
        if variant != "Plain":
            raise FrontendError("TODO: not a plain selfjoin")
        return factcols + gather_all(dimcols, gm)
    selectboolean, gathermask = deduce_masks(config, spec)
    selectmask = complete(("Fold", "FSel", pos_(selectboolean), selectboolean)).replace(comment="selectmask")
    cleaned = gather_all([gathermask] + fact.list, selectmask)
    clean_gathermask, cleaned_fact = cleaned[0], cleaned[1:]
    joined_dim = gather_all(dim.list, clean_gathermask)
    if variant == "Plain":
        return cleaned_fact + joined_dim
    if variant == "LeftSemi":
        if spec["joinorder"] == "FactDim":
            return cleaned_fact
        sm = gathermask.replace(comment="dim semijoin fact scattermask")
        lo, hi = sm.info.bounds
        hinted = binop("Mod", sm, const_(hi, sm).replace(comment="scatter size hint for voodoo backend"))
        qualified = scattered_to(ones_(sm), hinted)
        dsel = complete(("Fold", "FSel", pos_(qualified), qualified))
        return gather_all(dim.list, dsel)
    if variant == "LeftAnti" and spec["joinorder"] == "FactDim":
        anti = binop("Sub", ones_(selectmask), selectmask)
        asel = complete(("Fold", "FSel", pos_(anti), anti))
        return gather_all(fact.list, asel)
    raise FrontendError("join variant %s not implemented for this side" % variant)


# ---- cleanup passes (Vlite.hs:1290-1417) --------------------------------------------------------------------
def _is_range(v, rmin, rstep):
    return v.vx[0] == "RangeV" and v.vx[1] == rmin and v.vx[2] == rstep


def redundant_range(vx):
    if vx[0] == "RangeV" and vx[3].vx[0] == "RangeV":
        return complete(("RangeV", vx[1], vx[2], vx[3].vx[3]))
    return None


def lowering(vx):
    if vx[0] == "Binop":
        op, l, r = vx[1], vx[2], vx[3]
        if op == "Max": return cond(binop("Gt", l, r), l, r)
        if op == "Min": return cond(binop("Gt", r, l), l, r)
        if op == "Neq": return binop("Sub", ones_(l), binop("Eq", l, r))
    return None


def algebraic_identities(vx):
    if vx[0] == "Binop":
        op, l, r = vx[1], vx[2], vx[3]
        if op in ("BitAnd", "BitOr") and l.key == r.key: return l
        if op == "BitAnd" and _is_range(l, 0, 0): return l
        if op == "BitAnd" and _is_range(r, 0, 0): return r
        if op == "BitOr" and _is_range(l, 0, 0): return r
        if op == "BitOr" and _is_range(r, 0, 0): return l
        if op == "BitShift" and _is_range(l, 0, 0): return l
        if op == "BitShift" and _is_range(r, 0, 0): return l
    if vx[0] == "Shuffle":
        if vx[1] == "Scatter" and _is_range(vx[3], 0, 1): return vx[2]
        if vx[1] == "Gather" and _is_range(vx[3], 0, 1) and vx[3].vx[3].key == vx[2].key: return vx[2]
    return None


def _transform(fn, v, memo):
    """Vlite.hs:1358-1374.  The memo is keyed by structure; names merge as in the reference."""
    if v.key not in memo:
        anon = v if v.vx[0] == "Load" else _transform_vx(fn, v.vx, memo)
        memo[v.key] = anon.replace(name=v.name, comment=v.comment, info=v.info)
    x = memo[v.key]
    newx = x.replace(name=v.name if v.name is not None else x.name)
    memo[v.key] = newx
    return newx


def _transform_vx(fn, vx, memo):
    k = vx[0]
    if k in ("Load", "RangeC"): new = vx
    elif k == "RangeV": new = (k, vx[1], vx[2], _transform(fn, vx[3], memo))
    elif k == "Binop":
        l = _transform(fn, vx[2], memo)
        new = (k, vx[1], l, _transform(fn, vx[3], memo))
    elif k == "Shuffle":
        s = _transform(fn, vx[2], memo)
        new = (k, vx[1], s, _transform(fn, vx[3], memo))
    elif k == "Fold":
        g = _transform(fn, vx[2], memo)
        new = (k, vx[1], g, _transform(fn, vx[3], memo))
    elif k == "Partition":
        pv = _transform(fn, vx[1], memo)
        new = (k, pv, _transform(fn, vx[2], memo))
    elif k == "CrossProduct":
        l = _transform(fn, vx[1], memo)
        new = (k, l, _transform(fn, vx[2], memo), vx[3])
    elif k == "Like": new = (k, _transform(fn, vx[1], memo), vx[2], vx[3])
    elif k in ("VShuffle", "Semisort"): new = (k, _transform(fn, vx[1], memo))      # Vlite.hs:1386-1388
    else: raise FrontendError("transform of %s" % k)
    out = fn(new)
    return out if out is not None else complete(new)


def xform(fn, vexps):
    memo = {}
    return [_transform(fn, v, memo).replace(name=v.name) for v in vexps]


def vexps_from_mplan(rel, config, apply_passes=True):
    """MainFuns.hs:183-186: redundantRange, then lowering, then algebraic identities."""
    vexps = solve_list(config, rel)
    if apply_passes:
        for fn in (redundant_range, lowering, algebraic_identities):
            vexps = xform(fn, vexps)
    return vexps
