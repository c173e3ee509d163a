"""Relational IR of the front end: parse tree -> RelExpr / ScalarExpr.

Restates /root/reference/src/Mplan.hs: date and interval folding (:46-57,368-388), literal typing via
the dictionary (:41-44,461-484), operator tables (:71-114), group-by outputs (:138-181), relational
operators (:227-356) and the two optional logical rewrites pushFKJoins / fuseSelects (:574-620).
Nodes are tagged tuples:
  scalar: ("Ref", name) ("Literal", dtype, int) ("Identity", e) ("Unary", op, e) ("Binop", op, l, r)
          ("IfThenElse", c, t, e) ("Cast", mtype, e) ("In", l, [e]) ("Like", e, pattern)
  rel   : ("Table", name, [(col, alias)]) ("Project", child, [(scalar, alias)], order) ("Select", child, pred)
          ("GroupBy", child, [(key, alias)], [(agg, alias)]) ("Join", l, r, [conds], variant)
          ("CartesianProduct", l, r) ("TopN", child, n)
  agg   : ("GAvg", e) ("GCount",) ("GFold", op, e)
"""
import calendar
import datetime

from .config import D_DATE, FrontendError, d_decimal, resolve_typespec
from .parse import Leaf, Node


def day_count(datestr):
    """Mplan.hs:46-57: days since 0000-01-01 (proleptic Gregorian; year 0 is a leap year)."""
    y, m, d = (int(x) for x in datestr.split("-"))
    return datetime.date(y, m, d).toordinal() + 365


def _add_months_rollover(date, months):
    """Data.Time addGregorianMonthsRollOver: day-of-month overflow rolls into the next month."""
    y, m = date.year, date.month - 1 + months
    y += m // 12
    m = m % 12 + 1
    last = calendar.monthrange(y, m)[1]
    if date.day <= last:
        return datetime.date(y, m, date.day)
    return datetime.date(y, m, last) + datetime.timedelta(days=date.day - last)


def read_int(s):
    try:
        return int(s)
    except ValueError:
        raise FrontendError("unrecognizable integer literal: %s" % s)


_INFIX = {"<": "Lt", ">": "Gt", "<=": "Leq", ">=": "Geq", "=": "Eq", "!=": "Neq", "or": "LogOr"}
_BINFUN = {"sql_add": "Add", "sql_sub": "Sub", "sql_mul": "Mul", "sql_div": "Div", "sql_min": "Min", "sql_max": "Max",
           "=": "Eq", "or": "LogOr", "and": "LogAnd", ">": "Gt", "<>": "Neq", "scale_down": "Div"}
_UNFUN = {"year": "Year", "sql_neg": "Neg", "isnull": "IsNull"}


class _Ctx:
    def __init__(self, config, dt=None):
        self.config, self.dt = config, dt


def _ref_dtype(config, scalar):
    """The display type a string literal on the other side of `scalar` must take.  Lazy, as in the
    reference (Haskell only forces the lookup when a char literal needs it, Mplan.hs:436-439,491-494)."""
    if scalar[0] == "Ref":
        return lambda: config.colinfo.lookup(scalar[1])[1].dtype[0]
    return None


def sc(expr, ctx):
    """Parser scalar -> Mplan scalar, Mplan.hs:361-549."""
    k = expr[0]
    config = ctx.config
    if k == "Ref":
        return ("Ref", expr[1])
    if k == "Call":
        fname, args = expr[1], expr[2]
        op = fname[0] if len(fname) == 1 else None
        a = [x.expr for x in args]
        # date +/- interval folded into a date literal (:368-388)
        if (op in ("sql_add", "sql_sub") and len(a) == 2 and a[0][0] == "Literal" and a[0][1][0] == "date"
                and a[1][0] == "Literal" and a[1][1][0] in ("month_interval", "sec_interval")):
            y, m, d = (int(x) for x in a[0][2].split("-"))
            date = datetime.date(y, m, d)
            num = read_int(a[1][2])
            if op == "sql_sub":
                num = -num
            if a[1][1][0] == "month_interval":
                out = _add_months_rollover(date, num)
            else:
                millis = 1000 * 60 * 60 * 24
                days = abs(num) // millis * (1 if num >= 0 else -1)          # Haskell `quot`
                out = date + datetime.timedelta(days=days)
            return sc(("Literal", ("date", ()), "%04d-%02d-%02d" % (out.year, out.month, out.day)), ctx)
        if fname == ("identity",) and len(a) == 1:
            return ("Identity", sc(a[0], ctx))
        if fname == ("like",):
            if (len(a) == 2 and a[1][0] == "Cast" and a[1][1] == ("char", ()) and a[1][2].expr[0] == "Literal"
                    and a[1][2].expr[1][0] == "char" and len(a[1][2].expr[1][1]) == 1):
                return ("Like", sc(a[0], ctx), a[1][2].expr[2])
            raise FrontendError("implement this 'like' case")
        if fname == ("ifthenelse",) and len(a) == 3:
            return ("IfThenElse", sc(a[0], ctx), sc(a[1], ctx), sc(a[2], ctx))
        if len(a) == 1:
            if op not in _UNFUN:
                raise FrontendError("unexpected scalar function %s" % ".".join(fname))
            return ("Unary", _UNFUN[op], sc(a[0], ctx))
        if len(a) == 2:
            left = sc(a[0], ctx)
            right = sc(a[1], _Ctx(config, _ref_dtype(config, left)))
            if op not in _BINFUN:
                raise FrontendError("unexpected binary function %s" % ".".join(fname))
            return ("Binop", _BINFUN[op], left, right)
        raise FrontendError("unexpected scalar operator %r" % (expr,))
    if k == "Cast":
        return ("Cast", resolve_typespec(*expr[1]), sc(expr[2].expr, ctx))
    if k == "Literal":
        mtype = resolve_typespec(*expr[1])
        s = expr[2]
        mk = mtype[0]
        if mk == "MDate":
            return ("Literal", D_DATE, day_count(s))
        if mk == "MDecimal":
            return ("Literal", d_decimal(mtype[2]), read_int(s))     # decimal(3,2) "6" is 0.06: the int is the representation
        if mk == "MBoolean":
            if s not in ("true", "false"):
                raise FrontendError("invalid boolean literal %s" % s)
            return ("Literal", d_decimal(0), 1 if s == "true" else 0)
        if mk in ("MTinyint", "MSmallint", "MInt", "MBigInt"):
            return ("Literal", d_decimal(0), read_int(s))
        if mk == "MChar":
            dt = ctx.dt() if callable(ctx.dt) else ctx.dt
            if dt is not None and dt[0] == "DString":
                if s not in config.dictionary:
                    raise FrontendError("not found in dictionary: %s" % s)
                return ("Literal", dt, config.dictionary[s])
            raise FrontendError("need more information to assign type to char literal %r" % s)
        raise FrontendError("unexpected literal: %r" % (expr,))
    if k == "Infix":
        left = sc(expr[2].expr, ctx)
        right = sc(expr[3].expr, _Ctx(config, _ref_dtype(config, left)))
        if expr[1] not in _INFIX:
            raise FrontendError("unexpected infix symbol %s" % expr[1])
        return ("Binop", _INFIX[expr[1]], left, right)
    if k == "Interval":
        first, mid, last = sc(expr[1].expr, ctx), sc(expr[3].expr, ctx), sc(expr[5].expr, ctx)
        return ("Binop", "LogAnd", ("Binop", _INFIX[expr[2]], first, mid), ("Binop", _INFIX[expr[4]], mid, last))
    if k == "In":
        arg, neg, items = expr[1], expr[2], expr[3]
        if arg.expr[0] != "Ref" or neg:
            raise FrontendError("implement this case of IN operator")
        left_dt = config.colinfo.lookup(arg.expr[1])[1].dtype[0]
        return ("In", sc(arg.expr, ctx), [sc(x.expr, _Ctx(config, left_dt)) for x in items])
    if k == "Nested":
        return conjunction(config, expr[1])
    if k == "Filter":
        arg, oper, neg, pat, esc = expr[1], expr[2], expr[3], expr[4], expr[5]
        if (oper == "like" and pat.expr[0] == "Cast" and pat.expr[1] == ("char", ()) and pat.alias is None
                and pat.expr[2].expr[0] == "Literal" and esc[0] == "Literal" and esc[2] == ""):
            like = ("Like", sc(arg.expr, ctx), pat.expr[2].expr[2])
            return ("Unary", "Neg", like) if neg else like
        raise FrontendError("unexpected operator %s" % oper)
    raise FrontendError("unexpected scalar operator %r" % (expr,))


def rsc(config, expr):
    return sc(expr, _Ctx(config))


def conjunction(config, exprs):
    """Mplan.hs:552-559: left-associated LogAnd of a predicate list."""
    solved = [rsc(config, e.expr) for e in exprs]
    if not solved:
        raise FrontendError("empty conjunction list")
    out = solved[0]
    for nxt in solved[1:]:
        out = ("Binop", "LogAnd", out, nxt)
    return out


def _group_output(config, e):
    """Mplan.hs:138-181."""
    x, alias = e.expr, e.alias
    if x[0] == "Ref":
        return (("GFold", "FChoose", ("Ref", x[1])), alias if alias is not None else x[1])
    if x[0] == "Call" and x[1] == ("count",) and not x[2]:
        return (("GCount",), alias)
    if x[0] == "Call" and len(x[2]) == 1:
        inner_p = x[2][0].expr
        inner = rsc(config, inner_p)
        f = x[1]
        if f == ("sum",): return (("GFold", "FSum", inner), alias)
        if f == ("avg",): return (("GAvg", inner), alias)
        if f == ("max",): return (("GFold", "FMax", inner), alias)
        if f == ("min",): return (("GFold", "FMin", inner), alias)
        if f == ("count",) and inner_p[0] == "Ref": return (("GCount",), alias)
        raise FrontendError("unexpected unary aggregate %s" % ".".join(f))
    raise FrontendError("unexpected group_by output expression")


_JOINS = {"join": "Plain", "semijoin": "LeftSemi", "antijoin": "LeftAnti", "left outer join": "LeftOuter"}


def solve(config, rel):
    """Parse tree -> RelExpr, Mplan.hs:227-356."""
    if isinstance(rel, Leaf):
        cols = []
        for e in rel.columns:
            if e.expr[0] != "Ref":
                raise FrontendError("table outputs should only have reference expressions")
            rname, attrs = e.expr[1], e.expr[2]
            fk = [a[1] for a in attrs if a[0] == "JoinIdx"]
            if len(fk) > 1:
                raise FrontendError("multiple fkey indices")
            if e.alias is None:
                cols.append((fk[0], rname) if fk else (rname, None))       # notice the reversal for join indices
            else:
                cols.append((fk[0], e.alias) if fk else (rname, e.alias))
        return ("Table", rel.source, cols)
    assert isinstance(rel, Node)
    op, ch, lists = rel.relop, rel.children, rel.arg_lists
    if op == "project" and len(ch) == 1:
        if len(lists) > 1:
            raise FrontendError("unexpected order-by clauses")
        return ("Project", solve(config, ch[0]), [(rsc(config, e.expr), e.alias) for e in lists[0]], [])
    if op == "group by" and len(ch) == 1 and len(lists) == 2:
        keys = []
        for e in lists[0]:
            if e.expr[0] != "Ref":
                raise FrontendError("non-ref in group by key")
            keys.append((e.expr[1], e.alias))
        return ("GroupBy", solve(config, ch[0]), keys, [_group_output(config, e) for e in lists[1]])
    if op == "select" and len(ch) == 1 and len(lists) == 1:
        return ("Select", solve(config, ch[0]), conjunction(config, lists[0]))
    if op in _JOINS and len(ch) == 2 and len(lists) == 1:
        if config.cross_product and op == "join":
            cross = ("CartesianProduct", solve(config, ch[0]), solve(config, ch[1]))
            return ("Select", cross, conjunction(config, lists[0]))
        conds = [rsc(config, e.expr) for e in lists[0]]
        if not conds:
            raise FrontendError("empty join condition list is invalid")
        return ("Join", solve(config, ch[0]), solve(config, ch[1]), conds, _JOINS[op])
    if op == "top N" and len(ch) == 1:
        e = lists[0][0]
        return ("TopN", solve(config, ch[0]), read_int(e.expr[2]))
    raise FrontendError("relational operator not implemented: %s" % op)


def _rewrite(node, fn):
    """uniplate `rewrite`: apply fn bottom-up until it no longer applies anywhere."""
    k = node[0]
    if k == "Project": node = (k, _rewrite(node[1], fn)) + node[2:]
    elif k in ("Select", "GroupBy", "TopN"): node = (k, _rewrite(node[1], fn)) + node[2:]
    elif k in ("Join", "CartesianProduct"): node = (k, _rewrite(node[1], fn), _rewrite(node[2], fn)) + node[3:]
    new = fn(node)
    return _rewrite(new, fn) if new is not None else node


def push_fk_joins(rel):
    """Mplan.hs:574-604: selects below a single-condition plain join move above it."""
    def swap(n):
        if n[0] == "Join" and len(n[3]) == 1 and n[4] == "Plain":
            l, r = n[1], n[2]
            if r[0] == "Select":
                return ("Select", ("Join", l, r[1], n[3], n[4]), r[2])
            if l[0] == "Select":
                return ("Select", ("Join", l[1], r, n[3], n[4]), l[2])
        return None
    return _rewrite(rel, swap)


def fuse_selects(rel):
    """Mplan.hs:607-620."""
    def fuse(n):
        if n[0] == "Select" and n[1][0] == "Select":
            return ("Select", n[1][1], ("Binop", "LogAnd", n[1][2], n[2]))
        return None
    return _rewrite(rel, fuse)
