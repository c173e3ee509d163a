"""Scanner and parsers of the mplan -> VDL front end.

Restates /root/reference/src/Scanner.x (tokens), Parser.y (MonetDB `plan` text -> parse tree) and
SchemaParser.y (`msqldump -D` DDL -> tables with primary / foreign keys) as a hand-written
recursive-descent parser.  Parse-tree nodes are plain tuples / small classes.
"""
import re

from .config import FKey, FrontendError, PKey, Table

# ---- comment stripping, MainFuns.hs:83-92 ------------------------------------------------------


def filter_comments(text):
    out = []
    for ln in text.split("\n"):
        s = ln.lstrip(" ")
        out.append("" if s.startswith(("#", "%", "--", "[")) else ln)
    return "\n".join(out)


def read_commented(path):
    with open(path) as f:
        return filter_comments(f.read())


# ---- scanner, Scanner.x:17-45 --------------------------------------------------------------------
_NAME = r"[A-Za-z0-9<>=!_%]"
_TOKEN = re.compile(
    r"(?P<ws>[\s|]+)"
    r"|(?P<punct>[\[\]\(\),\.;])"
    r"|(?P<lit>\"[A-Za-z0-9<>=!_%\- #]*\")"
    r"|(?P<multi>NOT NULL|no nil|PRIMARY KEY|FOREIGN KEY|CREATE TABLE)"
    r"|(?P<word>" + _NAME + r"+)")


def scan(text):
    """Tokens: ('p', ch) punctuation, ('lit', s) quoted literal (quotes kept), ('num', int), ('w', word)."""
    toks, pos, line = [], 0, 1
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise FrontendError("lexical error at line %d near %r" % (line, text[pos:pos + 20]))
        kind = m.lastgroup
        s = m.group(kind)
        if kind == "multi":
            # alex takes the longest match: "NOT NULLx" would be a word, keep that behaviour
            m2 = re.compile(_NAME + r"+").match(text, pos)
            if m2 and m2.end() > m.end():
                kind, s, m = "word", m2.group(0), m2
        if kind == "ws":
            line += s.count("\n")
        elif kind == "punct":
            toks.append(("p", s, line))
        elif kind == "lit":
            toks.append(("lit", s, line))
        elif kind == "multi":
            toks.append(("w", s, line))
        else:
            toks.append(("num", int(s), line) if s.isdigit() else ("w", s, line))
        pos = m.end()
    return toks


# ---- parse tree, Parser.y:217-271 ------------------------------------------------------------------
class Expr:
    """Parser.Expr: a scalar expression with an optional `as` alias."""

    def __init__(self, expr, alias=None):
        self.expr, self.alias = expr, alias


# scalar expression nodes are tuples tagged by their first element:
#   ("Literal", (tname, tparams), string)        ("Ref", name, attrs)         ("Call", fname, [Expr])
#   ("Cast", (tname, tparams), Expr)             ("Infix", op, Expr, Expr)    ("Interval", Expr, op1, Expr, op2, Expr)
#   ("Filter", Expr, oper, negated, Expr, scalar)  ("In", Expr, negated, [Expr])  ("Nested", [Expr])
class Leaf:
    def __init__(self, source, columns):
        self.source, self.columns = source, columns


class Node:
    def __init__(self, relop, children, arg_lists):
        self.relop, self.children, self.arg_lists = relop, children, arg_lists


_SPECIAL = {"COUNT", "NOT NULL", "HASHCOL", "JOINIDX", "HASHIDX", "FETCH", "ASC", "FILTER", "in", "notin", "no nil",
            "table", "as", "!"}


def dropsys(parts):
    return parts[1:] if parts and parts[0] == "sys" else parts


class _P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None, -1)

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def fail(self, what):
        tok = self.peek()
        raise FrontendError("At line %s: expected %s, got %r" % (tok[2], what, tok[1]))

    def is_p(self, ch, k=0):
        tok = self.peek(k)
        return tok[0] == "p" and tok[1] == ch

    def eat_p(self, ch):
        if not self.is_p(ch):
            self.fail("'%s'" % ch)
        self.i += 1

    def is_w(self, word, k=0):
        tok = self.peek(k)
        return tok[0] == "w" and tok[1] == word

    def eat_w(self, word):
        if not self.is_w(word):
            self.fail("'%s'" % word)
        self.i += 1

    def is_ident(self, k=0):
        tok = self.peek(k)
        return tok[0] == "w" and tok[1] not in _SPECIAL

    # -- relational tree
    def tree(self):
        if self.is_w("table"):
            self.next()
            self.eat_p("(")
            src = self.qname()
            self.eat_p(")")
            self.eat_p("[")
            cols = self.expr_list_ne()
            self.eat_p("]")
            self.eat_w("COUNT")
            return Leaf(src, cols)
        words = []
        while self.is_ident():
            words.append(self.next()[1])
        if not words:
            self.fail("relational operator")
        self.eat_p("(")
        children = [self.tree()]
        while self.is_p(","):
            self.next()
            children.append(self.tree())
        self.eat_p(")")
        lists = []
        while self.is_p("["):
            self.next()
            lists.append([] if self.is_p("]") else self.expr_list_ne())
            self.eat_p("]")
        if not lists:
            self.fail("'['")
        return Node(" ".join(words), children, lists)

    def qname(self):
        parts = []
        if not self.is_ident():
            self.fail("identifier")
        parts.append(self.next()[1])
        while self.is_p(".") and self.is_ident(1):
            self.next()
            parts.append(self.next()[1])
        return tuple(dropsys(parts))

    def expr_list_ne(self):
        out = [self.expr()]
        while self.is_p(","):
            self.next()
            out.append(self.expr())
        return out

    # -- Expr: ExprBind [op ExprBind [op ExprBind]]   (Parser.y:139-153)
    def expr(self):
        first = self.expr_bind()
        if not self.is_ident():
            return first
        op1 = self.next()[1]
        mid = self.expr_bind()
        if not self.is_ident():
            return Expr(("Infix", op1, first, mid))
        op2 = self.next()[1]
        last = self.expr_bind()
        return Expr(("Interval", first, op1, mid, op2, last))

    def expr_bind(self):
        e = self.basic()
        alias = None
        if self.is_w("as"):
            self.next()
            alias = self.qname()
        out = Expr(e, alias)
        # postfix forms that start with an ExprBind: FILTER / in / notin (Parser.y:199-212)
        while True:
            if self.is_w("FILTER") or (self.is_w("!") and self.is_w("FILTER", 1)):
                neg = self.is_w("!")
                if neg:
                    self.next()
                self.next()
                oper = self.next()[1]
                self.eat_p("(")
                pat = self.expr()
                self.eat_p(",")
                esc = self.basic()
                self.eat_p(")")
                out = Expr(("Filter", out, oper, neg, pat, esc))
            elif self.is_w("in") or self.is_w("notin"):
                neg = self.next()[1] == "notin"
                self.eat_p("(")
                items = [] if self.is_p(")") else self.expr_list_ne()
                self.eat_p(")")
                out = Expr(("In", out, neg, items))
            else:
                return out

    def attrs(self):
        out = []
        while True:
            if self.is_w("NOT NULL"): self.next(); out.append(("NotNull",))
            elif self.is_w("ASC"): self.next(); out.append(("Asc",))
            elif self.is_w("HASHCOL"): self.next(); out.append(("HashCol",))
            elif self.is_w("HASHIDX"): self.next(); out.append(("HashIdx",))
            elif self.is_w("FETCH"): self.next(); out.append(("Fetch",))
            elif self.is_w("JOINIDX"):
                self.next()
                out.append(("JoinIdx", self.qname()))
            else:
                return out

    def basic(self):
        if self.is_p("("):
            self.next()
            items = self.expr_list_ne()
            self.eat_p(")")
            return ("Nested", items)
        if not self.is_ident():
            self.fail("expression")
        # TypeSpec '[' Expr ']'  |  TypeSpec literal   (type name, optional numeric parameters)
        k = 1
        params = ()
        if self.is_p("(", 1) and self.peek(2)[0] == "num":
            j = 2
            nums = []
            while self.peek(j)[0] == "num":
                nums.append(self.peek(j)[1])
                j += 1
                if self.is_p(",", j) and self.peek(j + 1)[0] == "num":
                    j += 1
                else:
                    break
            if self.is_p(")", j) and (self.peek(j + 1)[0] == "lit" or self.is_p("[", j + 1)):
                params, k = tuple(nums), j + 1
        if self.peek(k)[0] == "lit" or (self.is_p("[", k) and (k > 1 or not self.is_p(".", 1))):
            tname = self.next()[1]
            self.i += k - 1
            if self.peek()[0] == "lit":
                return ("Literal", (tname, params), self.next()[1][1:-1])
            self.eat_p("[")
            inner = self.expr()
            self.eat_p("]")
            return ("Cast", (tname, params), inner)
        name = self.qname()
        if self.is_w("no nil"):
            self.next()
        if self.is_p("("):
            self.next()
            args = [] if self.is_p(")") else self.expr_list_ne()
            self.eat_p(")")
            self.attrs()
            return ("Call", name, args)
        return ("Ref", name, self.attrs())


def parse_mplan(text):
    """MonetDB logical plan text (comments already blanked) -> Leaf / Node tree."""
    p = _P(scan(text))
    tree = p.tree()
    if p.peek()[0] != "eof":
        p.fail("end of plan")
    return tree


# ---- msqldump schema, SchemaParser.y:62-127 ----------------------------------------------------------
def parse_schema(text):
    p = _P(scan(text))

    def qq():
        parts = []
        while True:
            tok = p.next()
            if tok[0] != "lit":
                p.i -= 1
                p.fail("quoted identifier")
            parts.append(tok[1][1:-1])
            if p.is_p("."):
                p.next()
            else:
                break
        return tuple(dropsys(parts))

    def keyspec():
        p.eat_p("(")
        cols = [qq()]
        while p.is_p(","):
            p.next()
            cols.append(qq())
        p.eat_p(")")
        return cols

    p.eat_w("SET")
    p.eat_w("SCHEMA")
    qq()
    p.eat_p(";")
    tables = []
    while p.is_w("CREATE TABLE"):
        p.next()
        name = qq()
        p.eat_p("(")
        columns = []
        while p.peek()[0] == "lit":
            cname = qq()
            tname = p.next()[1]
            params = ()
            if p.is_p("("):
                p.next()
                nums = [p.next()[1]]
                while p.is_p(","):
                    p.next()
                    nums.append(p.next()[1])
                p.eat_p(")")
                params = tuple(nums)
            if p.is_w("NOT NULL"):
                p.next()
            p.eat_p(",")
            columns.append((cname, (tname, params)))
        keys = []
        while p.is_w("CONSTRAINT"):
            p.next()
            cons = qq()
            if p.is_w("PRIMARY KEY"):
                p.next()
                keys.append(PKey(keyspec(), cons))
            else:
                p.eat_w("FOREIGN KEY")
                local = keyspec()
                p.eat_w("REFERENCES")
                ref = qq()
                remote = keyspec()
                keys.append(FKey(ref, list(zip(local, remote)), cons))
            if p.is_p(","):
                p.next()
        p.eat_p(")")
        p.eat_p(";")
        if not keys or not isinstance(keys[0], PKey):
            raise FrontendError("table %s: the first constraint must be the primary key" % ".".join(name))
        tables.append(Table(name, columns, keys[0], keys[1:]))
    return tables
