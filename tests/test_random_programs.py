"""Random well-formed VDL programs (filters, nested filters, FK gathers with out-of-range keys, identity scatters,
Partition + Scatter + Fold group-bys, global folds, element-wise chains) through the oracle and the GPU engine:
the statement-by-statement executor rewrites vector forms aggressively (sparse selections, fused expression trees,
view scatters), and this is the net under it.  Each program also runs with those rewrites forced on / off."""
import os

import numpy as np
import pytest

from helpers import check_against_oracle, compare_traced, engine_with, oracle_run, prog

BINOPS = ["Add", "Subtract", "Multiply", "Greater", "Equals", "LogicalAnd", "LogicalOr", "BitwiseAnd", "BitwiseOr", "Divide", "Modulo", "BitShift"]
FOLDS = ["FoldSum", "FoldMin", "FoldMax", "FoldChoose", "FoldCount"]


class Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.lines, self.nid = [], 0
        self.pool = {}                                   # table -> statement ids of vectors of that table's length
        self.cols = {}
        big = int(os.environ.get("VDL_FUZZ_ROWS", "0"))     # tools/fuzz_general_path.py: tables of many tiles / blocks
        self.n = {"t": int(self.rng.integers(1, 4000)), "u": int(self.rng.integers(1, 300))}
        if big:
            self.n = {"t": int(self.rng.integers(big // 2, big)), "u": int(self.rng.integers(1, max(big // 50, 2)))}
        if seed % 17 == 0:
            self.n["t"] = 1 + seed % 3                       # degenerate tables now and then
        if seed % 23 == 0:
            self.n["u"] = 1
        for tab, names in (("t", "abcd"), ("u", "xy")):
            self.pool[tab] = []
            for c in names:
                lo, hi = (-20, 20) if c in "ax" else (0, int(self.rng.integers(2, 40)))
                self.cols["%s.%s" % (tab, c)] = self.rng.integers(lo, hi + 1, self.n[tab]).astype(self.rng.choice([np.int64, np.int32]))
                self.pool[tab].append(self.project(self.emit("Load,%s.%s" % (tab, c)), c))
        self.cols["t.fk"] = self.rng.integers(-2, self.n["u"] + 2, self.n["t"]).astype(np.int64)      # some keys fall outside u
        self.fk = self.project(self.emit("Load,t.fk"), "fk")
        self.pool["t"].append(self.fk)
        self.outputs = 0

    def emit(self, body):
        self.nid += 1
        self.lines.append("%d,%s" % (self.nid, body))
        return self.nid

    def project(self, v, field): return self.emit("Project,val,Id %d,%s" % (v, field))
    def rangev(self, frm, ref, step): return self.emit("RangeV,val,%d,Id %d,%d" % (frm, ref, step))
    def binary(self, op, a, b): return self.emit("%s,val,Id %d,val,Id %d,val" % (op, a, b))
    def gather(self, src, pos): return self.emit("Gather,Id %d,Id %d,val" % (src, pos))
    def pick(self, tab): return int(self.rng.choice(self.pool[tab]))
    def output(self, v):
        self.emit("MaterializeCompact,Id %d" % v)
        self.outputs += 1

    def step(self):
        r = self.rng
        tab = "t" if r.random() < 0.75 else "u"
        kind = r.choice(["bin", "bin", "const", "filter", "fk", "dimmask", "group", "fold", "out", "selruns"])
        if kind == "bin":
            a, b = self.pick(tab), self.pick(tab)
            if r.random() < 0.4:
                b = self.rangev(int(r.integers(-3, 9)), a, 0)
            self.pool[tab].append(self.binary(str(r.choice(BINOPS)), a, b))
        elif kind == "const":
            ref = self.pick(tab)
            self.pool[tab].append(self.rangev(int(r.integers(-5, 50)), ref, int(r.choice([0, 0, 1, 3]))))
        elif kind == "filter":                       # Gather(x, FoldSelect(RangeV 0 1 c, c)) for a few x
            cvec = self.pick(tab)
            if r.random() < 0.7:                     # make it selective now and then
                cvec = self.binary("Equals", cvec, self.rangev(int(r.integers(0, 6)), cvec, 0))
            sel = self.emit("FoldSelect,val,Id %d,val,Id %d,val" % (self.rangev(0, cvec, 1), cvec))
            for _ in range(int(r.integers(1, 4))):
                self.pool[tab].append(self.gather(self.pick(tab), sel))
        elif kind == "fk":                           # dimension vector through the (possibly filtered) foreign key
            key = self.fk if r.random() < 0.5 else self.gather(self.fk, self.emit(
                "FoldSelect,val,Id %d,val,Id %d,val" % (self.rangev(0, self.fk, 1), self.binary("Greater", self.pick("t"), self.rangev(3, self.fk, 0)))))
            self.pool["t"].append(self.gather(self.pick("u"), key))
        elif kind == "dimmask":                      # the dim side of a join: ones / row ids scattered back by filtered row ids
            cvec = self.pick("u")
            sel = self.emit("FoldSelect,val,Id %d,val,Id %d,val" % (self.rangev(0, cvec, 1), cvec))
            ids = self.gather(self.rangev(0, cvec, 1), sel)
            src = self.rangev(1, ids, 0) if r.random() < 0.5 else self.rangev(0, ids, 1)
            self.pool["u"].append(self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (src, self.rangev(0, src, 1), ids)))
        elif kind == "group":                        # Partition + Scatter + Fold (Vlite.hs:1056-1098)
            domain = int(r.choice([4, 32, 1000]))
            key = self.binary("BitwiseAnd", self.pick(tab), self.rangev(domain - 1, self.pick(tab), 0))
            part = self.emit("Partition,val,Id %d,val,Id %d,val" % (key, self.emit("RangeC,val,0,%d,1" % domain)))
            skey = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (key, self.rangev(0, key, 1), part))
            for _ in range(int(r.integers(1, 4))):
                x = self.pick(tab)
                sx = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (x, self.rangev(0, x, 1), part))
                self.output(self.emit("%s,val,Id %d,val,Id %d,val" % (r.choice(FOLDS), skey, sx)))
        elif kind == "selruns":                      # FoldSelect over runs of a low-cardinality control vector
            ctl = self.binary("BitwiseAnd", self.pick(tab), self.rangev(int(r.choice([1, 3])), self.pick(tab), 0))
            self.pool[tab].append(self.emit("FoldSelect,val,Id %d,val,Id %d,val" % (ctl, self.pick(tab))))
        elif kind == "fold":
            x = self.pick(tab)
            self.output(self.emit("%s,val,Id %d,val,Id %d,val" % (r.choice(FOLDS[:3]), self.rangev(0, x, 0), x)))
        else:
            self.output(self.pick(tab))

    def build(self, steps):
        for _ in range(steps):
            self.step()
        if not self.outputs:
            self.output(self.pick("t"))
        return prog(*self.lines), self.cols


def check(seed, steps, tag="random_programs"):
    text, cols = Gen(seed).build(steps)
    want = oracle_run(text, cols)
    e = engine_with(cols)
    got = e.run_vdl(text)["results"]
    e.close()
    check_against_oracle(tag, seed, text, cols, got, want)      # on a mismatch: report file + traced reruns (helpers.explain_mismatch)


def test_generator_produces_programs_the_oracle_accepts():
    for seed in range(40):
        text, cols = Gen(seed).build(25)
        assert oracle_run(text, cols)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [None, "VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE", "VDL_NO_EXPR_FUSION", "VDL_NO_FILTER_FUSION", "VDL_NO_GROUP_BATCH"])
def test_random_programs_match_the_oracle(monkeypatch, mode):
    if mode:
        monkeypatch.setenv(mode, "1")
    for seed in range(120):
        check(seed, 10 + seed % 30, "random_programs_%s" % (mode or "default"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [None, "VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE"])
def test_every_statement_of_random_programs_matches_the_oracle(monkeypatch, mode):
    """Not only the outputs: with tracing on, the vector of every statement the executor evaluated (whatever form it
    keeps it in) must equal the oracle's vector of that statement, slot by slot."""
    import oracle

    if mode:
        monkeypatch.setenv(mode, "1")
    for seed in range(200, 260):
        text, cols = Gen(seed).build(10 + seed % 30)
        orc = oracle.Oracle()
        orc.keep_vectors(True)
        for k, v in cols.items():
            orc.add_column(k, v)
        want = orc.run(text)["results"]
        e = engine_with(cols)
        p = e.parse(text)
        p.set_fusion(False)
        p.set_trace(True)
        got = p.run()["results"]
        first = compare_traced(p, orc, text)
        e.close()
        orc.close()
        assert first is None, "seed %d (%s): %s\n%s" % (seed, mode, first, text)
        check_against_oracle("traced_%s" % (mode or "default"), seed, text, cols, got, want)
