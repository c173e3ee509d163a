"""Random filter + aggregate programs in the shape mplan2vdl emits (Select -> FoldSelect + Gather, aggregates over
products of affine column terms, optional dense GROUP BY through Partition + Scatter): the planner turns these into
per-column ranges and affine factors for the fused scans, and falls back to statement-by-statement execution when it
cannot -- either way the answer must be the oracle's."""
import numpy as np
import pytest

import mplan2vdl_amd as m
from helpers import check_against_oracle, engine_with, oracle_run, prog


class Gen:
    def __init__(self, seed):
        self.rng = r = np.random.default_rng(seed)
        self.lines, self.nid = [], 0
        n = int(r.integers(1, 20000))
        self.cols = {}
        self.c = []
        for name, (lo, hi, dt) in {"a": (-50, 50, np.int64), "b": (0, 30, np.int32), "c": (1, 1000, np.int64), "d": (0, 7, np.int16)}.items():
            self.cols["t." + name] = r.integers(lo, hi + 1, n).astype(dt)
            self.c.append(self.emit("Project,val,Id %d,%s" % (self.emit("Load,t." + name), name)))

    def emit(self, body):
        self.nid += 1
        self.lines.append("%d,%s" % (self.nid, body))
        return self.nid

    def const(self, k, ref): return self.emit("RangeV,val,%d,Id %d,0" % (k, ref))
    def bin(self, op, a, b): return self.emit("%s,val,Id %d,val,Id %d,val" % (op, a, b))
    def col(self): return int(self.rng.choice(self.c))

    def predicate(self):
        r = self.rng
        terms = []
        for _ in range(int(r.integers(1, 4))):
            x = self.col()
            k = self.const(int(r.integers(-10, 40)), self.c[0])
            form = r.choice(["gt", "lt", "ge", "le", "eq", "between"])
            if form == "gt": t = self.bin("Greater", x, k)
            elif form == "lt": t = self.bin("Greater", k, x)
            elif form == "ge": t = self.bin("LogicalOr", self.bin("Greater", x, k), self.bin("Equals", k, x))
            elif form == "le": t = self.bin("LogicalOr", self.bin("Greater", k, x), self.bin("Equals", x, k))
            elif form == "eq": t = self.bin("Equals", x, k)
            else:
                k2 = self.bin("Add", k, self.const(int(r.integers(0, 20)), self.c[0]))          # constants that are not folded in the text
                t = self.bin("LogicalAnd", self.bin("Greater", x, k), self.bin("Greater", k2, x))
            terms.append(t)
        p = terms[0]
        for t in terms[1:]:
            p = self.bin("LogicalAnd" if r.random() < 0.85 else "LogicalOr", p, t)
        return p

    def term(self, sel):
        r = self.rng
        g = lambda v: self.emit("Gather,Id %d,Id %d,val" % (v, sel))
        t = g(self.col())
        for _ in range(int(r.integers(0, 3))):
            y = g(self.col())
            form = r.choice(["mul", "kminus", "kplus", "scale"])
            if form == "mul": t = self.bin("Multiply", t, y)
            elif form == "kminus": t = self.bin("Multiply", t, self.bin("Subtract", self.const(int(r.integers(1, 200)), y), y))
            elif form == "kplus": t = self.bin("Multiply", t, self.bin("Add", self.const(int(r.integers(1, 200)), y), y))
            else: t = self.bin("Multiply", t, self.const(int(r.integers(-3, 9)), t))
        return t

    def build(self):
        r = self.rng
        p = self.predicate()
        sel = self.emit("FoldSelect,val,Id %d,val,Id %d,val" % (self.emit("RangeV,val,0,Id %d,1" % p), p))
        grouped = r.random() < 0.5
        if grouped:
            g = lambda v: self.emit("Gather,Id %d,Id %d,val" % (v, sel))
            kb, kd = g(self.c[1]), g(self.c[3])
            shl = lambda v, k: self.bin("BitShift", v, self.bin("Subtract", self.const(0, v), self.const(k, v)))
            shape = int(r.integers(0, 5))
            if shape == 0:          # b << 3 | d
                key = self.bin("BitwiseOr", shl(kb, 3), kd)
            elif shape == 1:        # ((b >> 1) - 1) << 3 | (d - 2): right shifts and offsets per component (makeCompositeKey, Vlite.hs:1123-1170)
                kb1 = self.bin("Subtract", self.bin("BitShift", kb, self.const(1, kb)), self.const(1, kb))
                key = self.bin("BitwiseOr", shl(kb1, 3), self.bin("Subtract", kd, self.const(2, kd)))
            elif shape == 2:        # three components: ((b << 2) | (d >> 1)) << 2 | (d & ... no: | b) -- nested shifts distribute
                inner = self.bin("BitwiseOr", shl(kb, 2), self.bin("BitShift", kd, self.const(1, kd)))
                key = self.bin("BitwiseOr", shl(inner, 2), self.bin("BitShift", kb, self.const(2, kb)))
            elif shape == 3:        # not of the composite shape (an Add joins the parts): the step interpreter
                key = self.bin("Add", shl(kb, 3), kd)
            else:                   # not of the composite shape either: an offset after the shift-left
                key = self.bin("BitwiseOr", self.bin("Add", shl(kb, 3), self.const(8, kb)), kd)
            if r.random() < 0.5:
                key = self.bin("BitwiseAnd", key, self.const(255, key))
            dom = int(r.choice([256, 64]))           # 64: keys outside the pivots -> the engine must notice and fall back
            part = self.emit("Partition,val,Id %d,val,Id %d,val" % (key, self.emit("RangeC,val,0,%d,1" % dom)))
            skey = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (key, self.emit("RangeV,val,0,Id %d,1" % key), part))
        outs = []
        for _ in range(int(r.integers(1, 5))):
            t = self.term(sel)
            kind = str(r.choice(["FoldSum", "FoldSum", "FoldMin", "FoldMax", "FoldCount"] + (["FoldChoose"] if grouped else [])))
            if grouped:
                if kind == "FoldChoose":
                    t = self.emit("Gather,Id %d,Id %d,val" % (self.col(), sel))
                st = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (t, self.emit("RangeV,val,0,Id %d,1" % t), part))
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, skey, st)))
            else:
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, self.emit("RangeV,val,0,Id %d,0" % t), t)))
        if len(outs) >= 2 and r.random() < 0.5:      # avg-like scalar expression over two folds
            outs.append(self.bin("Divide", outs[0], outs[1]))
        for o in outs:
            self.emit("MaterializeCompact,Id %d" % o)
        return prog(*self.lines), self.cols


def test_generator_is_accepted_by_oracle_and_planner():
    fused = composite = interpreted = 0
    e = m.Engine(device=None)
    for seed in range(60):
        text, cols = Gen(seed).build()
        assert oracle_run(text, cols) is not None
        p = e.parse(text)
        fused += p.is_fused
        composite += "key form: composite" in p.describe()
        interpreted += "key form: general" in p.describe()
    assert fused >= 20                                # a good share of them really exercises the fused scans
    assert composite >= 5 and interpreted >= 3        # both ways of evaluating a group key


@pytest.mark.gpu
def test_random_filter_aggregate_programs_match_the_oracle():
    fused = 0
    for seed in range(250):
        text, cols = Gen(seed).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        fused += p.is_fused
        got = p.run()["results"]
        p.set_fusion(False)
        unfused = p.run()["results"]
        e.close()
        check_against_oracle("random_fused_as_planned", seed, text, cols, got, want)
        check_against_oracle("random_fused_statement_by_statement", seed, text, cols, unfused, want)
    assert fused >= 80
