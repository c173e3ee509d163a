"""Random FK-join + aggregate programs in the shape the compiler emits (handleGatherJoin / deduceMasks,
/root/reference/src/Vlite.hs:1199-1282): the dimension side reduced to a validity mask and a position vector scattered by
its filtered row ids, the fact side looking both up through its join-index column, cleaned by a FoldSelect, dimension
columns gathered through the cleaned positions; then ungrouped or dense-domain grouped aggregates.  The planner turns these
into ONE scan of the fact table with derived columns (dim_col[fk[row]], dim_bitmap[fk[row]], LIKE tables) -- or refuses and
the program runs statement by statement; either way, and with fusion switched off, the answer must be the oracle's.
Join-index values outside the dimension table occur on purpose (the row is EPS: it must not count)."""
import numpy as np
import pytest

import mplan2vdl_amd as m
from helpers import check_against_oracle, engine_with, make_heap, oracle_run, prog

WORDS = ["PROMO BRUSHED", "PROMO PLATED", "STANDARD", "MEDIUM POLISHED", "ECONOMY", "PROMOTION", "SMALL PROMO"]


class Gen:
    def __init__(self, seed):
        self.rng = r = np.random.default_rng(seed)
        self.lines, self.nid = [], 0
        nt, nu = int(r.integers(1, 30000)), int(r.integers(1, 400))
        heap, where = make_heap(WORDS)
        offs = np.array(sorted(where.values()), np.int64)
        self.cols = {"t.a": r.integers(-50, 50, nt).astype(np.int64), "t.b": r.integers(0, 30, nt).astype(np.int32),
                     "t.d": r.integers(0, 7, nt).astype(np.int16), "t.t_pkey": np.zeros(nt, np.int64),
                     "t.t_u": r.integers(-1 if seed % 3 == 0 else 0, nu + (2 if seed % 3 == 0 else 0), nt).astype(np.int64),
                     "u.x": r.integers(0, 100, nu).astype(np.int64), "u.y": r.integers(0, 5, nu).astype(np.int32),
                     "u.s": offs[r.integers(0, len(offs), nu)], "u.s.heap": heap, "u.u_pkey": np.zeros(nu, np.int64)}
        self.c = {}
        for name in ("t.a", "t.b", "t.d", "t.t_pkey", "t.t_u", "u.x", "u.y", "u.s", "u.u_pkey", "u.s.heap"):
            self.c[name] = self.emit("Project,val,Id %d,%s" % (self.emit("Load," + name), name.split(".", 1)[1]))

    def emit(self, body):
        self.nid += 1
        self.lines.append("%d,%s" % (self.nid, body))
        return self.nid

    def const(self, k, ref): return self.emit("RangeV,val,%d,Id %d,0" % (k, ref))
    def pos(self, ref): return self.emit("RangeV,val,0,Id %d,1" % ref)
    def bin(self, op, a, b): return self.emit("%s,val,Id %d,val,Id %d,val" % (op, a, b))
    def gather(self, src, p): return self.emit("Gather,Id %d,Id %d,val" % (src, p))
    def select(self, pred): return self.emit("FoldSelect,val,Id %d,val,Id %d,val" % (self.pos(pred), pred))

    def build(self):
        r = self.rng
        c = self.c
        # ---- fact-side Select (optional)
        fact_pos = self.pos(c["t.t_pkey"])
        fact_cols = {k: c[k] for k in ("t.a", "t.b", "t.d")}
        if r.random() < 0.7:
            k = self.const(int(r.integers(0, 25)), c["t.b"])
            pred = self.bin("Greater", c["t.b"], k) if r.random() < 0.5 else self.bin("Greater", k, c["t.b"])
            sel = self.select(pred)
            fact_cols = {k2: self.gather(v, sel) for k2, v in fact_cols.items()}
            fact_pos = self.gather(fact_pos, sel)
        # ---- dimension side: valid mask + positions, scattered by the (filtered) row ids (deduceMasks)
        dim_pos = self.pos(c["u.u_pkey"])
        if r.random() < 0.6:
            form = r.choice(["x", "y", "like"])
            if form == "x":
                pred = self.bin("Greater", c["u.x"], self.const(int(r.integers(0, 90)), c["u.x"]))
            elif form == "y":
                pred = self.bin("Equals", c["u.y"], self.const(int(r.integers(0, 5)), c["u.y"]))
            else:
                pred = self.emit("Like,val,Id %d,val,Id %d,val,%s" % (c["u.s"], c["u.s.heap"], str(r.choice(["PROMO%", "%PROMO%", "%ED", "S_ALL%"]))))
            sel_u = self.select(pred)
            ids = self.gather(dim_pos, sel_u)
            ones = self.const(1, ids)
            valid = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (ones, self.pos(ones), ids))
            posv = self.pos(ids)
            idx = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (posv, self.pos(posv), ids))
        else:
            valid, idx = self.const(1, c["u.u_pkey"]), dim_pos
        # ---- fact side: look both up through the join index, clean (handleGatherJoin)
        fkf = self.gather(c["t.t_u"], fact_pos)
        fvalid = self.gather(valid, fkf)
        sm = self.select(fvalid)
        fact_cols = {k2: self.gather(v, sm) for k2, v in fact_cols.items()}
        gm = self.gather(self.gather(idx, fkf), sm)
        dim_cols = {k2: self.gather(c[k2], gm) for k2 in ("u.x", "u.y", "u.s")}
        a, b, d = fact_cols["t.a"], fact_cols["t.b"], fact_cols["t.d"]
        x, y = dim_cols["u.x"], dim_cols["u.y"]
        like = self.emit("Like,val,Id %d,val,Id %d,val,%s" % (dim_cols["u.s"], c["u.s.heap"], str(r.choice(["PROMO%", "%POLISHED", "STANDARD"]))))
        # ---- aggregate inputs
        def term():
            form = r.choice(["a", "ax", "a100x", "case", "xy"])
            if form == "a": return a
            if form == "ax": return self.bin("Multiply", a, x)
            if form == "a100x": return self.bin("Multiply", a, self.bin("Subtract", self.const(100, x), x))
            if form == "xy": return self.bin("Multiply", x, self.bin("Add", self.const(3, y), y))
            # CASE WHEN like THEN a * x ELSE 0 as the emitter writes it (Vlite.hs:240-245)
            neg = self.bin("Equals", like, self.const(0, like))
            posc = self.bin("Subtract", self.const(1, like), neg)
            left = self.bin("Multiply", posc, self.bin("Multiply", a, x))
            right = self.bin("Multiply", neg, self.bin("Multiply", self.const(0, a), self.const(10000, a)))
            return self.bin("Add", left, right)

        outs = []
        grouped = r.random() < 0.5
        if grouped:
            shl = lambda v, k: self.bin("BitShift", v, self.bin("Subtract", self.const(0, v), self.const(k, v)))
            key = self.bin("BitwiseOr", shl(d, 3), y) if r.random() < 0.5 else d          # by a fact column, or fact x dimension column
            dom = 64
            part = self.emit("Partition,val,Id %d,val,Id %d,val" % (key, self.emit("RangeC,val,0,%d,1" % dom)))
            skey = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (key, self.pos(key), part))
        for _ in range(int(r.integers(1, 4))):
            t = term()
            kind = str(r.choice(["FoldSum", "FoldSum", "FoldMin", "FoldMax", "FoldCount"]))
            if grouped:
                st = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (t, self.pos(t), part))
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, skey, st)))
            else:
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, self.const(0, t), t)))
        if len(outs) >= 2 and r.random() < 0.4:
            outs.append(self.bin("Divide", outs[0], outs[1]))
        for o in outs:
            self.emit("MaterializeCompact,Id %d" % o)
        return prog(*self.lines), self.cols


def test_generator_is_accepted_and_a_good_share_fuses():
    fused = derived = 0
    e = m.Engine(device=None)
    for seed in range(80):
        text, cols = Gen(seed).build()
        assert oracle_run(text, cols) is not None
        p = e.parse(text)
        fused += p.is_fused
        derived += p.is_fused and "[col" in p.describe()
    assert fused >= 40 and derived >= 40                       # they fuse, and as JOIN scans (derived columns), not by accident


@pytest.mark.gpu
def test_random_join_aggregate_programs_match_the_oracle():
    fused = 0
    for seed in range(200):
        text, cols = Gen(seed).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        fused += p.is_fused
        got = p.run()["results"]
        p.set_fusion(False)
        unfused = p.run()["results"]
        e.close()
        check_against_oracle("random_join_as_planned(fused=%s)" % p.is_fused, seed, text, cols, got, want)
        check_against_oracle("random_join_statement_by_statement", seed, text, cols, unfused, want)
    assert fused >= 100
