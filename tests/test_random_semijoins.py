"""Random semi-join programs in the shape the compiler emits for EXISTS / IN with the dimension on the left (LeftSemi,
/root/reference/src/Vlite.hs:1212-1222; TPC-H Q4): the fact side is filtered, looks up the (filtered) dimension row ids
through its join index, scatters ones by those ids (mod N) into a fact-length vector -- duplicates land on the same slot --
and a FoldSelect over it gives the dimension rows that some fact row points at; dimension columns are gathered through
those positions, then grouped or ungrouped aggregates.  The planner turns this into a scan of the fact table that sets bits
by join index and a scan of the dimension table testing the bit of its own row id -- or refuses; either way, and with fusion
off, the oracle's answer.  Join indices outside the dimension occur on purpose; N is at least the dimension's row count
(smaller N: the engine must notice at run time and run statement by statement)."""
import numpy as np
import pytest

import mplan2vdl_amd as m
from helpers import check_against_oracle, engine_with, oracle_run, prog


class Gen:
    def __init__(self, seed, wrap=False, short_fact=False, sparse_domain=False):
        self.rng = r = np.random.default_rng(seed)
        self.lines, self.nid = [], 0
        nu = int(r.integers(2 if short_fact else 1, 3000))
        # the scattered vector has the fact table's length: normally at least as long as the dimension; with short_fact it is
        # shorter, so positions in [nt, nu) fall off its end and those dimension rows are NOT selected
        nt = int(r.integers(1, nu)) if short_fact else int(r.integers(max(nu, 2), 30000))
        self.nu, self.wrap, self.sparse_domain = nu, wrap, sparse_domain
        self.cols = {"t.a": r.integers(-50, 50, nt).astype(np.int64), "t.b": r.integers(0, 30, nt).astype(np.int32),
                     "t.t_pkey": np.zeros(nt, np.int64),
                     "t.t_u": r.integers(-1 if seed % 3 == 0 else 0, nu + (2 if seed % 3 == 0 else 0), nt).astype(np.int64),
                     "u.x": r.integers(0, 100, nu).astype(np.int64), "u.y": r.integers(0, 5, nu).astype(np.int32), "u.u_pkey": np.zeros(nu, np.int64)}
        self.c = {}
        for name in ("t.a", "t.b", "t.t_pkey", "t.t_u", "u.x", "u.y", "u.u_pkey"):
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
        r, c = self.rng, self.c
        # ---- dimension side (optional filter): its columns and its row ids, restricted; row ids scattered back to their slots
        ucols = {"x": c["u.x"], "y": c["u.y"]}
        uids = self.pos(c["u.u_pkey"])
        if r.random() < 0.6:
            pred = self.bin("Greater", c["u.x"], self.const(int(r.integers(0, 80)), c["u.x"])) if r.random() < 0.5 else \
                   self.bin("Equals", c["u.y"], self.const(int(r.integers(0, 5)), c["u.y"]))
            su = self.select(pred)
            ucols = {k: self.gather(v, su) for k, v in ucols.items()}
            uids = self.gather(uids, su)
        p0 = self.pos(uids)
        idmap = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (p0, p0, uids))              # out[id] = id for the dimension rows that are in
        # ---- fact side (optional filter), looks the dimension row ids up through the join index
        fk = c["t.t_u"]
        if r.random() < 0.7:
            pred = self.bin("Greater", c["t.a"], c["t.b"]) if r.random() < 0.4 else self.bin("Greater", c["t.b"], self.const(int(r.integers(0, 25)), c["t.b"]))
            st = self.select(pred)
            fk = self.gather(fk, self.gather(self.pos(c["t.t_pkey"]), st))
        # the dimension row id, EPS where there is none -- or (not what the compiler emits, but legal text) the raw join index:
        # negative ones are never scattered, those beyond N wrap around
        hit = self.gather(idmap, fk) if r.random() < 0.75 else fk
        ones = self.const(1, hit)
        n_mod = self.nu + int(r.integers(0, 50)) if not self.wrap else max(1, self.nu - int(r.integers(1, 5)))
        where = self.bin("Modulo", hit, self.const(n_mod, hit))
        marks = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (ones, self.pos(ones), where))  # duplicates: many fact rows per dimension row
        sel = self.select(marks)
        x, y = self.gather(ucols["x"], sel), self.gather(ucols["y"], sel)
        outs = []
        if r.random() < 0.5 or self.sparse_domain:
            # (a sparse pivot domain keeps the plan from fusing as a whole: the semi-join set then belongs to a fused FRONT)
            part = self.emit("Partition,val,Id %d,val,Id %d,val" % (y, self.emit("RangeC,val,0,%d,1" % (1 << 30 if self.sparse_domain else 8))))
            sy = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (y, self.pos(y), part))
            outs.append(self.emit("FoldChoose,val,Id %d,val,Id %d,val" % (sy, sy)))
            for kind in [str(k) for k in r.choice(["FoldSum", "FoldMin", "FoldMax", "FoldCount"], int(r.integers(1, 3)))]:
                sx = self.emit("Scatter,Id %d,Id %d,val,Id %d,val" % (x, self.pos(x), part))
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, sy, sx)))
        else:
            for kind in [str(k) for k in r.choice(["FoldSum", "FoldMin", "FoldMax", "FoldCount"], int(r.integers(1, 3)))]:
                t = x if r.random() < 0.6 else self.bin("Multiply", x, self.bin("Add", y, self.const(3, y)))
                outs.append(self.emit("%s,val,Id %d,val,Id %d,val" % (kind, self.const(0, t), t)))
        for o in outs:
            self.emit("MaterializeCompact,Id %d" % o)
        return prog(*self.lines), self.cols


def test_generator_is_accepted_and_fuses_as_a_semi_join():
    fused = semi = 0
    reasons = {}
    e = m.Engine(device=None)
    for seed in range(80):
        text, cols = Gen(seed).build()
        assert oracle_run(text, cols) is not None
        p = e.parse(text)
        d = p.describe()
        fused += p.is_fused
        semi += p.is_fused and "semi-join set" in d
        if not p.is_fused:
            reasons[d.split("\n")[0][:80]] = reasons.get(d.split("\n")[0][:80], 0) + 1
    assert fused >= 60 and semi >= 60, (fused, semi, reasons)


@pytest.mark.gpu
def test_random_semi_join_programs_match_the_oracle():
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
        check_against_oracle("random_semijoin_as_planned", seed, text, cols, got, want)
        check_against_oracle("random_semijoin_statement_by_statement", seed, text, cols, unfused, want)
    assert fused >= 150


@pytest.mark.gpu
def test_a_modulus_smaller_than_the_dimension_is_noticed_at_run_time():
    """positions mod N with N below the dimension's row count wrap around: the fused form would be wrong, the engine abandons it
    for this run (timings say so) and the answer is still the oracle's"""
    abandoned = 0
    for seed in range(30):
        text, cols = Gen(seed, wrap=True).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        out = p.run()
        e.close()
        check_against_oracle("random_semijoin_wrapping", seed, text, cols, out["results"], want)
        abandoned += any("fusedPlanAbandoned" in k and "semi-join" in k for k in out["timings"])
    assert abandoned >= 10


@pytest.mark.gpu
def test_a_wrapping_modulus_under_a_fused_front_falls_back_to_statements():
    """the same with a sparse Partition domain above the semi-join: the plan does not fuse as a whole, the set belongs to the
    fused front (run_projection), and a wrapping modulus must abandon the front for that run -- not fail the run"""
    abandoned = fronts = 0
    for seed in range(30):
        text, cols = Gen(seed, wrap=True, sparse_domain=True).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        fronts += (not p.is_fused) and "semi-join set" in p.describe()
        out = p.run()
        e.close()
        check_against_oracle("random_semijoin_wrapping_front", seed, text, cols, out["results"], want)
        abandoned += any("fusedPlanAbandoned" in k and "semi-join" in k for k in out["timings"])
    assert fronts >= 10 and abandoned >= 10, (fronts, abandoned)


@pytest.mark.gpu
def test_a_fact_table_shorter_than_the_dimension_drops_positions_beyond_it():
    """Scatter(ones, pos(ones), fk mod N) is as long as the FACT table: with fewer fact rows than dimension rows the positions in
    [n_fact, n_dim) fall off its end (oracle/vdl_oracle.c Scatter; the statement-by-statement path), so the fused set must not
    hold them either"""
    fused = 0
    for seed in range(120):
        for sparse in (False, True):
            text, cols = Gen(seed, short_fact=True, sparse_domain=sparse).build()
            want = oracle_run(text, cols)
            e = engine_with(cols)
            p = e.parse(text)
            fused += p.is_fused
            got = p.run()["results"]
            p.set_fusion(False)
            unfused = p.run()["results"]
            e.close()
            check_against_oracle("random_semijoin_short_fact_as_planned", seed, text, cols, got, want)
            check_against_oracle("random_semijoin_short_fact_statement_by_statement", seed, text, cols, unfused, want)
    assert fused >= 80
