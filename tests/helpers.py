"""Shared helpers for the parity tests."""
import threading

import numpy as np

from mplan2vdl_amd import datagen


def lineitem(names, n, seed=datagen.SEED, row0=0):
    return {name: datagen.generate(datagen.LINEITEM[name], row0, n, seed) for name in names}


def oracle_run(text, cols):
    import oracle

    o = oracle.Oracle()
    for k, v in cols.items():
        o.add_column(k, v)
    try:
        try:
            return o.run(text)["results"]
        except oracle.OracleError as first:
            # Evidence trap (the intermittent failures): twice in this build's history a program TEXT was seen
            # with a byte changed ("t.g" read as "t,g" by the engine's parser; "MaterializeCompact" cut after "Mate" here, i.e. a NUL
            # in the copy handed over) and was fine on the next run.  Was it the str itself, or the transient copy?  Look, and retry.
            import re
            msg = str(first)
            odd = [hex(ord(ch)) for ch in text if ord(ch) < 9 or ord(ch) > 126]
            m = re.search(r"line (\d+)", msg)
            line = text.splitlines()[int(m.group(1)) - 1] if m and int(m.group(1)) <= len(text.splitlines()) else None
            try:
                again = oracle.Oracle()
                for k, v in cols.items():
                    again.add_column(k, v)
                res = again.run(text)["results"]
                again.close()
            except oracle.OracleError:
                raise first
            raise AssertionError("INTERMITTENT ORACLE PARSE FAILURE: %r on the first attempt, accepted on the second (same str object); "
                                 "bytes outside printable ASCII in the str now: %r; the line it named, as the str holds it now: %r; result of the "
                                 "second attempt has %d outputs" % (msg, odd, line, len(res)))
    finally:
        o.close()


def engine_with(cols, device=0):
    import mplan2vdl_amd as m

    e = m.Engine(device=device)
    for k, v in cols.items():
        e.upload(k, v)
    return e


# ---- W ranks = W threads of this process, one context each on device 0 (tests/test_comm_gpu.py, tests/test_full_size.py) ----
class Rendezvous:
    """In-process stand-in for a host collective library (TEST ONLY)."""

    def __init__(self, world):
        self.world, self.barrier = world, threading.Barrier(world)
        self.slots = [None] * world

    def transport(self, rank):
        def all_gather(send):
            self.slots[rank] = send
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out

        def all_to_all(pieces):
            self.slots[rank] = pieces
            self.barrier.wait()
            out = [self.slots[src][rank] for src in range(self.world)]
            self.barrier.wait()
            return out

        return all_gather, all_to_all


def run_ranks(world, work, timeout=300):
    """work(rank, rendezvous) in `world` threads; returns the per-rank results, re-raising the first failure."""
    rv = Rendezvous(world)
    out, errs = [None] * world, []

    def body(rank):
        try:
            out[rank] = work(rank, rv)
        except BaseException as exc:          # noqa: BLE001
            errs.append(exc)
            rv.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    if errs:
        raise errs[0]
    return out


def prog(*lines):
    return "\n".join(lines) + "\n"


def rand_cols(rng, n, spec):
    """spec: {name: (dtype, lo, hi)} -> uniform random integer columns."""
    return {k: rng.integers(lo, hi + 1, size=n, dtype=np.int64).astype(dt) for k, (dt, lo, hi) in spec.items()}


def sql_q3(t):
    """TPC-H Q3 evaluated from its SQL text (/root/reference/tests/tpch10noorder/03.sql.mplan:1-19) with numpy,
    over the join-index catalog of datagen.q3_tables.  Returns the four output columns in the order the
    VDL program produces them: ascending composite key (l_orderkey, o_orderdate, o_shippriority)."""
    seg = t["customer.c_mktsegment"]
    o_date, o_prio, o_cust = t["orders.o_orderdate"], t["orders.o_shippriority"], t["orders.orders_customer"]
    l_ord, l_key = t["lineitem.lineitem_orders"], t["lineitem.l_orderkey"].astype(np.int64)
    l_ship, l_ep, l_disc = t["lineitem.l_shipdate"], t["lineitem.l_extendedprice"], t["lineitem.l_discount"]
    order_ok = (o_date < 728732) & (seg[o_cust] == 16)            # date '1995-03-15', 'BUILDING'
    li_ok = (l_ship > 728732) & order_ok[l_ord]
    rows = np.nonzero(li_ok)[0]
    key = ((l_key[rows] - 1) << 12) | (o_date[l_ord[rows]].astype(np.int64) - 727563)
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rev = np.zeros(len(uniq), np.int64)
    np.add.at(rev, inv, l_ep[rows] * (100 - l_disc[rows]))
    fr = rows[first]
    return {"l_orderkey__lineitem__l_orderkey": [int(x) for x in l_key[fr]], "revenue": [int(x) for x in rev],
            "o_orderdate__orders__o_orderdate": [int(x) for x in o_date[l_ord[fr]]],
            "o_shippriority__orders__o_shippriority": [int(x) for x in o_prio[l_ord[fr]]]}


def make_heap(strings, base=16, align=8):
    """MonetDB-style string heap: NUL-terminated strings at `align`-byte aligned offsets starting at
    `base` (the codes of the reference's dictionary.csv are such offsets: 'BUILDING' -> 16).
    Returns (heap as int8 array, {string: offset})."""
    buf = bytearray(base)
    where = {}
    for s in strings:
        if s in where:
            continue
        while len(buf) % align:
            buf.append(0)
        where[s] = len(buf)
        buf += s.encode() + b"\0"
    return np.frombuffer(bytes(buf), dtype=np.int8).copy(), where


def sql_like(s, pattern):
    """SQL LIKE ('%', '_', no escape) through a regular expression: an independent statement of the
    semantics oracle/vdl_oracle.c:like_match implements."""
    import re

    rx = "".join(".*" if ch == "%" else "." if ch == "_" else re.escape(ch) for ch in pattern)
    return 1 if re.fullmatch(rx, s, flags=re.S) else 0


# ---- statement-by-statement comparison (engine trace vs oracle vectors) -------------------------------------------
def _statement_lines(text):
    out = {}
    for ln in text.splitlines():
        head = ln.split(",", 1)[0].strip()
        if head.isdigit():
            out[int(head)] = ln.strip()
    return out


def compare_traced(plan, orc, text):
    """First statement of the last traced run whose vector differs from the oracle's (orc ran `text` with
    keep_vectors): None if every evaluated statement agrees, else a dict describing the divergence.
    One deliberate difference is tolerated: a RangeV over a gather that the engine never runs (its only reader is the
    filter idiom FoldSelect(RangeV 0 1 g, g)) lends its length and an UPPER BOUND of its validity
    (vdl_genexec.h, `lazy_gather_ok`)."""
    import numpy as np

    lines = _statement_lines(text)
    compared = 0
    for node, form, n, vals, ok in plan.traced():
        if vals is None:
            continue
        ref = orc.vector(node)
        if ref is None:
            continue
        rv, ro = ref
        line = lines.get(node, "?")
        if len(rv) != n:
            return {"statement": node, "line": line, "form": form, "why": "length %d, oracle %d" % (n, len(rv))}
        compared += 1
        bad_ok = np.nonzero(ok != ro)[0]
        if len(bad_ok) and ",RangeV," in line and not np.any(ro & ~ok):
            bad_ok = bad_ok[:0]
            ok = ro
        both = ok & ro
        bad_val = np.nonzero(both & (vals != rv))[0]
        if len(bad_ok) or len(bad_val):
            return {"statement": node, "line": line, "form": form,
                    "validity_differs_at": [int(i) for i in bad_ok[:8]], "n_validity_diffs": int(len(bad_ok)),
                    "engine_holds": [bool(ok[i]) for i in bad_ok[:8]],
                    "value_differs_at": [int(i) for i in bad_val[:8]], "n_value_diffs": int(len(bad_val)),
                    "engine": [int(vals[i]) for i in bad_val[:8]], "oracle": [int(rv[i]) for i in bad_val[:8]]}
    return None if compared else {"why": "nothing was traced"}


def explain_mismatch(tag, seed, text, cols, got, want, reruns=3):
    """A result differed from the oracle: write everything needed to study it to gpurun_out/mismatch/<tag>_<seed>.txt --
    the switches in force, the program, both result dicts -- then run the program again statement by statement with
    tracing on (`reruns` times: the failure may not show every time) and name the first statement whose vector
    differs from the oracle's.  Returns the path of the report."""
    import json
    import os
    import sys
    import tempfile

    import oracle

    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out", "mismatch")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "%s_seed%s.txt" % (tag, seed))
    with open(path, "w") as f:
        f.write("tag %s  seed %s\nswitches: %s\n" % (tag, seed, {k: v for k, v in os.environ.items() if k.startswith("VDL_")}))
        f.write("columns: %s\n" % {k: (str(v.dtype), len(v)) for k, v in cols.items()})
        f.write("---- program\n%s\n---- engine\n%s\n---- oracle\n%s\n" % (text, json.dumps(got, sort_keys=True), json.dumps(want, sort_keys=True)))
        for k in sorted(set(got) | set(want)):
            if got.get(k) != want.get(k):
                f.write("differs: %s\n  engine %s\n  oracle %s\n" % (k, got.get(k), want.get(k)))
        orc = oracle.Oracle()
        orc.keep_vectors(True)
        for k, v in cols.items():
            orc.add_column(k, v)
        orc.run(text)
        for attempt in range(reruns + 1):
            e = engine_with(cols)
            p = e.parse(text)
            as_planned = attempt == 0                      # first with the fused front (if the plan has one), then without any fusion
            p.set_fusion(as_planned)
            p.set_trace(True)
            forms = ""
            try:
                # the engine's own account of the forms it chose (VDL_TRACE_FORMS goes to the C stderr): captured at fd level
                sys.stderr.flush()
                with tempfile.TemporaryFile() as cap:
                    saved = os.dup(2)
                    os.environ["VDL_TRACE_FORMS"] = "1"
                    os.dup2(cap.fileno(), 2)
                    try:
                        again = p.run()["results"]
                    finally:
                        os.dup2(saved, 2)
                        os.close(saved)
                        os.environ.pop("VDL_TRACE_FORMS", None)
                    cap.seek(0)
                    forms = cap.read().decode(errors="replace")
                first = compare_traced(p, orc, text)
                f.write("---- traced rerun %d (%s): results %s the oracle; first diverging statement: %s\n"
                        % (attempt, "as planned" if as_planned else "statement by statement", "EQUAL" if again == want else "DIFFER FROM", json.dumps(first)))
                if again != want or first is not None or attempt == 0:
                    f.write(forms)
            finally:
                e.close()
        orc.close()
    return path


def check_against_oracle(tag, seed, text, cols, got, want):
    if got != want:
        path = explain_mismatch(tag, seed, text, cols, got, want)
        raise AssertionError("%s seed %s: engine and oracle differ; report in %s\n%s" % (tag, seed, path, open(path).read()[-6000:]))
