"""Value pins for every TPC-H plan the front end compiles (the reference holds no result vectors).

Three layers:
  1. tests/sql_eval.py + tests/golden/make_plan_goldens.py evaluate the SQL text of each query with numpy over the
     coherent synthetic catalog -- independent of VDL, the oracle and the engine;
  2. the oracle running the compiled program must produce those rows (checked here at further scales / seeds, and
     before any fixture under tests/golden/plans/ was written);
  3. the committed fixtures (oracle output, SQL-checked) must come out of the GPU engine bit for bit (`-m gpu`).
Where a plan's program does not mean its SQL, the cause is a reference compiler bug, named and isolated:
all RangeC share one identity (Q16 / Q18: compile_plan(distinct_rangec=True) removes it), and the anti-join of Q16 keeps
the rows it should drop (sql_eval.q16(as_compiled=True) states what it keeps)."""
import glob
import hashlib
import json
import os
import sys

import pytest

from conftest import ROOT
from helpers import check_against_oracle, engine_with, oracle_run

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_plan_goldens as mk  # noqa: E402
import sql_eval  # noqa: E402
from mplan2vdl_amd import frontend  # noqa: E402

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "plans", "q*.json")))


@pytest.fixture(scope="module")
def cfg():
    return frontend.load_metadata(mk.META)


def load(path, cfg):
    fx = json.load(open(path))
    n = int(fx["plan"][:2])
    text, db, cols = mk.build(n, fx["variant"] == "distinct_rangec", cfg)
    assert (fx["scale"], fx["seed"]) == (mk.SCALE, mk.SEED)
    return fx, n, text, db, cols


def test_every_compiled_plan_has_a_fixture_and_selects_rows():
    names = {os.path.basename(p) for p in FIXTURES}
    assert names == {"q%02d.json" % n for n in mk.PLANS} | {"q16_distinct_rangec.json", "q18_distinct_rangec.json"}
    for p in FIXTURES:
        fx = json.load(open(p))
        assert fx["rows"] >= 1, p                                     # no vacuous parity: every plan returns something
        assert fx["pinned_by_sql"] or os.path.basename(p) == "q18.json"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_fixture_is_what_the_oracle_computes_for_the_program_the_front_end_emits(cfg, path):
    fx, n, text, db, cols = load(path, cfg)
    assert hashlib.sha256(text.encode()).hexdigest() == fx["program_sha256"]      # the fixture belongs to THIS program text
    assert oracle_run(text, cols) == fx["results"]


@pytest.mark.parametrize("scale,seed", [(2e-4, 1), (1e-3, 7)])
@pytest.mark.parametrize("n", mk.PLANS)
def test_oracle_equals_sql_at_other_scales(cfg, n, scale, seed):
    text = frontend.compile_plan(open(os.path.join(mk.META, "%02d.sql.mplan" % n)).read(), cfg, distinct_rangec=n in mk.NEEDS_DISTINCT_RANGEC)
    db = sql_eval.Db(mk.META, cfg, text, scale, seed)
    assert sql_eval.rows_of(oracle_run(text, dict(db.cols))) == mk.SQL[n](db)


def test_q16_as_written_differs_from_what_the_reference_compiles(cfg):
    """The anti-join bug is real: the query as written keeps other rows than the compiled program."""
    text = frontend.compile_plan(open(os.path.join(mk.META, "16.sql.mplan")).read(), cfg, distinct_rangec=True)
    db = sql_eval.Db(mk.META, cfg, text, 1e-3, 7)
    got = sql_eval.rows_of(oracle_run(text, dict(db.cols)))
    assert got == sql_eval.q16(db, as_compiled=True) and got != sql_eval.q16(db, as_compiled=False)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [None, "VDL_SPARSE_ALWAYS", "VDL_NO_SPARSE", "VDL_NO_PROJECTION", "VDL_NO_GROUP_BATCH"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_engine_reproduces_the_committed_fixtures(cfg, monkeypatch, path, mode):
    if mode:
        monkeypatch.setenv(mode, "1")
    fx, n, text, db, cols = load(path, cfg)
    e = engine_with(cols)
    got = e.run_vdl(text)["results"]
    e.close()
    check_against_oracle("fixture_%s_%s" % (os.path.basename(path)[:-5], mode or "default"), 0, text, cols, got, fx["results"])
