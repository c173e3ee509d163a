"""The Python restatement of the mplan -> VDL compiler against the reference's own golden lines
(README.md:40-52), the hand-derived fixtures of SURVEY.md section 8(c), and the reference's test plans
(tests/golden/tpch10noorder/ = data files of /root/reference/tests/tpch10noorder)."""
import glob
import os

import pytest

from conftest import ROOT, golden
from mplan2vdl_amd import frontend
from test_fixtures import README_HEAD, README_TAIL

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")


@pytest.fixture(scope="module")
def cfg():
    return frontend.load_metadata(META)


def compile_q(cfg, n, **kw):
    return frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg, **kw)


def test_q6_reproduces_readme_lines_and_hand_derived_fixture(cfg):
    lines = compile_q(cfg, 6).split("\n")
    assert lines[:9] == README_HEAD and lines[-3:] == README_TAIL and len(lines) == 42
    assert lines == golden("q6.vdl").strip().split("\n")


def test_q1_equals_hand_derived_fixture(cfg):
    # two independent derivations (by hand in SURVEY.md, by this compiler) agree on all 101 statements,
    # including the FoldSum lines the reference emits twice (CSE keyed on metadata, Vdl.hs:302,314-320)
    assert compile_q(cfg, 1).split("\n") == golden("q1.vdl").strip().split("\n")


def test_q3_fixture_is_what_the_compiler_emits(cfg):
    out = compile_q(cfg, 3)
    assert out.split("\n") == golden("q3.vdl").strip().split("\n")
    assert "86,RangeC,val,0,274877906944,1" in out            # 2^38 group domain: sparse partition
    assert "33,Load,orders.orders_customer" in out and "43,Load,lineitem.lineitem_orders" in out   # FK join indices
    assert ",728732," in out                                  # date '1995-03-15'
    assert "25,RangeV,val,16,Id 24,0" in out                  # 'BUILDING' -> 16 via dictionary.csv:74


def test_which_tpch_plans_compile(cfg):
    ok = []
    for f in sorted(glob.glob(os.path.join(META, "*.mplan"))):
        try:
            frontend.compile_plan(open(f).read(), cfg)
            ok.append(int(os.path.basename(f)[:2]))
        except frontend.FrontendError:
            pass
    assert ok == [1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20]


def test_every_emitted_program_parses_in_the_engine(cfg):
    import mplan2vdl_amd as m
    from mplan2vdl_amd import _lib

    e = m.Engine(device=None)
    for n in (1, 3, 4, 5, 6, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20):      # 9, 14, 16, 20 carry LIKE predicates
        assert e.parse(compile_q(cfg, n)).describe()
    vcfg = frontend.load_metadata(META, format="vlite")       # the VLite dialect parses too
    for n in (1, 3, 6, 14):
        assert "Semisort" in e.parse(frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), vcfg)).describe()


def test_metadata_suffix_and_flags(cfg):
    cfgm = frontend.load_metadata(META, show_metadata=True)
    lines = compile_q(cfgm, 6).split("\n")
    assert " ;; Metadata {databounds = (728294,728294)" in lines[4]        # the date literal knows its bounds
    assert " ;; " not in lines[0]                                          # Loads carry no metadata (Vdl.hs:168)
    assert [l.split(" ;;")[0] for l in lines] == golden("q6.vdl").strip().split("\n")
    unclean = compile_q(cfg, 6, apply_passes=False).split("\n")
    assert len(unclean) > 42                                  # -c: nested ranges, `& 0` masks and identity scatters stay
    shuffled = frontend.compile_plan(open(os.path.join(META, "01.sql.mplan")).read(),
                                     frontend.load_metadata(META, aggregation_strategy=("AggShuffle",)))
    assert ",Shuffle,Id " in shuffled


def test_cli_prints_the_program(capsys):
    from mplan2vdl_amd.frontend.__main__ import main

    main([META, os.path.join(META, "06.sql.mplan")])
    assert capsys.readouterr().out.strip().split("\n") == golden("q6.vdl").strip().split("\n")


def test_date_arithmetic_and_errors(cfg):
    from mplan2vdl_amd.frontend import mplan

    assert mplan.day_count("1994-01-01") == 728294
    with pytest.raises(frontend.FrontendError):
        frontend.compile_plan("frobnicate (\n table(sys.lineitem) [ lineitem.l_tax NOT NULL ] COUNT\n) [ lineitem.l_tax ]", cfg)
    with pytest.raises(frontend.FrontendError):
        frontend.compile_plan("project (\n table(sys.nosuch) [ nosuch.x ] COUNT\n) [ nosuch.x ]", cfg)
