"""mplan -> VDL front end: a Python restatement of the reference compiler pipeline
(/root/reference/src/MainFuns.hs:172-188: parse -> Mplan -> [pushFKJoins . fuseSelects] -> Vlite
-> cleanup passes -> Vdl text), so that the pipeline `./tpchrun DIR plan | vdlrun` runs without GHC.

    from mplan2vdl_amd import frontend
    cfg = frontend.load_metadata("/path/to/tests/tpch10noorder")
    vdl_text = frontend.compile_plan(open(".../06.sql.mplan").read(), cfg)
"""
import os

from . import mplan as _mplan
from . import parse as _parse
from . import vdl as _vdl
from . import vlite as _vlite
from .config import FrontendError, load_config  # noqa: F401


def load_metadata(directory, **flags):
    """`tpchrun DIR ...`: DIR/bounds.csv, schema.msqldump, storage.csv, dictionary.csv (tpchrun:4)."""
    j = os.path.join
    return load_config(j(directory, "bounds.csv"), j(directory, "storage.csv"), j(directory, "schema.msqldump"),
                       j(directory, "dictionary.csv"), **flags)


def compile_plan(plan_text, config, apply_passes=True, push_joins=False, distinct_rangec=False):
    """`distinct_rangec` switches off one reference bug (see vlite.DISTINCT_RANGEC); the default is bug-compatible."""
    tree = _parse.parse_mplan(_parse.filter_comments(plan_text))
    rel = _mplan.solve(config, tree)
    if push_joins:
        rel = _mplan.fuse_selects(_mplan.push_fk_joins(rel))
    saved = _vlite.DISTINCT_RANGEC
    _vlite.DISTINCT_RANGEC = bool(distinct_rangec)
    try:
        vexps = _vlite.vexps_from_mplan(rel, config, apply_passes=apply_passes)
        return _vdl.vdl_from_vexps(vexps, config)
    finally:
        _vlite.DISTINCT_RANGEC = saved
