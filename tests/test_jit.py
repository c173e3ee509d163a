"""Run-time specialisation of the fused scans (vdl_plan_set_jit, csrc/vdl_jit.cpp): the scan kernels' own device code built
by hiprtc with one plan's descriptor as compile-time constants.  Building needs no GPU (hiprtc cross-compiles), so the CPU
suite checks that every kind of fused plan yields a kernel that compiles; the GPU suite checks that the specialised kernels
give the oracle's answers -- TPC-H plans, random filter / join / condition programs, tuned and untuned, sharded."""
import os

import numpy as np
import pytest

import mplan2vdl_amd as m
from mplan2vdl_amd import catalog, frontend
from conftest import ROOT
from helpers import check_against_oracle, engine_with, oracle_run

META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")
FUSED_PLANS = [1, 4, 12, 14, 19]              # multi-aggregate scans (Q6's single-aggregate scan runs on k_scan, which is not specialised)


def compiled(n, scale, seed=3):
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % n)).read(), cfg)
    return text, catalog.synth_columns(META, cfg, text, scale=scale, seed=seed)


def host_engine_with_declared(cols):
    """No device: the columns are only declared (address, width, length) -- enough to bind a plan's scans and build them."""
    e = m.Engine(device=None)
    for k, v in cols.items():
        e.register_pointer(k, 0x10000, v.dtype.itemsize, len(v))
    return e


def code_bytes(note):
    import re
    return [int(x) for x in re.findall(r"(\d+) B of code", note)]


@pytest.mark.parametrize("n", FUSED_PLANS)
def test_specialised_kernels_of_the_tpch_plans_build_without_a_gpu(n, tmp_path, monkeypatch):
    monkeypatch.setenv("VDL_JIT_CACHE", str(tmp_path))
    text, cols = compiled(n, 1e-4)
    e = host_engine_with_declared(cols)
    p = e.parse(text)
    note = p.jit_check()
    assert "k_mscan_specialised<" in note and "B of code" in note, note
    # the descriptor folded: straight-line code of 10-25 KB (a build in which it did not -- the descriptor then lives in scratch
    # memory and every loop over it stays -- was 230-360 KB and slower than the precompiled kernel)
    assert code_bytes(note) and max(code_bytes(note)) < 64 << 10, note
    # ... and the form that reads late, with the filter columns staged (the tuner's other candidate)
    monkeypatch.setenv("VDL_JIT_LATE", "1")
    monkeypatch.setenv("VDL_JIT_ASSUME_SELECTIVITY", "0.2")
    late = p.jit_check()
    assert "(late)" in late and max(code_bytes(late)) < 64 << 10, late
    # ... and its census build (vdl_plan_scan_traffic: the same form counting the 128-byte lines its late loads ask for)
    monkeypatch.setenv("VDL_JIT_CENSUS", "1")
    census = p.jit_check()
    assert "(late)" in census and code_bytes(census) != code_bytes(late) and max(code_bytes(census)) < 96 << 10, census
    monkeypatch.delenv("VDL_JIT_CENSUS")
    # ... and the queue form (one filter column with the tile, the rows still in queued per wave and finished 64 at a time), with its census build
    monkeypatch.setenv("VDL_JIT_LATE", "3")
    queue = p.jit_check()
    assert "(queue)" in queue and max(code_bytes(queue)) < 64 << 10, queue
    monkeypatch.setenv("VDL_JIT_CENSUS", "1")
    assert "(queue)" in p.jit_check()
    monkeypatch.delenv("VDL_JIT_CENSUS")
    monkeypatch.delenv("VDL_JIT_LATE")
    monkeypatch.delenv("VDL_JIT_ASSUME_SELECTIVITY")
    assert ("derived" in note) == (n != 1)                      # the join scans carry looked-up / condition columns
    kept = [f for f in os.listdir(tmp_path) if f.endswith(".vdlco")]           # the code object is kept for the next process
    assert kept and all(os.stat(os.path.join(tmp_path, f)).st_mode & 0o777 == 0o600 for f in kept)
    assert p.jit_check() == note                                # ... and for this one (no second compile: same text, same key)


def test_the_code_object_cache_is_only_used_when_nobody_else_can_write_it(tmp_path):
    """Cached code objects run against the process's GPU memory: a directory that another user could have prepared (group- or
    world-writable, reached through a symbolic link, not ours) is never read or written; missing directories are created 0700;
    an entry whose header does not match the source about to be compiled is ignored and replaced.  Every build runs in a
    process of its own (a process keeps what it built in memory and would not look at the disk twice)."""
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); os.environ['VDL_JIT_CACHE'] = sys.argv[1]; os.environ['VDL_JIT_GROUP_U'] = sys.argv[2];"
            "import test_jit as t; tx, c = t.compiled(1, 1e-4); e = t.host_engine_with_declared(c); n = e.parse(tx).jit_check();"
            "assert 'k_mscan_specialised<' in n, n") % (ROOT, os.path.join(ROOT, "tests"))

    def build(cache, u="2"):
        r = subprocess.run([sys.executable, "-c", code, str(cache), u], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return sorted(os.listdir(cache)), r.stderr

    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    files, err = build(shared)
    assert files == [] and "writable by group or others" in err            # built, nothing written there
    real, link = tmp_path / "real", tmp_path / "link"
    real.mkdir(mode=0o700)
    os.symlink(real, link)
    files, err = build(link)
    assert files == [] and "symbolic links are not followed" in err
    fresh = tmp_path / "a" / "b" / "cache"
    (name,), err = build(fresh)
    assert name.endswith(".vdlco") and "not used" not in err
    for d in (tmp_path / "a", tmp_path / "a" / "b", fresh):
        assert os.stat(d).st_mode & 0o777 == 0o700
    good = open(fresh / name, "rb").read()
    (other,), _ = build(tmp_path / "other", u="3")                         # a valid entry -- of another kernel
    planted = open(tmp_path / "other" / other, "rb").read()
    assert other != name and planted != good
    victim = tmp_path / "victim"
    victim.mkdir(mode=0o700)
    with open(victim / name, "wb") as f:
        f.write(planted)
    os.chmod(victim / name, 0o600)
    build(victim)
    assert open(victim / name, "rb").read() == good                        # not loaded: compiled again and replaced
    with open(victim / name, "wb") as f:                                   # an entry somebody else could rewrite is not read either
        f.write(planted)
    os.chmod(victim / name, 0o666)
    build(victim)
    assert open(victim / name, "rb").read() == good and os.stat(victim / name).st_mode & 0o777 == 0o600


def test_specialised_kernels_of_random_programs_build_without_a_gpu():
    from test_random_conditions import Gen as CondGen
    from test_random_fused import Gen as FusedGen
    built = 0
    for gen, seeds in ((FusedGen, range(6)), (CondGen, range(6))):
        for seed in seeds:
            text, cols = gen(seed).build()
            e = host_engine_with_declared(cols)
            p = e.parse(text)
            if not p.is_fused:
                continue
            note = p.jit_check()
            assert "not specialised (" not in note, (seed, note)
            built += "k_mscan_specialised<" in note
    assert built >= 6


@pytest.mark.parametrize("n", [3, 5, 9, 10, 11, 15, 16, 20])
def test_specialised_fronts_build_without_a_gpu(n):
    """Plans with a fused front: the one-pass front (both descriptors in one kernel) and the dimension scans of the prelude."""
    text, cols = compiled(n, 1e-4)
    e = host_engine_with_declared(cols)
    p = e.parse(text)
    note = p.jit_check()
    assert "front: vdl_jit_project_front<" in note, note
    assert ("dim" in note) == (n != 15), note
    assert max(code_bytes(note)) < 64 << 10, note


def test_a_plan_without_fused_scans_is_refused():
    text, cols = compiled(18, 1e-4)                            # Q18: neither fused nor a fused front
    e = host_engine_with_declared(cols)
    p = e.parse(text)
    with pytest.raises(m.VdlError, match="no fused scans"):
        p.jit_check()


@pytest.mark.gpu
@pytest.mark.parametrize("tune,scale", [(False, 2e-3), (True, 2e-3), (True, 3e-2)])
def test_specialised_tpch_plans_match_the_oracle(tune, scale):
    """(3e-2: 1.8 M lineitems -- thousands of tiles per scan, every block busy, the tuner's timings mean something)"""
    for n in FUSED_PLANS + [6]:
        text, cols = compiled(n, scale)
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        plain = p.run()["results"]
        p.set_jit(True, tune=tune)
        got = p.run()["results"]
        again = p.run()["results"]
        note = p.jit_note()
        p.set_jit(False)
        back = p.run()["results"]
        e.close()
        assert plain == want and got == want and again == want and back == want, (n, tune, note)
        if n != 6:
            assert "k_mscan_specialised<" in note and "not specialised" not in note, note
            assert ("tuned:" in note) == tune, note


@pytest.mark.gpu
@pytest.mark.parametrize("late", [0, 1, 2, 3])
def test_specialised_random_programs_match_the_oracle(late, monkeypatch):
    """late: staged reads -- that many filter columns with the tile, the other table columns for the rows still in; 3 = the queue
    form: one filter column with the tile, the rows still in queued per wave and finished 64 at a time with every lane busy
    (VDL_JIT_LATE forces what the tuner otherwise decides by timing)."""
    from test_random_conditions import Gen as CondGen
    from test_random_fused import Gen as FusedGen
    from test_random_joins import Gen as JoinGen
    if late:
        monkeypatch.setenv("VDL_JIT_LATE", str(late))
        monkeypatch.setenv("VDL_JIT_ASSUME_SELECTIVITY", "0.3")      # every filter counts as selective: the staged-filter code runs wherever a scan has two

    ran = lates = 0
    for tag, gen in (("fused", FusedGen), ("joins", JoinGen), ("conditions", CondGen)):
        for seed in range(40):
            text, cols = gen(seed).build()
            e = engine_with(cols)
            p = e.parse(text)
            if not p.is_fused:
                e.close()
                continue
            want = oracle_run(text, cols)
            p.set_jit(True)
            got = p.run()["results"]
            note = p.jit_note()
            e.close()
            check_against_oracle("specialised_random_" + tag, seed, text, cols, got, want)
            ran += "k_mscan_specialised<" in note
            lates += (",queue" if late == 3 else ",late") in note
    assert ran >= 60 and (lates >= (20 if late == 3 else 30) if late else lates == 0), (ran, lates)


@pytest.mark.gpu
@pytest.mark.parametrize("late", [0, 1, 2, 3])
def test_scan_traffic_counts_the_lines_a_staged_scan_asks_for(late, monkeypatch, q6_text):
    """vdl_plan_scan_traffic (bench.py's `roofline.achieved` is built on it): a scan that reads everything moves its algorithmic
    bytes; a staged scan moves the eager columns plus 128 B per line of a late column in which a row was still in -- the census
    build's count equals a numpy count over the same columns, line by line (16 int64 rows per line)."""
    from mplan2vdl_amd import datagen
    from helpers import lineitem
    monkeypatch.setenv("VDL_JIT_U", "2")
    n = 1024 * 301                                               # whole tiles of 256 lanes x 2 rows x 2
    cols = lineitem(datagen.Q6_COLUMNS, n)
    e = engine_with(cols)
    p = e.parse(q6_text)
    if late:
        monkeypatch.setenv("VDL_JIT_LATE", str(late))
        p.set_jit(True)
    assert p.run()["results"] == oracle_run(q6_text, cols)
    moved, detail = p.scan_traffic()
    assert p.run()["results"] == oracle_run(q6_text, cols)       # the census launch leaves the plan as it was
    e.close()
    if not late:
        assert moved == 28 * n and "every column read with the tile" in detail, detail
        return
    d, disc, q, x = (cols["lineitem." + c] for c in ("l_shipdate", "l_discount", "l_quantity", "l_extendedprice"))
    lines = lambda alive: int(alive.reshape(-1, 16).any(axis=1).sum())
    a0 = (d >= 728294) & (d <= 728658)                           # 1994-01-01 <= shipdate < 1995-01-01 (tests/golden/q6.vdl:5-11)
    a1 = a0 & (disc >= 5) & (disc <= 7)
    a2 = a1 & (q < 2400)
    want = 4 * n + (8 * n if late == 2 else 128 * lines(a0)) + 128 * lines(a1) + 128 * lines(a2)
    if late == 3:                                                # the queue form: every other column for the rows inside the date range
        want = 4 * n + 3 * 128 * lines(a0)
    assert moved == want, (moved, want, detail)
    assert want < 28 * n and "late:" in detail


@pytest.mark.gpu
def test_specialised_scans_shard_like_the_precompiled_ones():
    from test_comm_gpu import lineitem_shards, run_ranks
    text, cols = compiled(1, 2e-3)
    want = oracle_run(text, cols)
    shards = lineitem_shards(cols, 2)

    def work(rank, rv):
        r0, part = shards[rank]
        e = engine_with(part)
        e.comm_init_host(rank, 2, *rv.transport(rank))
        p = e.parse(text)
        p.set_jit(True)
        p.set_row_offset(r0)
        res = p.run_sharded()["results"]
        note = p.jit_note()
        e.close()
        return res, note

    for res, note in run_ranks(2, work):
        assert res == want and "k_mscan_specialised<" in note


@pytest.mark.gpu
def test_specialised_fronts_and_dimension_scans_match_the_oracle():
    """Plans that do not fuse as a whole: the one-pass front and the dimension-side bitmap scans are specialised too (roles
    front / dim<k> in the note).  Every plan runs three times: the first run has no survivor count to go by (it takes the table's
    length as the capacity or counts first), the later ones guess an eighth more than last time."""
    for n in (3, 5, 9, 10, 11, 15, 16, 20):
        text, cols = compiled(n, 2e-3)
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        p.set_jit(True)
        got = p.run()["results"]
        again = p.run()["results"]
        third = p.run()["results"]
        note = p.jit_note()
        e.close()
        assert got == want and again == want and third == want, (n, note)
        assert "front: vdl_jit_project_front<" in note and "not specialised" not in note, (n, note)
        assert ("dim" in note) == (n != 15), (n, note)          # Q15's front has no dimension side
    from test_random_conditions import FrontGen
    fronts = 0
    for seed in range(30):
        text, cols = FrontGen(seed).build()
        want = oracle_run(text, cols)
        e = engine_with(cols)
        p = e.parse(text)
        p.set_jit(True)
        got = p.run()["results"]
        note = p.jit_note()
        e.close()
        check_against_oracle("specialised_front", seed, text, cols, got, want)
        fronts += "vdl_jit_project_front<" in note
    assert fronts >= 15


@pytest.mark.gpu
def test_without_hiprtc_the_precompiled_kernels_run_and_the_note_says_so():
    """vdlrun --jit in a process that cannot load libhiprtc (VDL_HIPRTC_LIB names nothing): same answer, through the
    precompiled kernels; the library says why in the plan's note (printed by --describe-after-run on stderr)."""
    import json
    import subprocess
    from helpers import lineitem
    from mplan2vdl_amd import datagen
    text = open(os.path.join(ROOT, "tests", "golden", "q1.vdl")).read()
    vdlrun = os.path.join(ROOT, "mplan2vdl_amd", "bin", "vdlrun")
    env = dict(os.environ, VDL_HIPRTC_LIB="/nonexistent/libhiprtc.so")
    r = subprocess.run([vdlrun, "--rows", "60175", "--jit", "--profile"], input=text.encode(), capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    reply = json.loads(r.stdout)
    assert reply["results"] == oracle_run(text, lineitem(datagen.Q1_COLUMNS, 60175))
    assert any("k_mscan<" in k for k in reply["timings"]), reply["timings"]          # the precompiled grouped scan ran
