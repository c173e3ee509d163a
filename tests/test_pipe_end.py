"""The executor end of the reference's pipe, on the device, in the suite the driver runs.

/root/reference/eval_query.sh:18-26 is   ./tpchrun DIR plan | sed 's/;;.*//' | curl ... voodoo server | resolve.py;
here the server hop is mplan2vdl_amd/bin/vdlrun (C++ over the C ABI, nothing else) fed on stdin, and its stdout is parsed
exactly as /root/reference/resolve.py:36-62 parses the reply (json.load, ["results"], one {".name": [...]} per tmpN)
before it is compared with the oracle and decoded by mplan2vdl_amd.resolve.  Also a plain C host (examples/q6_device.c,
gcc, no C++ / Python in the process) that generates, parses, runs and reads back through include/vdl.h on the device."""
import io
import json
import os
import subprocess

import numpy as np
import pytest

from mplan2vdl_amd import catalog, datagen, frontend, resolve
from conftest import ROOT, golden
from helpers import lineitem, oracle_run

pytestmark = pytest.mark.gpu

VDLRUN = os.path.join(ROOT, "mplan2vdl_amd", "bin", "vdlrun")
META = os.path.join(ROOT, "tests", "golden", "tpch10noorder")


def pipe(text, args):
    r = subprocess.run([VDLRUN] + args, input=text.encode(), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    reply = json.load(io.BytesIO(r.stdout))                    # resolve.py:36  `data = json.load(sys.stdin)`
    assert "results" in reply                                   # resolve.py:42-43: a reply without it is a failed query
    return reply


def reference_shape(results):
    """What resolve.py:44-62 walks: every tmpN maps ONE ".name" key to a list of ints."""
    for tmp, entry in results.items():
        assert tmp.startswith("tmp") and len(entry) == 1
        (k, v), = entry.items()
        assert k.startswith(".") and all(isinstance(x, int) for x in v)


@pytest.mark.parametrize("flags", [[], ["--no-fuse"]])
@pytest.mark.parametrize("rows", [60175, 1, 0])
def test_vdlrun_rows_runs_q6_on_generated_lineitem(q6_text, rows, flags):
    """`vdlrun --rows N`: the synthetic TPC-H-shaped lineitem generated in HBM (same generator and seed as datagen)."""
    reply = pipe(q6_text, ["--rows", str(rows)] + flags)
    reference_shape(reply["results"])
    assert reply["results"] == oracle_run(q6_text, lineitem(datagen.Q6_COLUMNS, rows))


@pytest.mark.parametrize("flags", [[], ["--jit"], ["--jit-tune"]])
def test_vdlrun_rows_runs_q1_grouped(q1_text, flags):
    """(--jit / --jit-tune: the grouped scan specialised for the plan by hiprtc inside the C host, vdl_plan_set_jit)"""
    reply = pipe(q1_text, ["--rows", "60175"] + flags)
    assert reply["results"] == oracle_run(q1_text, lineitem(datagen.Q1_COLUMNS, 60175))
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "q1_sf001.json")))
    assert reply["results"] == golden["results"]                # the committed fixture, through the pipe end


@pytest.mark.parametrize("plan", [3, 12, 14, 10])
def test_vdlrun_data_dir_runs_compiled_plans_and_the_reply_decodes(tmp_path, plan):
    """compile | sed | vdlrun --data DIR | resolve: columns exported as raw little-endian files (catalog.export_columns)."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan)).read(), cfg)
    text = "\n".join(ln.split(";;")[0].rstrip() for ln in text.splitlines()) + "\n"       # eval_query.sh:20  sed 's/;;.*//'
    cols = catalog.synth_columns(META, cfg, text, scale=1e-3, seed=5)
    coldir = str(tmp_path / "cols")
    catalog.export_columns(cols, coldir)
    reply = pipe(text, ["--data", coldir])
    reference_shape(reply["results"])
    want = oracle_run(text, cols)
    assert reply["results"] == want
    assert any(len(list(v.values())[0]) for v in want.values())
    # decoded the way the pipe's last stage does it (dictionary codes back to strings, one CSV row per result row)
    names, rows = resolve.decode(reply, resolve.load_dictionary(os.path.join(META, "dictionary.csv")))
    nrows = len(list(list(want.values())[0].values())[0])
    assert len(rows) == nrows and len(names) == len(want) and all(len(r) == len(want) for r in rows)


def test_vdlrun_reports_errors_with_a_nonzero_status(q6_text):
    r = subprocess.run([VDLRUN, "--data", "/nonexistent"], input=q6_text.encode(), capture_output=True, timeout=120)
    assert r.returncode != 0 and b"columns.csv" in r.stderr
    r = subprocess.run([VDLRUN, "--rows", "100"], input=b"1,Load,nosuch.column\n2,MaterializeCompact,Id 1\n", capture_output=True, timeout=120)
    assert r.returncode != 0 and b"nosuch.column" in r.stderr


@pytest.mark.parametrize("fuse", [1, 0])
def test_plain_c_host_runs_q6_on_the_device(tmp_path, q6_text, fuse):
    exe = str(tmp_path / "q6_device")
    lib = os.path.join(ROOT, "mplan2vdl_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "q6_device.c"),
                    "-L" + lib, "-lvdl", "-Wl,-rpath," + lib, "-o", exe], check=True)
    rows = 250001
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "q6.vdl"), str(rows), str(fuse)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = json.loads(r.stdout)["results"]
    assert got == oracle_run(q6_text, lineitem(datagen.Q6_COLUMNS, rows))


@pytest.mark.parametrize("query", ["q6", "q1"])
def test_vdlrun_gpus_forks_its_ranks_and_runs_through_rccl(query):
    """`vdlrun --gpus N`: ranks forked before HIP is touched, RCCL communicator id through a file, vdl_run_sharded.  This box
    has one GPU, so N = 1 here -- the same code path with a one-rank RCCL communicator (the driver's node runs N = 8)."""
    text = golden(query + ".vdl")
    reply = pipe(text, ["--gpus", "1", "--rows", "60175"])
    names = datagen.Q6_COLUMNS if query == "q6" else datagen.Q1_COLUMNS
    assert reply["results"] == oracle_run(text, lineitem(names, 60175))


@pytest.mark.parametrize("plan,table", [(3, "lineitem"), (14, "lineitem"), (4, "lineitem"), (11, "partsupp"), (16, "partsupp"), (15, "lineitem"), (18, "lineitem")])
def test_vdlrun_gpus_with_a_data_directory(tmp_path, plan, table):
    """--gpus with --data / --shard: the exchange route (Q3), the fold-record route (Q14), the merged sets (Q4), a global fold beside the
    Partition (Q11) and the front route (Q16) from exported column files, through a one-rank RCCL communicator."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan)).read(), cfg)
    text = "\n".join(ln.split(";;")[0].rstrip() for ln in text.splitlines()) + "\n"
    cols = catalog.synth_columns(META, cfg, text, scale=1e-3, seed=5)
    coldir = str(tmp_path / "cols")
    catalog.export_columns(cols, coldir)
    reply = pipe(text, ["--gpus", "1", "--shard", table, "--data", coldir])
    assert reply["results"] == oracle_run(text, cols)


def test_vdlrun_gpus_reports_a_missing_device():
    """More ranks than devices: the ranks without a device fail, and the one that got a device is not left waiting for them."""
    r = subprocess.run([VDLRUN, "--gpus", "3", "--rows", "1000"], input=golden("q6.vdl").encode(), capture_output=True, timeout=120)
    assert r.returncode != 0 and b"rank(s) failed" in r.stderr and b"not available" in r.stderr
