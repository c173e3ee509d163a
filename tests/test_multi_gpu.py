"""The RCCL legs with MORE THAN ONE GPU -- run only where the box has at least two devices (the pool's one-GPU boxes skip
them; an 8-GPU node validates them without anyone asking).  Everything here goes through the product's own entry points:
`vdlrun --gpus N` (C++ host over the C ABI: ranks forked before HIP is touched, communicator id through a file,
vdl_run_sharded) and `bench.py --gpus N` (one process per GPU, RCCL inside libvdl).  A communicator that does not report N
ranks is an error in both (vdl_comm_init / bench.py exit non-zero).

Routes: Q6 / Q1 = ONE all-gather of the partial words + merge kernel; Q3 = one count all-gather + ONE grouped send/receive of
the surviving rows by key range (the all-to-all of BASELINE.json config 5); Q14 = fold records merged the same way.
torch.cuda.device_count() does not initialise the GPU in this process (the ranks are child processes)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from mplan2vdl_amd import catalog, datagen, frontend
from conftest import ROOT, golden
from helpers import lineitem, oracle_run
from test_pipe_end import META, VDLRUN, pipe

N_DEV = torch.cuda.device_count()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_DEV < 2, reason="needs at least two GPUs (this box has %d)" % N_DEV)]
WORLDS = [w for w in (2, 4, 8) if w <= N_DEV]


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("query", ["q6", "q1"])
def test_vdlrun_gpus_merges_the_partial_words_over_rccl(query, world):
    text = golden(query + ".vdl")
    rows = 600175
    reply = pipe(text, ["--gpus", str(world), "--rows", str(rows)])
    names = datagen.Q6_COLUMNS if query == "q6" else datagen.Q1_COLUMNS
    assert reply["results"] == oracle_run(text, lineitem(names, rows))


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("plan,table", [(3, "lineitem"), (14, "lineitem"), (10, "lineitem"), (4, "lineitem"), (11, "partsupp"), (16, "partsupp"), (15, "lineitem"), (18, "lineitem")])
def test_vdlrun_gpus_exchanges_rows_over_rccl(tmp_path, plan, table, world):
    """Q3 / Q10: the Partition exchange (rows by key range, outputs of the ranks concatenate in rank order); Q14: fold records; Q4: merged
    semi-join sets; Q11: a global fold beside the Partition; Q16 / Q15: the front's survivors gathered, the tail on every rank; Q18: the chain
    (exchange up to its position set, the positions gathered, the second scan's survivors gathered)."""
    cfg = frontend.load_metadata(META)
    text = frontend.compile_plan(open(os.path.join(META, "%02d.sql.mplan" % plan)).read(), cfg)
    text = "\n".join(ln.split(";;")[0].rstrip() for ln in text.splitlines()) + "\n"
    cols = catalog.synth_columns(META, cfg, text, scale=2e-3, seed=5)
    coldir = str(tmp_path / "cols")
    catalog.export_columns(cols, coldir)
    reply = pipe(text, ["--gpus", str(world), "--shard", table, "--data", coldir])
    want = oracle_run(text, cols)
    assert reply["results"] == want
    assert any(len(list(v.values())[0]) for v in want.values())


def bench(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("world", WORLDS)
def test_bench_q6_shards_over_rccl_and_verifies(world):
    """`bench.py --gpus N --rows 2e8`: row-range shards, the collective inside libvdl on its own RCCL communicator of exactly N
    ranks (anything else exits non-zero), revenue bit-exact against the SQL loop over the regenerated rows."""
    out = bench(["--gpus", str(world), "--rows", "200000000", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"])
    assert out["n_gpus"] == world and out["verified_bit_exact_vs_cpu"] is True
    assert "RCCL inside libvdl" in out["config"]["finalise"], out["config"]
    assert out["roofline"]["frac"] <= 1.0


@pytest.mark.parametrize("world", WORLDS)
def test_bench_q3_exchanges_over_rccl_and_verifies(world):
    out = bench(["--gpus", str(world), "--query", "q3", "--rows", "80000000", "--steps", "5", "--warmup", "2"])
    assert out["n_gpus"] == world and out["verified_vs_torch_sql_checksums"] is True
    assert "RCCL inside libvdl" in out["config"]["exchange"], out["config"]
