"""bench.py's launcher contract, checked without a GPU: `python bench.py --gpus N` started bare must itself start N
ranks (one process per GPU on real hardware) and report the communicator's size; a launcher that started a different
number of ranks than --gpus asks for is an error, not a silently smaller run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_bare_gpus_2_starts_two_ranks_and_reports_the_communicator_size():
    r = run_bench(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["verified_bit_exact_vs_cpu"] is True
    assert out["value"] is None                 # a dry run never carries a throughput


def test_single_rank_dry_run():
    r = run_bench(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_rank_count_that_does_not_match_gpus_is_refused():
    r = run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "--gpus 2" in r.stderr
