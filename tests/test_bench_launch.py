"""CPU: `python bench.py --gpus N` with no launcher starts its own N ranks (it used to run one rank silently).
The GPU step cannot run here, so the launcher, rendezvous, weight broadcast, live-only bucketed all-reduce and the
barrier / max-over-ranks protocol are rehearsed with `--rehearse-cpu` (gloo, toy torch model; the line says valid: false)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, gpus=2, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--rehearse-cpu", "--steps", "3",
                           "--warmup", "1", "--batch", "8", *extra_args], env=env, capture_output=True, text=True, timeout=180)


def test_gpus_2_without_launcher_starts_two_ranks():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["valid"] is False and out["steps"] == 3 and out["warmup"] == 1
    ar = out["allreduce"]
    assert ar["replicas_agree"] is True
    # toy model: 48x64 + 64x64 (never receives a gradient) + 64x3 floats; after step 1 only the live ranges travel
    assert ar["bytes_per_step_first"] == (48 * 64 + 64 * 64 + 64 * 3) * 4
    assert ar["bytes_per_step"] == (48 * 64 + 64 * 3) * 4


def test_a_rank_that_dies_fails_the_run_instead_of_hanging():
    t0 = time.time()
    r = _run({"VNP_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited with status 3" in r.stderr
    assert time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_a_rank_that_is_stuck_fails_the_run_within_the_deadline():
    """A rank that is alive but never joins the rendezvous: nobody exits, so only the launcher's wall-clock deadline (and the
    rendezvous timeout of the ranks that did join) ends the run — non-zero exit, every rank terminated, well inside any driver limit."""
    t0 = time.time()
    r = _run({"VNP_BENCH_STUCK_RANK": "1", "VNP_DIST_TIMEOUT_S": "120"}, extra_args=("--deadline", "10"))
    assert r.returncode != 0
    assert "deadline" in r.stderr
    assert time.time() - t0 < 90
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
