"""The launcher half of `bench.py --gpus N` on CPU (no GPU work: --dry-run-ranks): a plain invocation starts
N ranks itself, the driver's torch.distributed.run form uses the launcher's ranks, and a request for more
GPUs than the box shows fails instead of reporting a smaller run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(stdout: str):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_invocation_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["rccl_world"] == 2 and rec["config"]["parallelism"] == "radial slabs x2"


def test_driver_form_uses_the_launchers_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--dry-run-ranks"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["rccl_world"] == 2


def test_more_gpus_than_visible_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    if n == 1:
        n = 2  # a plain 1-GPU run needs no ranks; ask for two on a GPU-less box
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_world_size_must_match_gpus():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_a_dying_rank_ends_the_run_instead_of_hanging_it():
    """rank 1 exits before the rendezvous; rank 0 would wait in it forever.  The parent polls all
    children, ends the survivors and reports the exit codes."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks", "--dry-run-fail-rank", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    assert r.stdout.strip() == ""
    assert "rank 1 exited with code 3" in r.stderr
    assert time.monotonic() - t0 < 120
