"""bench.py --gpus N starts N ranks by itself (no torch.distributed.run around it) before anything touches a GPU; a
mismatch between --gpus and WORLD_SIZE is an error.  CPU only: --dry-launch stops after the gloo rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_plain_invocation_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"dry_launch": True, "rank": 0, "local_rank": 0, "world": 2, "dist_world": 2}]


def test_a_rank_that_dies_early_ends_the_launch():
    """Rank 1 exits before the rendezvous: the launcher must notice, stop rank 0 (which would otherwise wait in
    init_process_group) and return the failing rank's code, well inside the collective's own timeout."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, env=dict(_env(), PPF_BENCH_DRY_FAIL_RANK="1"), timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-1000:])
    assert "rank 1 exited with 7" in r.stderr
    assert time.time() - t0 < 100
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def _free_port() -> str:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_under_torch_distributed_run_each_process_is_a_rank():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    objs = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", r.stdout)]  # the two ranks share one pipe: lines may interleave
    assert sorted((o["rank"], o["world"]) for o in objs) == [(0, 2), (1, 2)], r.stdout + r.stderr[-500:]
