"""bench.py's launcher rules (no GPU needed): a GPU count it cannot honour is an error, never a silent one-GPU run.
(One rank through the launcher and RCCL on a GPU box: tests/test_zy_launcher_rccl.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=120)


def test_more_gpus_than_visible_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    r = run(["--gpus", str(max(n, 2))])
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_world_size_must_equal_gpus():
    r = run(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode == 2 and "WORLD_SIZE is 2" in r.stderr and r.stdout.strip() == ""

