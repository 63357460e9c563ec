"""bench.py's launcher rules (no GPU needed): a GPU count it cannot honour is an error, never a silent one-GPU run; and, on a GPU
box, one rank through the launcher and RCCL in every mode."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


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


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,extra", [("tiles", []), ("spp", []), ("spp", ["--reduce", "allreduce"])])
def test_one_rank_under_the_launcher_over_rccl(mode, extra):
    """The N > 1 code path with the one GPU a test box has: bench.py starts torch.distributed.run itself, the rank initialises RCCL,
    renders its strips / sample range, goes through the gather / reduce and prints ONE JSON line with rccl_ranks = 1."""
    cmd = [sys.executable, BENCH, "--gpus", "1", "--launcher", "--mode", mode, "--workload", "weekend-400x225-16spp", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-target-scene"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["mode"] == mode and d["value"] > 0
    assert d["rays_per_step"]["primary"] == 400 * 225 * 16
