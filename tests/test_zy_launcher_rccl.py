"""bench.py's N > 1 code path on the one GPU a test box has: one rank through the launcher (torch.distributed.run started by bench.py
itself) and RCCL, in every mode.  Collected late on purpose (it depends on process spawning and a rendezvous port): a failure here must
not hide the parity tests behind `pytest -x`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


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
