"""bench.py --force-collectives on the one-GPU box: the process group of backend nccl (= RCCL), the
count exchange, the bounds, sla_hip_shard_analyze and the overlapped all-gather of the residual planes run exactly as they do
for N > 1 -- on a one-rank communicator -- so that the first real multi-GPU run is not the first execution of that code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra", [[], ["--sync-gather"]], ids=["overlapped-gather", "synchronous-gather"])
def test_bench_runs_the_rccl_path_at_world_one(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C2", "--seconds", "40", "--steps", "3", "--warmup", "2",
           "--force-collectives", "--no-cpu-baseline", "--no-e2e", "--no-other-configs"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["nccl_ranks_seen"] == 1 and "collectives_forced" in line
    assert line["verified"] is True and line["verification"]["round_trip_identical"] is True
    assert "all-gather" in line["config"]["parallelism"] or line["config"]["parallelism"] == "1 GPU"
