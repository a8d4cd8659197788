"""The plain-C host of examples/shard_rccl.c (libsla_hip.so + RCCL, no Python in the data path): one file sharded over
the ranks with ncclAllGather as the only exchange, byte-identical to the single-GPU file.  World size 1 on the test box
(the collectives still run through RCCL); the two-rank launch shares the single GPU."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sla_amd", "shard_rccl")


def _build():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "sla_amd", "csrc"), "examples"], check=True)


@pytest.mark.timeout(300)
def test_shard_rccl_world_1():
    _build()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([EXE, "12"], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ranks"] == 1 and line["identical_to_single_gpu"] and line["round_trip"]
