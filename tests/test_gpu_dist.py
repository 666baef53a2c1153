"""GPU test of the collectives the multi-GPU path uses, on the RCCL backend (world size 1: the one-GPU
box's limit; world size 2 runs on gloo in test_dist_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_collectives_world1(torch_cuda):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1.py")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
