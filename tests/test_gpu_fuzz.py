"""A short randomised parity run on the GPU (tools/fuzz_parity.py with a fixed seed): random supported clip
lengths and batch sizes, random ragged indexes and queries through every scan kernel, against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity(torch_cuda):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "8", "11"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
