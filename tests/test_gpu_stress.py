"""GPU: a short randomised parity sweep (tools/stress_parity.py with a fixed seed range): small scenes with many short-lived
trees, odd chunkings, both sensors, both association kernels — engine vs oracle, bit-exact."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep(oracle_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "10", "9000"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "failures: 0" in r.stdout
