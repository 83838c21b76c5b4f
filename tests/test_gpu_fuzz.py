"""Randomised differential test of the device path against the oracle (tests/tools/fuzz_parity.py): random shapes around the
tile edges, the five kernels, the four nugget types, zero / analytic constant / analytic linear mean, 1-9 outputs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed, env", [(11, {}), (12, {}), (13, {"MOGP_MCHOL": "0"})], ids=["11", "12", "13-multi-launch-schedules"])
def test_random_configurations_match_the_oracle(seed, env):
    # (the default Cholesky of these sizes is the one-launch task-queue kernel; the third run keeps the multi-launch
    # schedules it falls back to under the same test)
    out = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"), "70", str(seed)],
                         capture_output=True, text=True, timeout=500, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "70 cases, 0 mismatches" in out.stdout
