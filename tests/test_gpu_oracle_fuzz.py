"""GPU: a short run of tools/oracle_fuzz.py -- random problems (state dimension, ensemble size, controls, grid, dissipation, two
generators, absent control derivatives, functional, update shapes and lambda_a) on the HIP engine DIRECTLY against the NumPy oracle:
forward sweep and two chained iterations, states to 1e-11, pulses to 1e-10, the update itself to 1e-8 of its size."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [31, 32])
def test_random_problems_against_the_oracle(seed):
    e = dict(os.environ)
    e["ORACLE_FUZZ_NMAX"] = "130"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_fuzz.py"), "30", str(seed)], capture_output=True, text=True,
                       timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0 and "oracle_fuzz: 0 problem(s) in 30 shapes" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
