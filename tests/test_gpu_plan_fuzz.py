"""GPU: a short run of tools/plan_fuzz.py -- random shapes around the boundaries between the kernel families, one Krotov iteration
on the plan kr_create picks against the generic multi-pass kernels (independent code, themselves pinned to the oracle in
test_gpu_generic.py); pulses and states must agree to 1e-10, nothing may hang or fail to launch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env,seed", [pytest.param({}, 11, marks=pytest.mark.soak), pytest.param({"FUZZ_MAXWG": "64,100"}, 12, marks=pytest.mark.soak), ({"FUZZ_NMAX": "64"}, 13)])
def test_plans_agree_with_the_generic_kernels_on_random_shapes(env, seed):
    e = dict(os.environ)
    e.pop("KROTOV_HIP_GENERIC", None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "plan_fuzz.py"), "22", str(seed)], capture_output=True, text=True,
                       timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0 and "plan_fuzz: 0 problem(s)" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
