import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# The GPU tests put several shards (device list {0, 0}) or several ranks on the ONE GPU of the box.  Their persistent kernels wait for
# each other, so every shard's stream needs a hardware queue of its own, and HIP multiplexes a process's streams over
# GPU_MAX_HW_QUEUES (default 4) queues: a cached RCCL communicator (round 5: kept for the life of the process) or PyTorch's own
# streams take some of them, and two shards that end up on one queue time out in their first exchange (seen: the RCCL test in front of
# the device-list tests).  A test facility -- with one shard per GPU the question does not arise.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_hip_lib: needs libkrotov_hip.so (hipcc to build it), but no GPU")
    config.addinivalue_line("markers", "soak: long fuzz / soak / micro-benchmark legs of the GPU suite; skipped unless KROTOV_RUN_SOAK=1 or '-m soak' "
                                       "(the builder's own gpurun sessions; every BASELINE config and kernel family stays in the default selection)")


def _have_hipcc():
    return os.path.exists(HIPCC) or shutil.which("hipcc") is not None


def pytest_collection_modifyitems(config, items):
    """CPU-only tests of the oracle and of the host logic must run on a machine without hipcc; only the tests that
    load libkrotov_hip.so (gpu / needs_hip_lib) require the HIP build."""
    want_soak = os.environ.get("KROTOV_RUN_SOAK", "") not in ("", "0") or "soak" in (config.getoption("-m") or "")
    if not want_soak:
        skip_soak = pytest.mark.skip(reason="soak leg: set KROTOV_RUN_SOAK=1 or select with -m soak")
        for it in items:
            if "soak" in it.keywords:
                it.add_marker(skip_soak)
    lib = os.path.join(ROOT, "krotov.jl_amd", "lib", "libkrotov_hip.so")
    if _have_hipcc() or os.path.exists(lib):
        return
    skip = pytest.mark.skip(reason="hipcc not found and libkrotov_hip.so not built")
    for it in items:
        if "gpu" in it.keywords or "needs_hip_lib" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_builds(request):
    """Build the C oracle always; build libkrotov_hip.so (hipcc, gfx950; `make` is incremental, a no-op when the tree
    was built by __graft_entry__.build()) only when a collected test needs it and hipcc exists."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    needs = any(("gpu" in it.keywords or "needs_hip_lib" in it.keywords) for it in request.session.items)
    if needs and _have_hipcc():
        import __graft_entry__ as ge
        ge.build()


@pytest.fixture(scope="session")
def oracle_c():
    import oracle_c as oc
    oc.build()
    return oc
