"""The C99 consumer of include/krotov_hip.h (tests/abi_consumer.c): the runnable stand-in for the Julia `ccall` binding, which
cannot execute here (no Julia).  Built with gcc -std=c99 -pedantic against the header and linked against libkrotov_hip.so --
no Python, ctypes or NumPy between the caller and the ABI.

  CPU (-m "not gpu"): it compiles warning-free, links, passes the error half of the contract (NULL config, wrong ABI version,
                      L = 0 with the reference's message) and reports KR_ERR_NO_DEVICE (exit 3) instead of computing anything;
  GPU (-m gpu):       kr_create -> kr_set_initial_states -> kr_forward -> kr_iteration -> kr_read_storage -> kr_destroy on the
                      golden fixtures, compared inside the C program at the tolerances of test_gpu_parity.py.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "krotov.jl_amd", "lib")


def _build(tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "abi_consumer")
    out = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "abi_consumer.c"), "-o", exe, "-L", LIBDIR, "-lkrotov_hip", "-lm",
                          "-Wl,-rpath," + LIBDIR], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.needs_hip_lib
@pytest.mark.skipif(_gpu_present(), reason="a GPU is present: the gpu-marked test covers the full sequence")
def test_c99_consumer_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(LIBDIR, "libkrotov_hip.so")):
        ge.build()
    exe = _build(tmp_path)
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "hermitian_n16.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stdout, out.stderr)      # error half passed, then KR_ERR_NO_DEVICE
    assert "no CPU fallback" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["hermitian_n16", "nonhermitian_n8"])
def test_c99_consumer_runs_the_hot_path_against_the_golden_fixture(tmp_path, fixture):
    exe = _build(tmp_path)
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", fixture + ".bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi_consumer: ok" in out.stdout
    print(out.stdout.strip())
