"""CPU: the C-ABI library loads, exports every symbol include/krotov_hip.h declares, and fails loudly
without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.needs_hip_lib   # loads libkrotov_hip.so (no GPU needed); skipped where hipcc is absent


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "krotov_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kr_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(capi.LIB_PATH):
        ge.build()
    return capi.load()


def test_header_and_binding_agree(lib):
    declared = _declared_symbols()
    assert declared == sorted(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version(lib):
    assert lib.kr_abi_version() == capi.KR_ABI_VERSION == 3


def test_the_ipc_blob_is_sized_by_the_library(lib):
    """kr_ipc_export hands out kr_ipc_handle_size() opaque bytes: the HIP IPC handle (64) plus, since round 5, the generation of the
    exporter's pooled mailbox and a flag (16) -- every rank derives the tag base of a set of handles from all ranks' generations, so
    that a late store of an earlier set cannot satisfy a poll of the set that recycled the block.  Callers must size their buffers
    with the function (the Python binding does), never with a constant."""
    assert lib.kr_ipc_handle_size() == 80
    src = open(os.path.join(ROOT, "krotov.jl_amd", "engine.py")).read()
    assert "kr_ipc_handle_size()" in src


def test_a_v2_config_is_refused(lib):
    """ABI v3 gave two of v2's `reserved` words a meaning (spin_timeout_ms, the `options` pointer that kr_create dereferences): a v2
    caller that never zeroed them must be stopped by the version check, before anything reads those words."""
    cfg = capi.KrConfig()
    cfg.abi_version = 2
    cfg.N, cfg.NT, cfg.K, cfg.L, cfg.G = 2, 4, 1, 1, 1
    cfg.spin_timeout_ms = -12345              # what an unzeroed v2 `reserved` might hold
    cfg.options = C.cast(C.c_void_p(0x10), C.c_char_p)   # a wild pointer: must never be dereferenced
    h = C.c_void_p()
    assert lib.kr_create(C.byref(cfg), C.byref(h)) == capi.KR_ERR_INVALID and not h
    assert b"abi_version 2 != 3" in lib.kr_last_error(None)


def test_config_struct_layout():
    # mirrors `struct kr_config` (include/krotov_hip.h, ABI v3): 6 ints, 5 pointers, flags (+ pad), double, 4 ints,
    # 4 + 8 ints of v2, control_poly (2 ints + pointer), spin bound + pad, options pointer, 4 reserved
    assert C.sizeof(capi.KrConfig) == 24 + 40 + 4 + 4 + 8 + 16 + 16 + 32 + 16 + 8 + 8 + 16
    assert capi.KrConfig.options.offset == 168 and capi.KrConfig.spin_timeout_ms.offset == 160


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_no_controls_is_an_error(lib):
    """L = 0 -> the reference's message (workspace.jl:72, test_empty_optimization.jl:31-32)."""
    with pytest.raises(kr.KrotovHipError) as ei:
        kr.HipEngine(np.eye(2), np.zeros((0, 2, 2)), np.linspace(0, 1, 5), 1)
    assert "no controls in trajectories: cannot optimize" in str(ei.value)
    assert ei.value.code == capi.KR_ERR_INVALID


def test_bad_arguments_are_rejected(lib):
    with pytest.raises(kr.KrotovHipError):
        kr.HipEngine(np.eye(2), np.eye(2)[None], np.array([0.0, 0.1, 0.1, 0.2]), 1)  # tlist not increasing


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present")
def test_fails_loudly_without_gpu(lib):
    with pytest.raises(kr.KrotovHipError) as ei:
        kr.HipEngine(np.eye(2), np.eye(2)[None], np.linspace(0, 1, 5), 1)
    assert ei.value.code == capi.KR_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "krotov.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_c" not in src and "krotov_oracle" not in src, f


def test_every_option_the_library_accepts_is_documented():
    """kr_create rejects option names it does not know (krotov_hip.hip, kr_option_names); INTEGRATION.md is where an integrator looks them
    up: every accepted name must appear there as KROTOV_HIP_<NAME> (the environment spelling)."""
    import re
    src = open(os.path.join(ROOT, "krotov.jl_amd", "csrc", "krotov_hip.hip")).read()
    body = re.search(r"kr_option_names\[\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    names = re.findall(r'"([A-Z0-9_]+)"', body)
    assert len(names) > 30
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in names if ("KROTOV_HIP_" + n) not in doc]
    assert not missing, missing
