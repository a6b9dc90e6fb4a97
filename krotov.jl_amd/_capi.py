"""ctypes binding of libkrotov_hip.so (include/krotov_hip.h).

The library is the product; this file only marshals NumPy arrays across the C ABI.  There is
no CPU fallback: if the shared library is missing or no HIP device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KROTOV_HIP_LIB", os.path.join(_HERE, "lib", "libkrotov_hip.so"))  # env: A/B builds

KR_ABI_VERSION = 3
KR_OK, KR_ERR_INVALID, KR_ERR_NO_DEVICE, KR_ERR_HIP, KR_ERR_UNSUPPORTED, KR_ERR_TIMEOUT, KR_ERR_STATE, KR_ERR_INTERNAL = range(8)
KR_STORE_FW, KR_NO_STORE_FW = 1, 2
KR_J_T_SM, KR_J_T_SS, KR_J_T_RE = 0, 1, 2
KR_STORAGE_FW, KR_STORAGE_BW = 0, 1

# every symbol include/krotov_hip.h declares (tests check the .so exports exactly these)
EXPORTS = [
    "kr_create", "kr_destroy", "kr_last_error", "kr_abi_version", "kr_set_stream",
    "kr_set_initial_states", "kr_set_target_states", "kr_forward", "kr_backward",
    "kr_update_forward", "kr_iteration", "kr_set_update_params", "kr_forward_dev",
    "kr_iteration_functional_dev", "kr_final_states", "kr_taus", "kr_g_a_int", "kr_synchronize",
    "kr_read_storage", "kr_get_info", "kr_plan_description", "kr_get_timings", "kr_enable_timing", "kr_debug_stamps", "kr_debug_arrivals", "kr_get_norm_bounds",
    "kr_ipc_handle_size", "kr_ipc_export", "kr_ipc_connect",
    "kr_rccl_unique_id_size", "kr_rccl_get_unique_id", "kr_rccl_init", "kr_set_exchange_callback",
]
KR_EXCHANGE_AUTO, KR_EXCHANGE_MAILBOX, KR_EXCHANGE_RCCL, KR_EXCHANGE_CALLBACK = range(4)
KR_PROP_AUTO, KR_PROP_TAYLOR, KR_PROP_CHEBY, KR_PROP_NEWTON = range(4)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class KrConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("N", C.c_int32), ("NT", C.c_int32), ("K", C.c_int32),
        ("L", C.c_int32), ("G", C.c_int32),
        ("traj_generator", C.c_void_p), ("tlist", C.c_void_p), ("H0", C.c_void_p), ("Hc", C.c_void_p),
        ("hc_present", C.c_void_p),
        ("flags", C.c_uint32), ("prop_tol", C.c_double), ("device", C.c_int32),
        ("rank", C.c_int32), ("world", C.c_int32), ("K_global", C.c_int32),
        ("exchange", C.c_int32), ("prop_method", C.c_int32), ("max_workgroups", C.c_int32),
        ("n_devices", C.c_int32), ("devices", C.c_int32 * 8),
        ("control_poly_degree", C.c_int32), ("pad_", C.c_int32), ("control_poly", C.c_void_p),
        ("spin_timeout_ms", C.c_int32), ("reserved0", C.c_int32), ("options", C.c_char_p), ("reserved", C.c_int32 * 4),
    ]


class KrotovHipError(RuntimeError):
    """Raised for any non-zero status from libkrotov_hip (mirrors the ErrorException the Julia
    shim raises; /root/reference/src/optimize.jl:218-226 captures it into result.message)."""

    def __init__(self, code, message):
        super().__init__("libkrotov_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def _share_hip_runtime_with_torch():
    """One process, one HIP/HSA runtime.  PyTorch-ROCm bundles its own libamdhip64.so; libkrotov_hip.so names the
    same soname.  If ours (from /opt/rocm) is loaded first and torch's copy later, the second runtime finds no GPU
    ("No HIP GPUs are available").  So when torch is installed, its runtime is mapped first (without importing
    torch) and libkrotov_hip.so binds to it; a later `import torch` then reuses the same mapping."""
    if os.environ.get("KROTOV_HIP_SHARE_TORCH_RUNTIME", "1") == "0":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        cand = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        # the same for RCCL, but only by NAME: libkrotov_hip dlopens it when (and only when) exchange = RCCL is used
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("KROTOV_HIP_RCCL_LIB", rccl)
    except Exception:  # noqa: BLE001 -- fall back to the system runtime
        pass


def load():
    """dlopen the HIP engine; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KrotovHipError(KR_ERR_NO_DEVICE,
                             "%s not found -- build it with __graft_entry__.build() "
                             "(make -C krotov.jl_amd/csrc); there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    lib.kr_last_error.restype = C.c_char_p
    lib.kr_last_error.argtypes = [C.c_void_p]
    lib.kr_plan_description.restype = C.c_char_p
    lib.kr_plan_description.argtypes = [C.c_void_p]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("kr_last_error", "kr_plan_description"):
            fn.restype = C.c_int
    lib.kr_create.argtypes = [C.POINTER(KrConfig), C.POINTER(C.c_void_p)]
    _lib = lib
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def colmajor(H):
    """NumPy C-order (..., N, N) operator -> the ABI's column-major memory."""
    H = np.asarray(H, np.complex128)
    return np.ascontiguousarray(np.swapaxes(H, -1, -2))


def check(lib, h, rc):
    if rc != KR_OK:
        msg = lib.kr_last_error(h)
        raise KrotovHipError(rc, msg.decode() if msg else "?")
