"""HipEngine: the device-resident replacement of what KrotovWrk holds per trajectory
(/root/reference/src/workspace.jl:128-161: fw/bw propagators and the three storages) plus the
two hot functions that drive them (/root/reference/src/optimize.jl:247-265, 279-371).

Array conventions on the Python side: operators (N, N) / (L, N, N) complex128 in NumPy C order,
states (K, N) complex128, pulses (L, NT) float64.  Everything is handed to libkrotov_hip.so
through the C ABI of include/krotov_hip.h.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


class HipEngine:
    def __init__(self, H0, Hc, tlist, K, *, hc_present=None, traj_generator=None, store_fw=True, prop_tol=0.0,
                 device=-1, rank=0, world=1, K_global=0, exchange=0, prop_method=0, max_workgroups=0, devices=None,
                 control_poly=None, spin_timeout_ms=0, options=None):
        """H0 (N,N) and Hc (L,N,N) for a generator shared by all trajectories, or H0 (G,N,N), Hc (G,L,N,N)
        with traj_generator[k] in 0..G-1 for per-trajectory generators (control_derivs[k][l],
        /root/reference/src/workspace.jl:74)."""
        lib = capi.load()
        self._lib = lib
        self._h = C.c_void_p()
        H0 = np.asarray(H0, np.complex128)
        Hc = np.asarray(Hc, np.complex128)
        if H0.ndim == 2:
            G = 1
            if Hc.ndim == 2:
                Hc = Hc[None]
        else:
            G = H0.shape[0]
            if Hc.ndim != 4 or Hc.shape[0] != G:
                raise ValueError("per-trajectory generators need H0 (G,N,N) and Hc (G,L,N,N)")
        tl = np.ascontiguousarray(tlist, np.float64)
        self.N = int(H0.shape[-1])
        self.L = int(Hc.shape[-3]) if Hc.size else 0
        self.NT = len(tl) - 1
        self.K = int(K)
        self.G = G
        self.rank, self.world = rank, world
        H0c, Hcc = capi.colmajor(H0), capi.colmajor(Hc)
        pres = None if hc_present is None else np.ascontiguousarray(hc_present, np.uint8)
        tg = None
        if G > 1:
            if traj_generator is None:
                raise ValueError("traj_generator is required when G > 1")
            tg = np.ascontiguousarray(traj_generator, np.int32)
            if tg.shape != (self.K,):
                raise ValueError("traj_generator must have length K")
        cfg = capi.KrConfig()
        cfg.abi_version = capi.KR_ABI_VERSION
        cfg.N, cfg.NT, cfg.K, cfg.L, cfg.G = self.N, self.NT, self.K, self.L, G
        cfg.traj_generator = capi.ptr(tg)
        cfg.tlist, cfg.H0, cfg.Hc = capi.ptr(tl), capi.ptr(H0c), capi.ptr(Hcc)
        cfg.hc_present = capi.ptr(pres)
        cfg.flags = capi.KR_STORE_FW if store_fw else capi.KR_NO_STORE_FW
        cfg.prop_tol = float(prop_tol)
        cfg.device = device
        cfg.rank, cfg.world, cfg.K_global = rank, world, K_global
        cfg.exchange, cfg.prop_method, cfg.max_workgroups = int(exchange), int(prop_method), int(max_workgroups)
        if devices is not None:
            if not 1 <= len(devices) <= 8:
                raise ValueError("devices: 1..8 HIP device ordinals")
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = int(d)
        cfg.spin_timeout_ms = int(spin_timeout_ms)   # bound of the in-kernel spins; 0 = default 4000 ms
        # kr_config.options: schedule / diagnostics options ("NAME=integer;..." or a dict); parsed inside kr_create
        if isinstance(options, dict):
            options = ";".join("%s=%d" % (k, int(v)) for k, v in options.items())
        self._options = options.encode() if options else None
        cfg.options = self._options
        self._cb = None
        self._poly = None
        if control_poly is not None:   # f_l(x) = sum_p control_poly[l][p] x^p  (non-linear controls, static part)
            self._poly = np.ascontiguousarray(control_poly, np.float64)
            if self._poly.ndim != 2 or self._poly.shape[0] != self.L or not 2 <= self._poly.shape[1] <= 8:
                raise ValueError("control_poly must have shape (L, degree + 1) with degree 1..7")
            cfg.control_poly_degree = self._poly.shape[1] - 1
            cfg.control_poly = capi.ptr(self._poly)
        rc = lib.kr_create(C.byref(cfg), C.byref(self._h))
        if rc != capi.KR_OK:
            msg = lib.kr_last_error(None)
            self._h = C.c_void_p()
            raise capi.KrotovHipError(rc, msg.decode() if msg else "?")

    # -- plumbing ---------------------------------------------------------------------------
    def _chk(self, rc):
        capi.check(self._lib, self._h, rc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.kr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _states(self, a, name):
        a = np.ascontiguousarray(a, np.complex128)
        if a.shape != (self.K, self.N):
            raise ValueError("%s must have shape (K=%d, N=%d), got %r" % (name, self.K, self.N, a.shape))
        return a

    def _pulses(self, a, name):
        a = np.ascontiguousarray(a, np.float64)
        if a.shape != (self.L, self.NT):
            raise ValueError("%s must have shape (L=%d, NT=%d), got %r" % (name, self.L, self.NT, a.shape))
        return a

    # -- ABI ----------------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._chk(self._lib.kr_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_initial_states(self, psi0):
        a = self._states(psi0, "psi0")
        self._chk(self._lib.kr_set_initial_states(self._h, capi.ptr(a)))

    def set_target_states(self, tgt):
        a = self._states(tgt, "targets")
        self._chk(self._lib.kr_set_target_states(self._h, capi.ptr(a)))

    def forward(self, eps):
        e = self._pulses(eps, "eps")
        out = np.empty((self.K, self.N), np.complex128)
        self._chk(self._lib.kr_forward(self._h, capi.ptr(e), capi.ptr(out)))
        return out

    def backward(self, eps_guess, chiT):
        e = self._pulses(eps_guess, "eps_guess")
        c = self._states(chiT, "chiT")
        self._chk(self._lib.kr_backward(self._h, capi.ptr(e), capi.ptr(c)))

    def update_forward(self, eps_guess, S, lam):
        e = self._pulses(eps_guess, "eps_guess")
        Sc = self._pulses(S, "S")
        lc = np.ascontiguousarray(lam, np.float64)
        eps_new = np.empty((self.L, self.NT))
        ga = np.empty(self.L)
        psiT = np.empty((self.K, self.N), np.complex128)
        self._chk(self._lib.kr_update_forward(self._h, capi.ptr(e), capi.ptr(Sc), capi.ptr(lc), capi.ptr(eps_new),
                                              capi.ptr(ga), capi.ptr(psiT)))
        return eps_new, ga, psiT

    def iteration(self, eps_guess, chiT, S, lam):
        e = self._pulses(eps_guess, "eps_guess")
        c = self._states(chiT, "chiT")
        Sc = self._pulses(S, "S")
        lc = np.ascontiguousarray(lam, np.float64)
        eps_new = np.empty((self.L, self.NT))
        ga = np.empty(self.L)
        psiT = np.empty((self.K, self.N), np.complex128)
        self._chk(self._lib.kr_iteration(self._h, capi.ptr(e), capi.ptr(c), capi.ptr(Sc), capi.ptr(lc),
                                         capi.ptr(eps_new), capi.ptr(ga), capi.ptr(psiT)))
        return eps_new, ga, psiT

    def read_storage(self, which, k, n):
        out = np.empty(self.N, np.complex128)
        w = capi.KR_STORAGE_BW if which in ("bw", capi.KR_STORAGE_BW) else capi.KR_STORAGE_FW
        self._chk(self._lib.kr_read_storage(self._h, w, int(k), int(n), capi.ptr(out)))
        return out

    def storage(self, which):
        """Whole storage as (NT+1, K, N) -- test helper built on kr_read_storage."""
        out = np.empty((self.NT + 1, self.K, self.N), np.complex128)
        for n in range(self.NT + 1):
            for k in range(self.K):
                out[n, k] = self.read_storage(which, k, n)
        return out

    # device-resident chain
    def set_update_params(self, S, lam):
        Sc = self._pulses(S, "S")
        lc = np.ascontiguousarray(lam, np.float64)
        self._chk(self._lib.kr_set_update_params(self._h, capi.ptr(Sc), capi.ptr(lc)))

    def forward_dev(self, eps_dev_ptr):
        self._chk(self._lib.kr_forward_dev(self._h, C.c_void_p(eps_dev_ptr)))

    def iteration_functional_dev(self, functional, eps_guess_ptr, eps_new_ptr):
        self._chk(self._lib.kr_iteration_functional_dev(self._h, int(functional), C.c_void_p(eps_guess_ptr),
                                                        C.c_void_p(eps_new_ptr)))

    def synchronize(self):
        self._chk(self._lib.kr_synchronize(self._h))

    def final_states(self):
        out = np.empty((self.K, self.N), np.complex128)
        self._chk(self._lib.kr_final_states(self._h, capi.ptr(out)))
        return out

    def taus(self):
        out = np.empty(self.K, np.complex128)
        self._chk(self._lib.kr_taus(self._h, capi.ptr(out)))
        return out

    def g_a_int(self):
        out = np.empty(self.L)
        self._chk(self._lib.kr_g_a_int(self._h, capi.ptr(out)))
        return out

    def enable_timing(self, on=True):
        self._chk(self._lib.kr_enable_timing(self._h, 1 if on else 0))

    def timings_ms(self):
        """{kind: (sum of kernel durations in ms, launches)} since enable_timing(True)."""
        t = (C.c_double * 6)()
        self._chk(self._lib.kr_get_timings(self._h, t))
        return {"forward": (t[0], int(t[1])), "backward": (t[2], int(t[3])), "update": (t[4], int(t[5]))}

    def debug_stamps(self):
        a = (C.c_uint64 * 8)()
        self._chk(self._lib.kr_debug_stamps(self._h, a))
        return [int(x) for x in a]

    def debug_arrivals(self, workgroups):
        """Diagnostic builds (-DKR_ARRIVALS): array [workgroups][2 exchanges][publish, window end, completion, passes]."""
        import numpy as np
        a = (C.c_uint64 * (8 * workgroups))()
        self._chk(self._lib.kr_debug_arrivals(self._h, a, 8 * workgroups))
        return np.array(a, dtype=np.uint64).reshape(workgroups, 2, 4)

    def norm_bounds(self):
        """kr_get_norm_bounds: (G, 1 + L) rigorous upper bounds of ||H0||_2, ||H_l||_2 (they set the degree of every step polynomial)."""
        out = np.empty((self.G, 1 + self.L))
        self._chk(self._lib.kr_get_norm_bounds(self._h, capi.ptr(out)))
        return out

    def plan(self):
        """kr_plan_description: one line naming the kernels, grid and exchange of this handle (and why, if not the persistent ones)."""
        msg = self._lib.kr_plan_description(self._h)
        return msg.decode() if msg else ""

    def info(self):
        a = (C.c_int32 * 16)()
        self._chk(self._lib.kr_get_info(self._h, a))
        keys = ["NP", "workgroups", "traj_per_wg", "threads", "KT", "LMAX", "K_padded", "store_fw", "JQ", "precomputed_propagators", "rowsplit_R", "exchange_poll_delay", "hermitian_controls_path", "matrix_core_sweep", "generic_known", "generic_update_exchange"]
        d = {k: int(a[i]) for i, k in enumerate(keys)}
        ge = d.pop("generic_update_exchange")
        d["generic_update"] = ge & 1
        d["exchange"] = ("auto", "mailbox", "rccl", "callback")[(ge >> 4) & 3]
        d["economized_polynomial"] = (ge >> 8) & 1
        return d

    # per-step exchange outside the kernels (KR_EXCHANGE_RCCL / KR_EXCHANGE_CALLBACK)
    @staticmethod
    def _torch_first():
        """One RCCL per process: PyTorch-ROCm bundles a librccl.so that libkrotov_hip would otherwise map BEFORE torch
        does; a process that imported torch only afterwards aborted at exit (free(): invalid pointer, measured).  So
        where torch is installed it is imported first and libkrotov_hip then finds its RCCL already mapped."""
        try:
            import torch  # noqa: F401
        except ImportError:
            pass

    @staticmethod
    def rccl_unique_id():
        HipEngine._torch_first()
        lib = capi.load()
        buf = (C.c_ubyte * lib.kr_rccl_unique_id_size())()
        capi.check(lib, None, lib.kr_rccl_get_unique_id(buf))
        return bytes(buf)

    def rccl_init(self, unique_id):
        HipEngine._torch_first()
        self._chk(self._lib.kr_rccl_init(self._h, C.c_char_p(bytes(unique_id))))

    def set_exchange_callback(self, fn):
        """fn(numpy array of L doubles) must sum the array over all ranks IN PLACE (e.g. a gloo / MPI all-reduce)."""
        def _tramp(ctx, buf, count):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # noqa: BLE001 -- never raise across the C boundary
                return 1
        self._cb = capi.ALLREDUCE_FN(_tramp)   # keep alive for the life of the handle
        self._chk(self._lib.kr_set_exchange_callback(self._h, self._cb, None))

    # multi-GPU wiring
    def ipc_export(self):
        n = self._lib.kr_ipc_handle_size()
        buf = (C.c_ubyte * n)()
        self._chk(self._lib.kr_ipc_export(self._h, buf))
        return bytes(buf)

    def ipc_connect(self, all_handles):
        blob = b"".join(all_handles)
        self._chk(self._lib.kr_ipc_connect(self._h, C.c_char_p(blob)))
