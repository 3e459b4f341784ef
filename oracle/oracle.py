"""ctypes binding of the CPU oracle (oracle/spock_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product path (spock.jl_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

CP, SP = 1, 2
SILENT, LOG = 1, 2


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int), ("d", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("ny", C.c_int),
                ("A", C.c_void_p), ("B", C.c_void_p),
                ("nQ", C.c_int), ("nR", C.c_int), ("nQN", C.c_int),
                ("Q", C.c_void_p), ("R", C.c_void_p), ("QN", C.c_void_p),
                ("E", C.c_void_p), ("F", C.c_void_p), ("b", C.c_void_p),
                ("n_cones", C.c_int), ("cone_kind", C.c_void_p), ("cone_dim", C.c_void_p),
                ("x_min", C.c_double), ("x_max", C.c_double), ("u_min", C.c_double), ("u_max", C.c_double), ("n_rms", C.c_int)]


class _Result(C.Structure):
    _fields_ = [("iterations", C.c_int), ("iter_ref", C.c_int), ("status", C.c_int),
                ("xi1", C.c_double), ("xi2", C.c_double), ("xi", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "spock_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(_Problem), C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_sizes.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_offsets.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("orc_z", "orc_v", "orc_P", "orc_K", "orc_Rchol", "orc_ABK", "orc_ls", "orc_sqrtQ", "orc_sqrtR", "orc_sqrtQN"):
            getattr(L, name).restype = C.POINTER(C.c_double)
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_get_L_norm.restype = C.c_double
        L.orc_get_L_norm.argtypes = [C.c_void_p]
        L.orc_set_L_norm.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_sp_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_L.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_Lt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double, C.c_double]
        L.orc_prox_f.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.orc_prox_h_conj.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        L.orc_set_x0.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_T.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double, C.c_double]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_double, C.c_double, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                C.POINTER(_Result)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """One reference-style model (CPOCK or SPOCK) on the CPU.  State persists between solves,
    exactly like the Julia model struct (warm start = leftover state)."""

    def __init__(self, flat, algorithm: int = CP, sp_variant: int = 0, sp_max_backtrack: int = 0):
        """sp_variant 0: run_sp! / anderson! exactly as the reference has them (bug-compatible); 1: the corrected variant
        (least-squares Anderson, working line search) that spock_options_t.sp_variant = 1 selects on the GPU."""
        self.flat = flat
        self._keep = flat
        pr = _Problem(flat.N, flat.d, flat.nx, flat.nu, flat.ny, _p(flat.A), _p(flat.B), flat.nQ, flat.nR, flat.nQN,
                      _p(flat.Q), _p(flat.R), _p(flat.QN), _p(flat.E), _p(flat.F), _p(flat.b),
                      len(flat.cone_kind), _p(flat.cone_kind), _p(flat.cone_dim),
                      flat.x_min, flat.x_max, flat.u_min, flat.u_max, getattr(flat, "n_rms", 1))
        self.h = lib().orc_create(C.byref(pr), algorithm)
        if not self.h:
            raise ValueError("oracle: invalid problem")
        sz = np.zeros(6, dtype=np.int32)
        lib().orc_sizes(self.h, _p(sz))
        self.n, self.nn, self.nl, self.nz, self.nv, self.ny = (int(x) for x in sz)
        assert self.nz == flat.nz and self.nv == flat.nv
        self.algorithm = algorithm
        if sp_variant:
            lib().orc_set_sp_variant(self.h, int(sp_variant), int(sp_max_backtrack))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_free(self.h)
            self.h = None

    # --- state views (live memory of the oracle)
    @property
    def z(self):
        return np.ctypeslib.as_array(lib().orc_z(self.h), shape=(self.nz,))

    @property
    def v(self):
        return np.ctypeslib.as_array(lib().orc_v(self.h), shape=(self.nv,))

    @property
    def L_norm(self):
        return lib().orc_get_L_norm(self.h)

    @L_norm.setter
    def L_norm(self, val):
        lib().orc_set_L_norm(self.h, float(val))

    def build_products(self):
        f = self.flat
        N, d, nx, nu, m = f.N, f.d, f.nx, f.nu, f.ny + 2 * f.d
        g = lambda fn, k: np.ctypeslib.as_array(getattr(lib(), fn)(self.h), shape=(k,)).copy()
        return {"P": g("orc_P", N * nx * nx), "K": g("orc_K", (N - 1) * nu * nx), "Rchol": g("orc_Rchol", (N - 1) * nu * nu),
                "ABK": g("orc_ABK", N * d * nx * nx), "ls": g("orc_ls", m * m),
                "sqrtQ": g("orc_sqrtQ", f.nQ * nx * nx), "sqrtR": g("orc_sqrtR", f.nR * nu * nu),
                "sqrtQN": g("orc_sqrtQN", f.nQN * nx * nx)}

    # --- operators (new arrays returned; inputs untouched)
    def L(self, z):
        z = np.ascontiguousarray(z, dtype=np.float64); v = np.empty(self.nv)
        lib().orc_L(self.h, _p(z), _p(v)); return v

    def Lt(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64); z = np.empty(self.nz)
        lib().orc_Lt(self.h, _p(z), _p(v)); return z

    def prox_f(self, arg, gamma, x0=None):
        out = np.array(arg, dtype=np.float64, copy=True)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64); lib().orc_set_x0(self.h, _p(x0))
        lib().orc_prox_f(self.h, _p(out), float(gamma)); return out

    def prox_h_conj(self, arg, sigma):
        out = np.array(arg, dtype=np.float64, copy=True)
        lib().orc_prox_h_conj(self.h, _p(out), float(sigma)); return out

    def dot(self, az, av, bz, bv, gamma, sigma):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (az, av, bz, bv)]
        return lib().orc_dot(self.h, *[_p(x) for x in a], float(gamma), float(sigma))

    def T(self, z, v, gamma, sigma, x0=None):
        z = np.ascontiguousarray(z, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64); lib().orc_set_x0(self.h, _p(x0))
        zo = np.empty(self.nz); vo = np.empty(self.nv)
        lib().orc_T(self.h, _p(z), _p(v), _p(zo), _p(vo), float(gamma), float(sigma)); return zo, vo

    def solve(self, x0, tol=1e-3, verbose=SILENT, z0=None, v0=None, gamma=None, sigma=None, max_iter=0, history=0):
        """Mirror of solve_model!(model, x0; tol, verbose, z0, v0, gamma, sigma).  Returns a dict with the
        iteration counts and (if ``history`` > 0) the per-iteration residual logs."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        z0 = None if z0 is None else np.ascontiguousarray(z0, dtype=np.float64)
        v0 = None if v0 is None else np.ascontiguousarray(v0, dtype=np.float64)
        g = float("nan") if gamma is None or sigma is None else float(gamma)
        s = float("nan") if gamma is None or sigma is None else float(sigma)
        cap = int(history)
        hx = np.full(cap, np.nan); h1 = np.full(cap, np.nan); h2 = np.full(cap, np.nan); hb = np.zeros(cap, dtype=np.int32)
        hl = C.c_int(0); res = _Result()
        rc = lib().orc_solve(self.h, _p(x0), float(tol), int(verbose), _p(z0), _p(v0), g, s, int(max_iter),
                             _p(hx) if cap else None, _p(h1) if cap else None, _p(h2) if cap else None,
                             _p(hb) if cap else None, cap, C.byref(hl), C.byref(res))
        if rc != 0:
            raise ValueError("z0 and v0 must be given together")
        k = hl.value
        return {"iterations": res.iterations, "iter_ref": res.iter_ref, "status": res.status,
                "xi1": res.xi1, "xi2": res.xi2, "xi": res.xi,
                "hist_xi": hx[:k], "hist_xi1": h1[:k], "hist_xi2": h2[:k], "hist_backtrack": hb[:k]}
