"""ctypes view of the C ABI (include/spock_amd.h).  The product library is libspock_amd.so next to this
file (built by ``__graft_entry__.build()`` / ``build.py`` with hipcc for gfx950).  There is no fallback:
if it is missing or cannot be loaded, importing a model fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libspock_amd.so")
F32_LIB = os.path.join(_HERE, "libspock_amd_f32.so")      # the same sources built with real = float (fp32 tolerance study)

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_BUILD = 0, 1, 2, 3, 4
CONVERGED, MAX_ITER, NAN_STATUS = 0, 1, 2


class SpockError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"spock_amd error {code}: {msg}")
        self.code = code


class Problem(C.Structure):
    _fields_ = [("N", C.c_int32), ("d", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("ny", C.c_int32),
                ("anc", C.c_void_p), ("w", C.c_void_p), ("A", C.c_void_p), ("B", C.c_void_p),
                ("nQ", C.c_int32), ("nR", C.c_int32), ("nQN", C.c_int32),
                ("Q", C.c_void_p), ("R", C.c_void_p), ("QN", C.c_void_p),
                ("E", C.c_void_p), ("F", C.c_void_p), ("b", C.c_void_p),
                ("n_cones", C.c_int32), ("cone_kind", C.c_void_p), ("cone_dim", C.c_void_p),
                ("x_min", C.c_double), ("x_max", C.c_double), ("u_min", C.c_double), ("u_max", C.c_double), ("n_rms", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("algorithm", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("split_stage", C.c_int32),
                ("L_norm", C.c_double), ("max_iter", C.c_int32), ("graph_iters", C.c_int32), ("batch_layout", C.c_int32),
                ("sp_variant", C.c_int32), ("sp_max_backtrack", C.c_int32), ("per_instance_dynamics", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("iter_ref", C.c_int32), ("reserved", C.c_int32),
                ("xi1", C.c_double), ("xi2", C.c_double), ("xi", C.c_double), ("solve_ms", C.c_double)]


class Sizes(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("n", "n_nonleaf", "n_leaf", "nz", "nv", "batch", "z_x", "z_u", "z_s", "z_tau", "z_y",
                 "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v11", "v12", "v13", "v14",
                 "split_stage", "n_subtrees", "graph_iters", "fused_iteration", "interleaved", "layout_reason")]


# every symbol include/spock_amd.h declares: (restype, argtypes)
_P, _D, _I = C.c_void_p, C.c_double, C.c_int32
SYMBOLS = {
    "spock_create": (C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "spock_destroy": (C.c_int, [_P]),
    "spock_sizes": (C.c_int, [_P, C.POINTER(Sizes)]),
    "spock_solve": (C.c_int, [_P, _P, _D, _I, _P, _P, _D, _D, _P]),
    "spock_mpc_simulate": (C.c_int, [_P, _P, _I, _P, _D, _I, _P, _P, _P]),
    "spock_get_z": (C.c_int, [_P, _P]),
    "spock_get_v": (C.c_int, [_P, _P]),
    "spock_set_state": (C.c_int, [_P, _P, _P]),
    "spock_reset": (C.c_int, [_P]),
    "spock_get_history": (C.c_int, [_P, _I, _P, _P, _P, _P, _I, C.POINTER(_I)]),
    "spock_op_L": (C.c_int, [_P, _P, _P]),
    "spock_op_Lt": (C.c_int, [_P, _P, _P]),
    "spock_op_prox_f": (C.c_int, [_P, _P, _D, _P]),
    "spock_op_prox_h_conj": (C.c_int, [_P, _D, _P]),
    "spock_op_dot": (C.c_int, [_P, _P, _P, _P, _P, _D, _D, C.POINTER(_D)]),
    "spock_op_T": (C.c_int, [_P, _P, _P, _P, _D, _D, _P, _P]),
    "spock_get_build_product": (C.c_int, [_P, _I, _P, _I, C.POINTER(_I)]),
    "spock_L_norm": (C.c_int, [_P, C.POINTER(_D), C.POINTER(_D)]),
    "spock_get_timing": (C.c_int, [_P, C.POINTER(_D), C.POINTER(_I)]),
    "spock_profile_kernels": (C.c_int, [_P, _I, _P, _P]),
    "spock_update_dynamics": (C.c_int, [_P, _P, _P]),
    "spock_update_dynamics_batch": (C.c_int, [_P, _P, _P, _P]),
    "spock_profile_sequence": (C.c_int, [_P, _I, _P, _P, _P, _I, _P]),
    "spock_debug_stamps": (C.c_int, [_P, _P]),
    "spock_last_error": (C.c_char_p, []),
    "spock_version": (C.c_int, []),
    "spock_real_size": (C.c_int, []),
    "spock_abi_struct_sizes": (C.c_int, [C.POINTER(_I), _I]),
}

_cache = {}


def load_library(path: str | None = None) -> C.CDLL:
    """Load the C-ABI library and bind every declared symbol.  ``path`` defaults to the in-tree
    libspock_amd.so; a missing library or symbol raises (no CPU fallback exists)."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the mirrors above against the library's own structs: a stale mirror must fail here, not inside spock_create
    sz = (_I * 4)()
    lib.spock_abi_struct_sizes(sz, 4)
    mine = [C.sizeof(Problem), C.sizeof(Options), C.sizeof(Result), C.sizeof(Sizes)]
    if list(sz) != mine:
        raise ImportError(f"{path}: ABI struct sizes {list(sz)} differ from this binding's {mine} (rebuild the library)")
    _cache[path] = lib
    return lib


def check(lib, code):
    if code != OK:
        raise SpockError(code, lib.spock_last_error().decode("utf-8", "replace"))
