"""Host-side mirror of the reference's model API on top of the C ABI:

  build_model(scen_tree, cost, dynamics, rms, constraints, SolverOptions(CP | SP, nothing))   src/model.jl:273-297
  solve_model!(model, x0; tol=1e-3, verbose=SILENT, z0, v0, gamma, sigma)                     src/models/cpock.jl:169-189
  model.solver_state.z[model.solver_state_internal.{x,u,s,tau,y}_inds]                        examples/server_heat/mpc_simulation.jl:50-54

State persists inside the model between solves (warm start = leftover state), exactly like the Julia
struct.  ``batch`` > 1 (extension) holds several independent instances of the same problem data, each
with its own x0 and state; x0 then has shape [batch, nx] and solver_state.z shape [batch, nz].
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from enum import IntEnum
from typing import Optional

import numpy as np

from . import _lib
from .problem import FlatProblem, flatten_problem


class AlgorithmOptions(IntEnum):      # model.jl:61-64
    CP = 1
    SP = 2


class QNewtonOptions(IntEnum):        # model.jl:66-69 (ignored by the reference's builder: models/spock.jl:219-222)
    AA = 1
    RB = 2


class VERBOSE_LEVEL(IntEnum):         # model.jl:251-254
    SILENT = 1
    LOG = 2


CP, SP = AlgorithmOptions.CP, AlgorithmOptions.SP
SILENT, LOG = VERBOSE_LEVEL.SILENT, VERBOSE_LEVEL.LOG


@dataclass
class SolverOptions:                  # model.jl:71-74
    algorithm: AlgorithmOptions = AlgorithmOptions.CP
    qnewton: Optional[QNewtonOptions] = None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _SolverState:
    """model.solver_state: nz, nv, L_norm and the iterate z, v.  The iterate lives on the GPU; ``z`` / ``v`` copy it
    out on first access after a solve (shape [nz] / [nv], or [batch, .] for batch > 1)."""

    def __init__(self, model, nz, nv, L_norm):
        self._model = model
        self.nz, self.nv, self.L_norm = nz, nv, L_norm
        self._z = self._v = None

    def _invalidate(self):
        self._z = self._v = None

    @property
    def z(self):
        if self._z is None:
            self._z = self._model._fetch("z")
        return self._z

    @property
    def v(self):
        if self._v is None:
            self._v = self._model._fetch("v")
        return self._v


class _SolverStateInternal:
    """model.solver_state_internal: the index vectors and v offsets scripts use (0-based)."""

    def __init__(self, flat: FlatProblem):
        sl = flat.z_slices()
        self.x_inds = np.arange(sl["x"].start, sl["x"].stop)
        self.u_inds = np.arange(sl["u"].start, sl["u"].stop)
        self.s_inds = np.arange(sl["s"].start, sl["s"].stop)
        self.tau_inds = np.arange(sl["tau"].start, sl["tau"].stop)
        self.y_inds = np.arange(sl["y"].start, sl["y"].stop)
        off = flat.v_offsets()
        for k in ("v2", "v3", "v4", "v5", "v6", "v7", "v11", "v12", "v13", "v14"):
            setattr(self, k + "_offset", off[k])


class Model:
    """CPOCK / SPOCK equivalent (model.jl:127-131, :228-232) whose state lives on the GPU."""

    def __init__(self, flat: FlatProblem, options: SolverOptions, batch: int = 1, device: int = 0,
                 L_norm: float | str | None = None, max_iter: int = 0, split_stage: int = 0, graph_iters: int = 0,
                 lib_path: str | None = None, layout: str | int | None = None, dtype: str = "f64", sp_variant: int = 0,
                 sp_max_backtrack: int = 0, per_instance_dynamics: bool = False):
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        self._lib = _lib.load_library(lib_path or (_lib.F32_LIB if dtype == "f32" else None))
        self.dtype = "f32" if self._lib.spock_real_size() == 4 else "f64"
        self.problem_definition = flat
        self.options = options
        self.batch = int(batch)
        pr = _lib.Problem(flat.N, flat.d, flat.nx, flat.nu, flat.ny, _ptr(flat.anc), _ptr(flat.w), _ptr(flat.A), _ptr(flat.B),
                          flat.nQ, flat.nR, flat.nQN, _ptr(flat.Q), _ptr(flat.R), _ptr(flat.QN),
                          _ptr(flat.E), _ptr(flat.F), _ptr(flat.b), len(flat.cone_kind), _ptr(flat.cone_kind),
                          _ptr(flat.cone_dim), flat.x_min, flat.x_max, flat.u_min, flat.u_max, int(flat.n_rms))
        lay = {None: 0, "auto": 0, "instance": 1, "instance-major": 1, "interleaved": 2, "element-major": 2}.get(layout, layout)
        op = _lib.Options(int(options.algorithm), int(device), self.batch, int(split_stage),
                          float("nan") if L_norm is None else (-1.0 if L_norm == "auto" else float(L_norm)), int(max_iter), int(graph_iters),
                          int(lay), int(sp_variant), int(sp_max_backtrack), 1 if per_instance_dynamics else 0)
        h = C.c_void_p()
        _lib.check(self._lib, self._lib.spock_create(C.byref(pr), C.byref(op), C.byref(h)))
        self._h = h
        sz = _lib.Sizes()
        _lib.check(self._lib, self._lib.spock_sizes(self._h, C.byref(sz)))
        self.sizes = sz
        assert sz.nz == flat.nz and sz.nv == flat.nv
        used = C.c_double(0.); closed = C.c_double(0.)
        _lib.check(self._lib, self._lib.spock_L_norm(self._h, C.byref(used), C.byref(closed)))
        self.L_norm_closed_form = closed.value
        self.solver_state = _SolverState(self, sz.nz, sz.nv, used.value)
        self.solver_state_internal = _SolverStateInternal(flat)
        self.results = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spock_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state
    def _fetch(self, which):
        n = self.sizes.nz if which == "z" else self.sizes.nv
        out = np.empty((self.batch, n))
        fn = self._lib.spock_get_z if which == "z" else self._lib.spock_get_v
        _lib.check(self._lib, fn(self._h, _ptr(out)))
        return out[0] if self.batch == 1 else out

    def _refresh(self):
        self.solver_state._invalidate()

    def set_state(self, z, v):
        z = np.ascontiguousarray(np.broadcast_to(np.asarray(z, dtype=np.float64).reshape(-1, self.sizes.nz), (self.batch, self.sizes.nz)))
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1, self.sizes.nv), (self.batch, self.sizes.nv)))
        _lib.check(self._lib, self._lib.spock_set_state(self._h, _ptr(z), _ptr(v)))
        self._refresh()

    def reset(self):
        _lib.check(self._lib, self._lib.spock_reset(self._h))
        self._refresh()

    def history(self, inst: int = 0):
        n = C.c_int32(0)
        _lib.check(self._lib, self._lib.spock_get_history(self._h, inst, None, None, None, None, 0, C.byref(n)))
        k = n.value
        xi = np.empty(k); x1 = np.empty(k); x2 = np.empty(k); bt = np.zeros(k, dtype=np.int32)
        _lib.check(self._lib, self._lib.spock_get_history(self._h, inst, _ptr(xi), _ptr(x1), _ptr(x2), _ptr(bt), k, C.byref(n)))
        return {"xi": xi, "xi1": x1, "xi2": x2, "backtrack": bt}

    def build_product(self, which: int):
        n = C.c_int32(0)
        _lib.check(self._lib, self._lib.spock_get_build_product(self._h, which, None, 0, C.byref(n)))
        out = np.empty(n.value)
        _lib.check(self._lib, self._lib.spock_get_build_product(self._h, which, _ptr(out), n.value, C.byref(n)))
        return out

    def timing(self):
        ms = C.c_double(0.); it = C.c_int32(0)
        _lib.check(self._lib, self._lib.spock_get_timing(self._h, C.byref(ms), C.byref(it)))
        return ms.value, it.value

    def profile_kernels(self, reps: int = 50):
        """Average HIP-event duration (ms) per kernel kind over `reps` extra iterations:
        {"sub_a", "crown", "sub_b", "full"} -> (avg_ms, launches)."""
        ms = np.zeros(4); n = np.zeros(4, dtype=np.int32)
        _lib.check(self._lib, self._lib.spock_profile_kernels(self._h, int(reps), _ptr(ms), _ptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(("sub_a", "crown", "sub_b", "full"))}

    def update_dynamics(self, dynamics):
        """New dynamics of the same shape: ricatti_offline (implicit_l.jl:46-97) and the sweep matrices are recomputed on the GPU in
        the model's matrix pool (spock_update_dynamics); costs, risk measures, constraints and the solver state are kept."""
        from .problem import _stack_colmajor
        fl = self.problem_definition
        if len(dynamics.A) != fl.d or len(dynamics.B) != fl.d:
            raise ValueError("the new dynamics must have the shape of the old ones")
        A = _stack_colmajor(dynamics.A, fl.nx, fl.nx); B = _stack_colmajor(dynamics.B, fl.nx, fl.nu)
        _lib.check(self._lib, self._lib.spock_update_dynamics(self._h, _ptr(A), _ptr(B)))   # raises on failure: the model then keeps its dynamics
        fl.A, fl.B = A, B

    def update_dynamics_batch(self, dynamics_list):
        """One draw of the dynamics per instance (Monte-Carlo draws): instance b gets dynamics_list[b].A / .B; the device builds all of
        them side by side (spock_update_dynamics_batch).  The model must have been built with per_instance_dynamics=True.  Returns the
        per-instance status list (0 ok, 4 build failed: that instance keeps its previous dynamics; the call then raises, the statuses
        stay in ``last_build_status``)."""
        from .problem import _stack_colmajor
        fl = self.problem_definition
        if len(dynamics_list) != self.batch:
            raise ValueError("one Dynamics per instance expected")
        A = np.ascontiguousarray(np.concatenate([_stack_colmajor(dy.A, fl.nx, fl.nx) for dy in dynamics_list]))
        B = np.ascontiguousarray(np.concatenate([_stack_colmajor(dy.B, fl.nx, fl.nu) for dy in dynamics_list]))
        st = np.zeros(self.batch, dtype=np.int32)
        self.last_build_status = st
        _lib.check(self._lib, self._lib.spock_update_dynamics_batch(self._h, _ptr(A), _ptr(B), _ptr(st)))
        return st.tolist()

    MODE_NAMES = {0: "full", 1: "sub_a", 2: "crown", 3: "sub_b", 5: "sub_ba", 8: "sub_b_lat", 9: "crown_a", 10: "i_up", 11: "i_down",
                  12: "lb", 13: "lba", 14: "la", 15: "la0", 20: "lba2", 21: "sub_ba_r", 16: "flat_a", 17: "flat_b", 18: "flat_b_sp", 19: "flat_ba"}

    def profile_sequence(self, reps: int = 20):
        """Average HIP-event duration (ms) of every launch of one CP iteration, in launch order: [(kernel, level, avg_ms)]."""
        cap = 64
        modes = np.zeros(cap, dtype=np.int32); lv = np.zeros(cap, dtype=np.int32); ms = np.zeros(cap); n = C.c_int32(0)
        _lib.check(self._lib, self._lib.spock_profile_sequence(self._h, int(reps), _ptr(modes), _ptr(lv), _ptr(ms), cap, C.byref(n)))
        return [(self.MODE_NAMES.get(int(modes[k]), str(int(modes[k]))), int(lv[k]), float(ms[k])) for k in range(min(n.value, cap))]

    # ---- operator-level seam (single instance): L!, L_transpose!, prox_f!, prox_h_conj!, spock_dot
    def L(self, z):
        z = np.ascontiguousarray(z, dtype=np.float64); out = np.empty(self.sizes.nv)
        if z.shape != (self.sizes.nz,):
            raise ValueError("z or v has the wrong length")       # implicit_l.jl:184-186
        _lib.check(self._lib, self._lib.spock_op_L(self._h, _ptr(z), _ptr(out))); return out

    def L_transpose(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64); out = np.empty(self.sizes.nz)
        if v.shape != (self.sizes.nv,):
            raise ValueError("z or v has the wrong length")       # implicit_l.jl:327-329
        _lib.check(self._lib, self._lib.spock_op_Lt(self._h, _ptr(v), _ptr(out))); return out

    def prox_f(self, arg, gamma, x0):
        out = np.array(arg, dtype=np.float64, copy=True); x0 = np.ascontiguousarray(x0, dtype=np.float64)
        _lib.check(self._lib, self._lib.spock_op_prox_f(self._h, _ptr(x0), float(gamma), _ptr(out))); return out

    def prox_h_conj(self, arg, sigma):
        out = np.array(arg, dtype=np.float64, copy=True)
        _lib.check(self._lib, self._lib.spock_op_prox_h_conj(self._h, float(sigma), _ptr(out))); return out

    def spock_dot(self, a_z, a_v, b_z, b_v, gamma, sigma):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (a_z, a_v, b_z, b_v)]
        out = C.c_double(0.)
        _lib.check(self._lib, self._lib.spock_op_dot(self._h, *[_ptr(x) for x in arrs], float(gamma), float(sigma), C.byref(out)))
        return out.value

    def T(self, z, v, gamma, sigma, x0):
        z = np.ascontiguousarray(z, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        zo = np.empty(self.sizes.nz); vo = np.empty(self.sizes.nv)
        _lib.check(self._lib, self._lib.spock_op_T(self._h, _ptr(x0), _ptr(z), _ptr(v), float(gamma), float(sigma), _ptr(zo), _ptr(vo)))
        return zo, vo


def build_model(scen_tree, cost, dynamics, rms, constraints, solver_options: SolverOptions = SolverOptions(CP, None),
                **kw) -> Model:
    """model.jl:273-297.  Keyword extensions: batch, device, L_norm, max_iter, split_stage, graph_iters, lib_path, layout
    ("instance" / "interleaved": how the batch is laid out in HBM; default: interleaved from 320 instances on), per_instance_dynamics
    (every instance its own matrix pool: Model.update_dynamics_batch gives each instance its own draw of A_w, B_w), dtype ("f64", or
    "f32": libspock_amd_f32.so, the same kernels in single precision -- the reference itself is Float64 only)."""
    if solver_options.algorithm not in (AlgorithmOptions.CP, AlgorithmOptions.SP):
        raise ValueError("This combination of solver options is not supported.")
    return Model(flatten_problem(scen_tree, cost, dynamics, rms, constraints), solver_options, **kw)


def solve_model(model: Model, x0, tol: float = 1e-3, verbose: VERBOSE_LEVEL = SILENT, z0=None, v0=None,
                gamma: float | None = None, sigma: float | None = None):
    """solve_model! (cpock.jl:169-189 / models/spock.jl:231-251).  Returns the per-instance results
    (the reference returns nothing and prints "Finished in k iterations."; that count is ``iter_ref``)."""
    B, nx = model.batch, model.problem_definition.nx
    x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64).reshape(-1, nx), (B, nx)))
    if (z0 is None) != (v0 is None):
        raise ValueError("z0 and v0 must be given together")       # copyto!(.., nothing) throws in the reference
    if z0 is not None:
        z0 = np.ascontiguousarray(np.broadcast_to(np.asarray(z0, dtype=np.float64).reshape(-1, model.sizes.nz), (B, model.sizes.nz)))
        v0 = np.ascontiguousarray(np.broadcast_to(np.asarray(v0, dtype=np.float64).reshape(-1, model.sizes.nv), (B, model.sizes.nv)))
    both = gamma is not None and sigma is not None                 # cp.jl:198-201
    g = float(gamma) if both else math.nan
    s = float(sigma) if both else math.nan
    res = (_lib.Result * B)()
    _lib.check(model._lib, model._lib.spock_solve(model._h, _ptr(x0), float(tol), int(verbose), _ptr(z0), _ptr(v0), g, s, res))
    model._refresh()
    model.results = [{"status": r.status, "iterations": r.iterations, "iter_ref": r.iter_ref, "xi1": r.xi1, "xi2": r.xi2,
                      "xi": r.xi, "solve_ms": r.solve_ms} for r in res]
    return model.results[0] if B == 1 else model.results


solve_model_bang = solve_model   # `solve_model!`


def mpc_simulation(model: Model, x0, w_seq, tol: float = 1e-3, verbose: VERBOSE_LEVEL = SILENT):
    """The Spock branch of examples/server_heat/mpc_simulation.jl:47-54,164-177 as ONE call: len(w_seq[b]) warm-started
    solves per instance; between solves the first-stage input is applied and x0 advanced on the GPU.
    w_seq: [batch, steps] events in [0, d).  Returns (x_traj [batch, steps+1, nx], u_traj [batch, steps, nu], results)."""
    B, nx, nu = model.batch, model.problem_definition.nx, model.problem_definition.nu
    x0 = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64).reshape(-1, nx), (B, nx)))
    w = np.ascontiguousarray(np.broadcast_to(np.asarray(w_seq, dtype=np.int32).reshape(-1, np.asarray(w_seq).shape[-1]), (B, np.asarray(w_seq).shape[-1])))
    steps = w.shape[1]
    xt = np.empty((B, steps + 1, nx)); ut = np.empty((B, steps, nu))
    res = (_lib.Result * (B * steps))()
    _lib.check(model._lib, model._lib.spock_mpc_simulate(model._h, _ptr(x0), steps, _ptr(w), float(tol), int(verbose), _ptr(xt), _ptr(ut), res))
    model._refresh()
    out = [[{"status": res[b * steps + t].status, "iterations": res[b * steps + t].iterations, "solve_ms": res[b * steps + t].solve_ms}
            for t in range(steps)] for b in range(B)]
    return xt, ut, out
