"""Problem-data layer: host-side mirror of the reference's plain data types.

Mirrors (names, argument order and meaning) of
  * ``generate_scenario_tree_uniform_branching_factor_v2``  (/root/reference/src/scenario_tree.jl:61-109)
  * ``Cost(Q, R, QN)``                                      (src/cost.jl:3-7)
  * ``Dynamics(A, B)``                                      (src/dynamics.jl:1-4)
  * ``RiskMeasure(E, F, b, K)`` / ``ConvexCone``            (src/risk_measures.jl:73-84)
  * ``get_uniform_rms_v2`` / ``get_uniform_rms_avar_v2`` / ``get_uniform_rms_tv_v2`` (src/risk_measures.jl:93-147)
  * ``UniformRectangle``                                    (src/constraints.jl:11-22)

Indices are 0-based here (node 0 is the root, events are 0..d-1); the reference is 1-based.
``flatten_problem`` turns these objects into the flat, column-major arrays the C ABI
(include/spock_amd.h, ``spock_problem_t``) takes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

# cone kinds shared with include/spock_amd.h (SPOCK_CONE_*)
CONE_ZEROS, CONE_NONNEG, CONE_NONPOS, CONE_REALS, CONE_SOC = 0, 1, 2, 3, 4


@dataclass
class ScenarioTree:
    """Uniform-branching scenario tree, breadth-first numbering (scenario_tree.jl:44-55)."""

    N: int                      # number of stages INCLUDING the root: leaves = d**(N-1)
    d: int
    n: int
    n_non_leaf_nodes: int
    n_leaf_nodes: int
    leaf_node_min_index: int    # 0-based
    leaf_node_max_index: int    # 0-based, inclusive
    min_index_per_timestep: np.ndarray
    anc: np.ndarray             # int32[n], parent of each node, -1 for the root
    w: np.ndarray               # int32[n], event of the edge parent->node, -1 for the root
    nx: int = 0
    nu: int = 0

    @property
    def child_mapping(self) -> Dict[int, List[int]]:
        return {i: list(range(i * self.d + 1, i * self.d + self.d + 1)) for i in range(self.n_non_leaf_nodes)}

    @property
    def anc_mapping(self) -> Dict[int, int]:
        return {j: int(self.anc[j]) for j in range(1, self.n)}

    def children(self, i: int) -> range:
        return range(i * self.d + 1, i * self.d + self.d + 1)


def generate_scenario_tree_uniform_branching_factor_v2(N: int, d: int, nx: int, nu: int) -> ScenarioTree:
    """scenario_tree.jl:61-109.  Children of (0-based) node i are i*d+1 .. i*d+d; the event of
    (1-based) node j is ``(j % d) + 1``, i.e. 0-based node j0 has event ``(j0 + 1) % d``."""
    if d <= 1:
        raise ValueError(f"Branching factor d must be larger than 1, but is {d}.")
    n_total = (d ** N - 1) // (d - 1)
    n_leafs = d ** (N - 1)
    n_non_leafs = (d ** (N - 1) - 1) // (d - 1)
    idx = np.arange(n_total, dtype=np.int64)
    anc = ((idx - 1) // d).astype(np.int32)
    anc[0] = -1
    w = ((idx + 1) % d).astype(np.int32)
    w[0] = -1
    first = np.array([(d ** t - 1) // (d - 1) for t in range(N)], dtype=np.int64)
    return ScenarioTree(N, d, n_total, n_non_leafs, n_leafs, n_non_leafs, n_total - 1, first, anc, w, nx, nu)


@dataclass
class Cost:
    """Per-node cost matrices (cost.jl:3-7): Q[j-1], R[j-1] belong to the non-root node j (edge
    anc(j)->j); QN[i] to the i-th leaf.  A single matrix may be passed instead of a list when all
    nodes share it."""

    Q: Sequence[np.ndarray]
    R: Sequence[np.ndarray]
    QN: Sequence[np.ndarray]


@dataclass
class Dynamics:
    """A[w], B[w] indexed by the event w of the edge (dynamics.jl:1-4)."""

    A: Sequence[np.ndarray]
    B: Sequence[np.ndarray]


@dataclass
class ConvexCone:
    subcones: List[tuple]       # [(kind, dim), ...]


@dataclass
class RiskMeasure:
    E: np.ndarray
    F: np.ndarray
    b: np.ndarray
    K: ConvexCone


def get_uniform_rms_v2(E, F, b, K, d, N) -> List[RiskMeasure]:
    """risk_measures.jl:93-102 -- the same arrays are aliased into every entry."""
    rm = RiskMeasure(np.asarray(E, dtype=np.float64), np.asarray(F, dtype=np.float64), np.asarray(b, dtype=np.float64), K)
    return [rm for _ in range((d ** (N - 1) - 1) // (d - 1))]


def get_uniform_rms_avar_v2(p, alpha, d, N) -> List[RiskMeasure]:
    """AV@R_alpha with reference probabilities p (risk_measures.jl:110-119):
    E = [alpha I_d; -I_d; 1'], F = 0, b = [p; 0_d; 1], K = R+^{2d} x {0}."""
    p = np.asarray(p, dtype=np.float64)
    E = np.vstack([alpha * np.eye(d), -np.eye(d), np.ones((1, d))])
    F = np.zeros((2 * d + 1, d))
    b = np.concatenate([p, np.zeros(d), [1.0]])
    return get_uniform_rms_v2(E, F, b, ConvexCone([(CONE_NONNEG, 2 * d), (CONE_ZEROS, 1)]), d, N)


def get_nonuniform_rms_avar_v2(d, N, rng=None) -> List[RiskMeasure]:
    """One AV@R per non-leaf node with its own level and reference probabilities (risk_measures.jl:127-136):
    E_i = [alpha_i I; -I; 1'], b_i = [p_i; 0; 1], alpha_i ~ U(0, 1), p_i a random probability vector.  The reference draws
    them from Julia's global RNG (not reproducible); here from ``rng`` (numpy Generator, default PCG64(0))."""
    rng = np.random.default_rng(0) if rng is None else rng
    K = ConvexCone([(CONE_NONNEG, 2 * d), (CONE_ZEROS, 1)])
    out = []
    for _ in range((d ** (N - 1) - 1) // (d - 1)):
        alpha = rng.random()
        v = rng.random(d)
        E = np.vstack([alpha * np.eye(d), -np.eye(d), np.ones((1, d))])
        out.append(RiskMeasure(E, np.zeros((2 * d + 1, d)), np.concatenate([v / v.sum(), np.zeros(d), [1.0]]), K))
    return out


def get_uniform_rms_tv_v2(p, r, d, N) -> List[RiskMeasure]:
    """Total-variation ball (risk_measures.jl:138-147)."""
    p = np.asarray(p, dtype=np.float64)
    E = np.vstack([0.5 * np.eye(d), -0.5 * np.eye(d), np.zeros((d, d))])
    F = np.vstack([-np.eye(d), -np.eye(d), np.eye(d)])
    b = np.concatenate([0.5 * p, -0.5 * p, np.full(d, float(r))])
    return get_uniform_rms_v2(E, F, b, ConvexCone([(CONE_NONNEG, 3 * d)]), d, N)


@dataclass
class UniformRectangle:
    """constraints.jl:11-22 (same positional order)."""

    x_min: float
    x_max: float
    u_min: float
    u_max: float
    nGamma_leaf: int
    nGamma_nonleaf: int
    nx: int
    nu: int
    n_leafs: int
    n_nonleafs: int


@dataclass
class FlatProblem:
    """Flat column-major arrays = the fields of ``spock_problem_t`` (include/spock_amd.h)."""

    N: int
    d: int
    nx: int
    nu: int
    ny: int
    anc: np.ndarray
    w: np.ndarray
    A: np.ndarray
    B: np.ndarray
    nQ: int
    nR: int
    nQN: int
    Q: np.ndarray
    R: np.ndarray
    QN: np.ndarray
    E: np.ndarray
    F: np.ndarray
    b: np.ndarray
    cone_kind: np.ndarray
    cone_dim: np.ndarray
    x_min: float
    x_max: float
    u_min: float
    u_max: float
    n_rms: int = 1          # 1: E, F, b hold one risk measure shared by all non-leaf nodes; else one block per non-leaf node
    n: int = field(init=False)
    nn: int = field(init=False)
    nl: int = field(init=False)
    nz: int = field(init=False)
    nv: int = field(init=False)

    def __post_init__(self):
        d, N = self.d, self.N
        self.n = (d ** N - 1) // (d - 1)
        self.nl = d ** (N - 1)
        self.nn = (d ** (N - 1) - 1) // (d - 1)
        n, nn, nl, nx, nu, ny = self.n, self.nn, self.nl, self.nx, self.nu, self.ny
        # implicit_l.jl:5-15 and :17-44
        self.nz = nn * nu + n * nx + n + nn * ny + (n - 1)
        self.nv = nn * ny + nn + nx * (n - 1) + nu * (n - 1) + 2 * (n - 1) + nn * (nx + nu) + nx * nl + 2 * nl + nl * nx

    # --- index maps (implicit_l.jl:106-158), 0-based slices into z
    def z_slices(self):
        n, nn, nx, nu, ny = self.n, self.nn, self.nx, self.nu, self.ny
        ox = 0
        ou = n * nx
        os_ = ou + nn * nu
        ot = os_ + n
        oy = ot + (n - 1)
        return {"x": slice(ox, ou), "u": slice(ou, os_), "s": slice(os_, ot), "tau": slice(ot, oy), "y": slice(oy, oy + nn * ny)}

    # --- v block offsets (cpock.jl:51-60), 0-based
    def v_offsets(self):
        n, nn, nl, nx, nu, ny = self.n, self.nn, self.nl, self.nx, self.nu, self.ny
        o = {"v1": 0}
        o["v2"] = nn * ny
        o["v3"] = o["v2"] + nn
        o["v4"] = o["v3"] + nx * (n - 1)
        o["v5"] = o["v4"] + nu * (n - 1)
        o["v6"] = o["v5"] + (n - 1)
        o["v7"] = o["v6"] + (n - 1)
        o["v11"] = o["v7"] + nn * (nx + nu)
        o["v12"] = o["v11"] + nl * nx
        o["v13"] = o["v12"] + nl
        o["v14"] = o["v13"] + nl
        o["end"] = o["v14"] + nl * nx
        return o


def _stack_colmajor(mats: Sequence[np.ndarray], rows: int, cols: int) -> np.ndarray:
    out = np.empty((len(mats), rows * cols), dtype=np.float64)
    for k, m in enumerate(mats):
        m = np.asarray(m, dtype=np.float64)
        if m.shape != (rows, cols):
            raise ValueError(f"matrix {k} has shape {m.shape}, expected {(rows, cols)}")
        out[k] = m.reshape(-1, order="F")
    return np.ascontiguousarray(out.reshape(-1))


def _dedupe(mats, count: int, rows: int, cols: int):
    """Return (k, flat) with k = 1 when every matrix equals the first one (stored once)."""
    if isinstance(mats, np.ndarray) and mats.ndim == 2:
        mats = [mats]
    mats = list(mats)
    if len(mats) == 1:
        return 1, _stack_colmajor(mats, rows, cols)
    if len(mats) != count:
        raise ValueError(f"expected 1 or {count} matrices, got {len(mats)}")
    first = np.asarray(mats[0], dtype=np.float64)
    if all((m is mats[0]) or np.array_equal(np.asarray(m, dtype=np.float64), first) for m in mats[1:]):
        return 1, _stack_colmajor([first], rows, cols)
    return count, _stack_colmajor(mats, rows, cols)


def flatten_problem(scen_tree: ScenarioTree, cost: Cost, dynamics: Dynamics, rms: Sequence[RiskMeasure],
                    constraints: UniformRectangle) -> FlatProblem:
    """Same five positional objects ``build_model`` takes (model.jl:273-279) -> flat arrays."""
    B0 = np.asarray(dynamics.B[0], dtype=np.float64)
    nx, nu = B0.shape                       # cpock.jl:12
    d, N = scen_tree.d, scen_tree.N
    if len(dynamics.A) != d or len(dynamics.B) != d:
        raise ValueError("Dynamics must hold one (A, B) pair per event")
    rm = rms[0]
    uniform = True
    for other in rms[1:]:
        if other is rm:
            continue
        if other.K.subcones != rm.K.subcones or np.asarray(other.b).shape != np.asarray(rm.b).shape:
            raise NotImplementedError("risk measures must share length(b) and the cone K (implicit_l.jl:14,19)")
        if not (np.array_equal(other.E, rm.E) and np.array_equal(other.F, rm.F) and np.array_equal(other.b, rm.b)):
            uniform = False
    if not uniform and len(rms) != scen_tree.n_non_leaf_nodes:
        raise ValueError("one risk measure per non-leaf node expected")
    ny = int(np.asarray(rm.b).shape[0])
    nQ, Q = _dedupe(cost.Q, scen_tree.n - 1, nx, nx)
    nR, R = _dedupe(cost.R, scen_tree.n - 1, nu, nu)
    nQN, QN = _dedupe(cost.QN, scen_tree.n_leaf_nodes, nx, nx)
    kinds = np.array([k for k, _ in rm.K.subcones], dtype=np.int32)
    dims = np.array([m for _, m in rm.K.subcones], dtype=np.int32)
    if int(dims.sum()) != ny:
        raise ValueError("cone dimensions must add up to length(b)")
    return FlatProblem(
        N=N, d=d, nx=nx, nu=nu, ny=ny,
        anc=np.ascontiguousarray(scen_tree.anc, dtype=np.int32), w=np.ascontiguousarray(scen_tree.w, dtype=np.int32),
        A=_stack_colmajor(dynamics.A, nx, nx), B=_stack_colmajor(dynamics.B, nx, nu),
        nQ=nQ, nR=nR, nQN=nQN, Q=Q, R=R, QN=QN,
        E=np.ascontiguousarray(np.concatenate([np.asarray(r.E, dtype=np.float64).reshape(-1, order="F") for r in (rms if not uniform else [rm])])),
        F=np.ascontiguousarray(np.concatenate([np.asarray(r.F, dtype=np.float64).reshape(-1, order="F") for r in (rms if not uniform else [rm])])),
        b=np.ascontiguousarray(np.concatenate([np.asarray(r.b, dtype=np.float64) for r in (rms if not uniform else [rm])])),
        cone_kind=kinds, cone_dim=dims,
        x_min=float(constraints.x_min), x_max=float(constraints.x_max),
        u_min=float(constraints.u_min), u_max=float(constraints.u_max),
        n_rms=1 if uniform else len(rms),
    )
