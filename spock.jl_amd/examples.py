"""Benchmark / fixture problem families (host-side data generators, numpy only).

* ``get_server_heat_specs(N, nx, d)`` restates /root/reference/examples/server_heat/server_heat.jl:1-111
  (same argument order; nu = nx, B = I).  ``nu=`` (< nx) is the builder-defined extension named in
  BASELINE.json's config 2 (B = first nu columns of I, R = I_nu); ``p_ref=`` fixes the reference
  probabilities for d > 2, where the reference draws them from Julia's RNG (not reproducible).
* ``get_car_fixture()`` restates the N=3 "discretised car" fixture of the reference's own tests
  (test/model_algorithms/prox.jl:2-51, test/model_dynamics/implicit_L.jl:2-51).
* ``get_random_linear_specs`` is BASELINE.json config 5 (random stable linear system, AV@R).
* ``L_norm_sq`` is the closed form of ||L||^2 used to pick valid step sizes
  where the reference's hard-coded constants (10.0 / 3.28) do not apply.
"""
from __future__ import annotations

import numpy as np

from .problem import (Cost, Dynamics, UniformRectangle, generate_scenario_tree_uniform_branching_factor_v2,
                      get_uniform_rms_avar_v2)


def get_server_heat_specs(N: int, nx: int, d: int, alpha: float = 0.95, nu: int | None = None, p_ref=None):
    nu = nx if nu is None else nu
    scen_tree = generate_scenario_tree_uniform_branching_factor_v2(N, d, nx, nu)
    # server_heat.jl:24-37 -- Q = 0.1 I, R = I per non-root node, QN = 0.1 I per leaf (uniform: stored once)
    cost = Cost([np.eye(nx) * 1e-1], [np.eye(nu) * 1.0], [np.eye(nx) * 1e-1])
    # server_heat.jl:74-86
    A = [np.diag([1.0 + (1.0 + (j - 1) / nx) * (i - 1) / d for j in range(1, nx + 1)]) for i in range(1, d + 1)]
    f = 0.01
    for k in range(d):
        for i in range(nx):
            if i < nx - 1:
                A[k][i, i + 1] = f
            if i > 0:
                A[k][i, i - 1] = f
    B = [np.eye(nx)[:, :nu].copy() for _ in range(d)]
    dynamics = Dynamics(A, B)
    # server_heat.jl:89-95
    if p_ref is None:
        if d == 2:
            p_ref = [0.3, 0.7]
        else:
            raise ValueError("p_ref must be given for d > 2 (the reference draws it from Julia's RNG)")
    rms = get_uniform_rms_avar_v2(p_ref, alpha, d, N)
    # server_heat.jl:97-108
    constraints = UniformRectangle(-1.0, 1.0, -1.5, 1.5, scen_tree.n_leaf_nodes * nx,
                                   scen_tree.n_non_leaf_nodes * (nx + nu), nx, nu,
                                   scen_tree.n_leaf_nodes, scen_tree.n_non_leaf_nodes)
    return scen_tree, cost, dynamics, rms, constraints


def get_car_fixture(scale: float = 1.0):
    """N=3, d=2, nx=2, nu=1; ``scale`` divides the costs (src/main_profiling.jl:20-28 divides by 3.7
    before using SP so that ||L||^2 stays below the hard-coded 3.28)."""
    N, d, nx, nu = 3, 2, 2, 1
    scen_tree = generate_scenario_tree_uniform_branching_factor_v2(N, d, nx, nu)
    Q = np.array([[2.2, 0.0], [0.0, 3.7]]) / scale
    R = np.array([[3.2]]) / scale
    cost = Cost([Q], [R], [Q.copy()])
    T_s = 0.1
    A = [np.array([[1.0, T_s], [0.0, 1.0 - (i - 1) / d * T_s]]) for i in range(1, d + 1)]
    B = [np.array([[0.0], [T_s]]) for _ in range(d)]
    dynamics = Dynamics(A, B)
    rms = get_uniform_rms_avar_v2([0.3, 0.7], 0.95, d, N)
    constraints = UniformRectangle(-1.0, 1.0, -1.0, 1.0, scen_tree.n_leaf_nodes * nx,
                                   scen_tree.n_non_leaf_nodes * (nx + nu), nx, nu,
                                   scen_tree.n_leaf_nodes, scen_tree.n_non_leaf_nodes)
    return scen_tree, cost, dynamics, rms, constraints


def get_random_linear_specs(N: int, nx: int, nu: int, d: int = 2, seed: int = 2026, alpha: float = 0.95, p_ref=None):
    """BASELINE.json config 5: A_w = 0.95 G_w / rho(G_w), B_w = H_w / sqrt(nx), G, H iid N(0,1) from
    numpy PCG64(seed); Q = 0.1 I, R = I, QN = 0.1 I; AV@R; box +-1 / +-1.5."""
    rng = np.random.Generator(np.random.PCG64(seed))
    scen_tree = generate_scenario_tree_uniform_branching_factor_v2(N, d, nx, nu)
    A, B = [], []
    for _ in range(d):
        G = rng.standard_normal((nx, nx))
        H = rng.standard_normal((nx, nu))
        rho = np.max(np.abs(np.linalg.eigvals(G)))
        A.append(0.95 * G / rho)
        B.append(H / np.sqrt(nx))
    cost = Cost([np.eye(nx) * 1e-1], [np.eye(nu) * 1.0], [np.eye(nx) * 1e-1])
    if p_ref is None:
        p_ref = [0.3, 0.7] if d == 2 else list(np.arange(1, d + 1) / np.arange(1, d + 1).sum())
    rms = get_uniform_rms_avar_v2(p_ref, alpha, d, N)
    constraints = UniformRectangle(-1.0, 1.0, -1.5, 1.5, scen_tree.n_leaf_nodes * nx,
                                   scen_tree.n_non_leaf_nodes * (nx + nu), nx, nu,
                                   scen_tree.n_leaf_nodes, scen_tree.n_non_leaf_nodes)
    return scen_tree, cost, Dynamics(A, B), rms, constraints


def L_norm_sq(scen_tree, cost, rms) -> float:
    """Closed form of ||L||^2 for uniform costs: L'L is block diagonal over
    x (non-leaf / leaf), u, tau and (s_i, y_i)."""
    d = scen_tree.d
    Q = np.asarray(cost.Q[0]); R = np.asarray(cost.R[0]); QN = np.asarray(cost.QN[0])
    if len(cost.Q) != 1 or len(cost.R) != 1 or len(cost.QN) != 1:
        raise NotImplementedError("closed form implemented for uniform costs only")
    beta = float(np.dot(rms[0].b, rms[0].b))
    cands = [1.0 + d * np.linalg.eigvalsh(Q).max(), 1.0 + np.linalg.eigvalsh(QN).max(),
             1.0 + d * np.linalg.eigvalsh(R).max(), 0.5, 1.0 + (beta + np.sqrt(beta * beta + 4.0 * beta)) / 2.0]
    return float(max(cands))
