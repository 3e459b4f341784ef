"""Independent check of a solution: the constraints of the reference's validation model
(build_model_mosek, /root/reference/src/models/model_mosek.jl:9-131) evaluated on an iterate z = x|u|s|tau|y with plain
numpy -- no solver, no use of the operators L / L' / prox of the path being checked:

    min s_1   s.t.   x_1 = x0,  x_j = A_w x_anc(j) + B_w u_anc(j)                      (model_mosek.jl:32-45)
                     x_min <= x <= x_max,  u_min <= u <= u_max                         (:47-84)
                     x_a' Q_j x_a + u_a' R_j u_a <= tau_j (j >= 2, a = anc(j)),   x_i' QN_i x_i <= s_i (leaves)   (:9-28)
                     y_i in K*,  b'y_i <= s_i,  E'y_i = tau_ch + s_ch (one row per child),  F'y_i = 0            (:88-131)

``check_solution`` returns the largest violation of every group (0 = satisfied) and the objective s_1.  The iterates
of the splitting satisfy the linear equalities exactly (they are projected onto them every iteration) and the
inequality / cone constraints up to the termination tolerance."""
from __future__ import annotations

import numpy as np

from .problem import CONE_NONNEG, CONE_NONPOS, CONE_REALS, CONE_SOC, CONE_ZEROS, FlatProblem


def _dual_cone_violation(y: np.ndarray, kinds, dims) -> float:
    """Distance-like violation of y (nn x ny) from the DUAL of K = K_1 x ... (MOI.dual_set, model_mosek.jl:103-107)."""
    worst, off = 0.0, 0
    for kind, dim in zip(kinds, dims):
        blk = y[:, off:off + dim]
        if kind == CONE_ZEROS:            # dual: Reals -- free
            v = 0.0
        elif kind == CONE_REALS:          # dual: Zeros
            v = float(np.max(np.abs(blk), initial=0.0))
        elif kind == CONE_NONNEG:
            v = float(np.max(-blk, initial=0.0))
        elif kind == CONE_NONPOS:
            v = float(np.max(blk, initial=0.0))
        elif kind == CONE_SOC:            # self-dual, MOI ordering (t, x)
            v = float(np.max(np.linalg.norm(blk[:, 1:], axis=1) - blk[:, 0], initial=0.0))
        else:
            raise ValueError(f"unknown cone kind {kind}")
        worst = max(worst, v)
        off += dim
    return max(worst, 0.0)


def check_solution(flat: FlatProblem, x0, z) -> dict:
    n, nn, nl, nx, nu, ny, d = flat.n, flat.nn, flat.nl, flat.nx, flat.nu, flat.ny, flat.d
    z = np.asarray(z, dtype=np.float64)
    if z.shape != (flat.nz,):
        raise ValueError("z has the wrong length")
    sl = flat.z_slices()
    x = z[sl["x"]].reshape(n, nx); u = z[sl["u"]].reshape(nn, nu)
    s = z[sl["s"]]; tau = z[sl["tau"]]; y = z[sl["y"]].reshape(nn, ny)
    A = flat.A.reshape(d, nx, nx).transpose(0, 2, 1)      # column-major blocks -> A[w][row, col]
    B = flat.B.reshape(d, nu, nx).transpose(0, 2, 1)
    j = np.arange(1, n); anc = (j - 1) // d; w = (j + 1) % d   # scenario_tree.jl:77,83-95 (0-based)
    out = {"objective": float(s[0])}
    out["x0"] = float(np.max(np.abs(x[0] - np.asarray(x0, dtype=np.float64))))
    xn = np.einsum("jrc,jc->jr", A[w], x[anc]) + np.einsum("jrc,jc->jr", B[w], u[anc])
    out["dynamics"] = float(np.max(np.abs(x[1:] - xn)))
    out["box"] = float(max(np.max(x - flat.x_max), np.max(flat.x_min - x), np.max(u - flat.u_max), np.max(flat.u_min - u), 0.0))

    def quad(M, k, cnt, vec):   # vec_i' M_i vec_i with one shared matrix or one per row
        Ms = M.reshape(-1, k, k).transpose(0, 2, 1)
        Ms = np.broadcast_to(Ms, (cnt, k, k)) if Ms.shape[0] == 1 else Ms
        return np.einsum("ir,irc,ic->i", vec, Ms, vec)

    out["cost_nonleaf"] = float(max(np.max(quad(flat.Q, nx, n - 1, x[anc]) + quad(flat.R, nu, n - 1, u[anc]) - tau), 0.0))
    out["cost_leaf"] = float(max(np.max(quad(flat.QN, nx, nl, x[nn:]) - s[nn:]), 0.0))
    E = flat.E.reshape(d, ny).T; F = flat.F.reshape(-1, ny).T          # ny x d, ny x nF (column-major)
    out["risk_dual_cone"] = _dual_cone_violation(y, flat.cone_kind, flat.cone_dim)
    out["risk_epigraph_b"] = float(max(np.max(y @ flat.b - s[:nn]), 0.0))
    ch = np.arange(nn)[:, None] * d + 1 + np.arange(d)[None, :]        # children of node i
    out["risk_epigraph_E"] = float(np.max(np.abs(y @ E - (tau[ch - 1] + s[ch]))))
    out["risk_epigraph_F"] = float(np.max(np.abs(y @ F), initial=0.0))
    return out
