"""Randomised parity sweep (not a test, run on the GPU box): random problem shapes, costs, risk measures, split stages and
batch sizes; operator T and a short CP / SP run against the oracle.  Prints one line per failure and a summary.

    python tests/tools/fuzz_parity.py [n_cases] [seed]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = os.environ.get("SPOCK_FUZZ_LIB")   # e.g. the emulation library


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


def random_case():
    d = int(rng.choice([2, 2, 2, 3, 4]))
    N = int(rng.integers(2, {2: 10, 3: 7, 4: 5}[d] + 1))
    nx = int(rng.choice([1, 2, 3, 4, 5, 5, 5, 6, 8, 9, 12, 20]))
    if os.environ.get("SPOCK_FUZZ_LARGE"):   # lane-team kernels at awkward sizes
        nx = int(rng.choice([9, 16, 17, 24, 32, 33, 48, 64])); N = min(N, 6 if d == 2 else 4)
    nu = int(rng.integers(1, nx + 1))
    if rng.random() < 0.25 and nx in (5,):
        nu = int(rng.choice([2, 5]))
    if os.environ.get("SPOCK_FUZZ_MFMA"):    # the shapes with matrix-core sweep kernels (spock_skernels.hip) + flat node-local kernels
        nx, nu, d = [(64, 16, 2), (20, 20, 2), (20, 20, 3), (50, 50, 2)][int(rng.integers(0, 4))]
        N = int(rng.integers(4, 9)) if d == 2 else int(rng.integers(3, 6))
    tree, cost, dyn, rms, con = S.get_random_linear_specs(N, nx, nu, d=d, seed=int(rng.integers(1 << 30))) if _has_d() else S.get_random_linear_specs(N, nx, nu)
    kind = rng.choice(["diag", "dense_shared", "per_node"], p=[0.6, 0.2, 0.2])
    if os.environ.get("SPOCK_FUZZ_MFMA"):
        kind = "diag"                        # (the lane-team phases exist for shared diagonal costs)

    def spd(n):
        a = rng.standard_normal((n, n)); return 0.05 * (a @ a.T) / n + np.eye(n) * 0.1
    if kind == "dense_shared":
        cost = S.Cost([spd(nx)], [spd(nu) + np.eye(nu)], [spd(nx)])
    elif kind == "per_node" and tree.n <= 600:
        cost = S.Cost([np.eye(nx) * (0.1 + 0.01 * (k % 5)) for k in range(tree.n - 1)], [np.eye(nu) * (1 + 0.05 * (k % 3)) for k in range(tree.n - 1)],
                      [np.eye(nx) * (0.1 + 0.02 * (k % 4)) for k in range(tree.n_leaf_nodes)])
    if rng.random() < 0.3:
        p = rng.random(tree.d) + 0.2; p /= p.sum()
        rms = S.get_uniform_rms_tv_v2(p, float(rng.uniform(0.05, 0.5)), tree.d, tree.N)
    split = 0
    if N >= 4 and rng.random() < 0.6:
        split = int(rng.integers(1, N - 1))
    batch = int(rng.choice([1, 1, 2, 5]))
    return (tree, cost, dyn, rms, con), dict(split_stage=split, batch=batch), f"N={N} d={tree.d} nx={nx} nu={nu} cost={kind} split={split} batch={batch}"


def _has_d():
    import inspect
    return "d" in inspect.signature(S.get_random_linear_specs).parameters


fails = 0; worst = 0.0
for case in range(n_cases):
    try:
        specs, kw, desc = random_case()
        flat = S.flatten_problem(*specs)
        try:
            m = S.build_model(*specs, L_norm="auto", max_iter=40, lib_path=lib, **kw)
        except S.SpockError as e:
            if "split" in str(e) or "workgroup" in str(e) or "tiling" in str(e):
                continue
            raise
        B = kw["batch"]
        o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
        z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv); x0 = 0.1 * rng.standard_normal(flat.nx)
        errs = []
        if B == 1:
            zo, vo = m.T(z, v, 0.3, 0.25, x0); zr, vr = o.T(z, v, 0.3, 0.25, x0)
            errs += [relerr(zo, zr), relerr(vo, vr)]
        X0 = np.tile(x0, (B, 1)) * (1 + 0.1 * np.arange(B))[:, None]
        r = S.solve_model(m, X0 if B > 1 else x0, tol=0.0)
        ro = o.solve(x0, tol=0.0, max_iter=40)
        zz = np.asarray(m.solver_state.z).reshape(B, -1)[0]
        errs.append(relerr(zz, o.z))
        m.close()
        if rng.random() < 0.3 and flat.n <= 400:   # whole solves: the device-side termination logic, both rules, warm start
            log = bool(rng.random() < 0.5)
            mt = S.build_model(*specs, L_norm=float(o.L_norm), max_iter=1500, lib_path=lib, **kw)
            ot = O.Oracle(flat, O.CP); ot.L_norm = o.L_norm
            for xs in (x0, 1.2 * x0):
                rt = S.solve_model(mt, np.tile(xs, (B, 1)) if B > 1 else xs, tol=1e-3, verbose=S.LOG if log else S.SILENT)
                rot = ot.solve(xs, tol=1e-3, verbose=O.LOG if log else O.SILENT, max_iter=1500)
                rt0 = rt[0] if isinstance(rt, list) else rt
                if (rt0["iterations"], rt0["status"]) != (rot["iterations"], rot["status"]):
                    errs.append(1.0)
                errs.append(relerr(np.asarray(mt.solver_state.z).reshape(B, -1)[0], ot.z) * 1e-1)
            mt.close(); desc += " +solve" + ("LOG" if log else "")
        if rng.random() < 0.35:   # the SP driver on the same problem: first iterations of the residual history
            ms = S.build_model(*specs, S.SolverOptions(S.SP, None), L_norm=float(o.L_norm), max_iter=12, lib_path=lib, **kw)
            osp = O.Oracle(flat, O.SP); osp.L_norm = o.L_norm
            S.solve_model(ms, X0 if B > 1 else x0, tol=0.0)
            rsp = osp.solve(x0, tol=0.0, max_iter=12, history=50)
            h = ms.history(0); k = min(10, len(h["xi1"]), len(rsp["hist_xi1"]))
            d1 = np.abs(h["xi1"][:k] - rsp["hist_xi1"][:k]) / np.maximum(rsp["hist_xi1"][:k], 1e-300)
            if np.all(np.isfinite(rsp["hist_xi1"][:k])):
                errs.append(float(np.max(d1)) * 1e-2)      # SP amplifies rounding: 1e-7 on the history counts like 1e-9 on iterates
                if not np.array_equal(h["backtrack"][:k], rsp["hist_backtrack"][:k]):
                    errs.append(1.0)
            ms.close(); desc += " +SP"
        e = max(errs); worst = max(worst, e)
        if not (e <= 1e-9):
            fails += 1; print("FAIL", desc, errs, flush=True)
    except Exception as ex:   # noqa
        fails += 1; print("ERROR", desc if "desc" in dir() else "?", repr(ex)[:200], flush=True)
print(f"{n_cases} cases, {fails} failures, worst relative error {worst:.2e}")
