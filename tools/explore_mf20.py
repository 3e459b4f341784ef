"""Exploration script (not a test harness): BASELINE config 4 (nx = nu = 20, N = 10) with the sweep levels on the fp64 matrix
cores against the scalar lane-team levels (SPOCK_AMD_NO_MFMA=1): parity with the oracle and time per iteration.
    python tools/explore_mf20.py [lib]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
kw = {"lib_path": lib} if lib else {}
specs = S.get_server_heat_specs(10, 20, 2)
if os.environ.get("CFG3"):
    specs = S.get_server_heat_specs(9, 20, 3, p_ref=[0.2, 0.3, 0.5])
flat = S.flatten_problem(*specs)
x0 = 0.1 * np.cos(np.arange(20))
for nomf in (os.environ.get("SPOCK_AMD_NO_MFMA"),):      # (the library reads the switch once per process: run the script twice)
    tag = "scalar levels" if nomf else "MFMA levels  "
    m = S.build_model(*specs, max_iter=60, **kw)
    S.solve_model(m, x0, tol=0.0)
    r = S.solve_model(m, x0, tol=0.0)
    z = m.solver_state.z.copy()
    print(tag, "CP single instance:", round(r["solve_ms"] / 60 * 1e3, 1), "us/iter", flush=True)
    m.close()
    if os.environ.get("WITH_ORACLE"):
        from oracle import oracle as O
        o = O.Oracle(flat, O.CP); o.solve(x0, tol=0.0, max_iter=60); o.solve(x0, tol=0.0, max_iter=60)
        print(tag, "parity after 120 iterations (warm): rel err", float(np.max(np.abs(z - o.z)) / max(1.0, float(np.max(np.abs(o.z))))), flush=True)
    for B in (() if os.environ.get("CFG3") else (8, 64)):
        ms = S.build_model(*specs, S.SolverOptions(S.SP, None), batch=B, **kw)
        X = 0.1 * (1.0 + 0.01 * np.arange(B))[:, None] * np.ones((B, 20))
        w = np.stack([np.random.default_rng(s).integers(0, 2, size=6) for s in range(B)]).astype(np.int32)
        S.mpc_simulation(ms, X, w, tol=1e-3); ms.reset()
        t = time.perf_counter(); xt, ut, res = S.mpc_simulation(ms, X, w, tol=1e-3); dt = time.perf_counter() - t
        its = sum(r_["iterations"] for rr in res for r_ in (rr if isinstance(rr, list) else [rr]))
        print(tag, f"SP closed loop, {B} simulations x 6 steps: {dt*1e3:.1f} ms, {its} SP iterations, final |x| {np.abs(xt[:, -1]).max():.4f}", flush=True)
        ms.close()
