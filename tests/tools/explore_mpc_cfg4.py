"""Exploration (not a test): BASELINE config 4 -- mpc_simulation.jl (nx = nu = 20, d = 2, N = 10, SP, tol 1e-3, 20 steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

specs = S.get_server_heat_specs(10, 20, 2)
flat = S.flatten_problem(*specs)
rng = np.random.default_rng(0)
B, T = int(os.environ.get("B", "1")), 20
w = rng.integers(0, 2, size=(B, T))
x0 = np.full((B, 20), 0.1)
for alg in (S.SP, S.CP):
    m = S.build_model(*specs, S.SolverOptions(alg, None), batch=B)
    t = time.time(); xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3); wall = time.time() - t
    its = [r["iterations"] for r in res[0]]; ms = [round(r["solve_ms"], 1) for r in res[0]]
    print(f"GPU {alg.name} B={B}: wall {wall:.2f}s iterations/step {its} ms/step {ms}", flush=True)
    m.close()
if os.environ.get("ORACLE", "1") == "1":
    o = O.Oracle(flat, O.SP); x = x0[0].copy(); A, Bm = specs[2].A, specs[2].B
    its, tms = [], []
    for t_ in range(T):
        t = time.time(); r = o.solve(x, tol=1e-3); tms.append(round((time.time() - t) * 1e3, 1)); its.append(r["iterations"])
        u = o.z[flat.z_slices()["u"]][:20].copy(); x = A[w[0, t_]] @ x + Bm[w[0, t_]] @ u
    print("oracle SP: iterations/step", its, "ms/step", tms, flush=True)
