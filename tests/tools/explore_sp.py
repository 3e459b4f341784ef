"""Exploration (not a test): SP (SuperMann) timing on the GPU vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

for (N, nu, B) in ((13, 5, 1), (13, 5, 8), (10, 5, 1)):
    specs = S.get_server_heat_specs(N, 5, 2, nu=nu)
    flat = S.flatten_problem(*specs)
    x0 = np.full((B, 5), 0.1) * (1 + 0.01 * np.arange(B))[:, None]
    for alg in (S.SP, S.CP):
        m = S.build_model(*specs, S.SolverOptions(alg, None), batch=B)
        S.solve_model(m, x0, tol=1e-3); m.reset()
        t = time.time(); r = S.solve_model(m, x0, tol=1e-3); wall = time.time() - t
        rs = r if isinstance(r, list) else [r]
        its = [q["iterations"] for q in rs]
        print(f"N={N} nu={nu} B={B} {alg.name}: iterations {its} status {[q['status'] for q in rs]} device {rs[0]['solve_ms']:.2f} ms wall {wall*1e3:.2f} ms "
              f"-> {rs[0]['solve_ms']/max(its)*1e3:.1f} us/iter", flush=True)
        m.close()
    if B == 1:
        for alg, name in ((O.SP, "SP"), (O.CP, "CP")):
            o = O.Oracle(flat, alg); t = time.time(); ro = o.solve(x0[0], tol=1e-3); dt = time.time() - t
            print(f"   oracle {name}: {ro['iterations']} iterations {dt*1e3:.0f} ms -> {dt/ro['iterations']*1e3:.2f} ms/iter", flush=True)
