"""Exploration (not a test): one SP and one CP solve of BASELINE config 4 (nx = nu = 20, N = 10) for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(10, 20, 2)
for alg in (S.SP, S.CP):
    m = S.build_model(*specs, S.SolverOptions(alg, None))
    S.solve_model(m, np.full(20, 0.1), tol=1e-3); m.reset()
    r = S.solve_model(m, np.full(20, 0.1), tol=1e-3)
    print(alg.name, r["iterations"], r["solve_ms"], r["solve_ms"] / r["iterations"] * 1e3, "us/iter")
    m.close()
