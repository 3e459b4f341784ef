"""Exploration (not a test): one single-instance SP solve for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(13, 5, 2)
m = S.build_model(*specs, S.SolverOptions(S.SP, None))
S.solve_model(m, np.full(5, 0.1), tol=1e-3); m.reset()
r = S.solve_model(m, np.full(5, 0.1), tol=1e-3)
print(r["iterations"], r["solve_ms"], r["solve_ms"] / r["iterations"] * 1e3, "us/iter")
