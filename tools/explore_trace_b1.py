"""Exploration (not a test): one B=1 CP solve (500 iterations) for a rocprofv3 kernel trace (durations and gaps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
m = S.build_model(*specs, batch=1, max_iter=500)
S.solve_model(m, np.full(5, 0.1), tol=0.0)
r = S.solve_model(m, np.full(5, 0.1), tol=0.0)
print(r["iterations"], r["solve_ms"], r["solve_ms"] / r["iterations"] * 1e3, "us/iter")
m.close()
