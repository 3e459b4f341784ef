import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
specs = S.get_server_heat_specs(10, 20, 2) if (len(sys.argv) < 3 or sys.argv[2] == "cfg4") else S.get_random_linear_specs(15, 64, 16)
nx = S.flatten_problem(*specs).nx
m = S.build_model(*specs, S.SolverOptions(S.CP, None), max_iter=12, L_norm="auto", batch=B, graph_iters=12)
x0 = 0.1 * (1 + 0.01 * np.arange(B))[:, None] * np.cos(np.arange(nx))[None, :]
S.solve_model(m, x0, tol=0.0)
m.close()
