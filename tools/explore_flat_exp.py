import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
os.environ["SPOCK_AMD_FLAT_FUSE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
for lib in [None] + sorted(glob.glob("exp/libspock_*.so")):
    for name, specs, B in (("cfg4", S.get_server_heat_specs(10, 20, 2), 64), ("cfg5", S.get_random_linear_specs(15, 64, 16), 1)):
        nx = S.flatten_problem(*specs).nx
        m = S.build_model(*specs, S.SolverOptions(S.CP, None), max_iter=12, L_norm="auto", batch=B, graph_iters=12, **({"lib_path": lib} if lib else {}))
        x0 = 0.1 * (1 + 0.01 * np.arange(B))[:, None] * np.cos(np.arange(nx))[None, :]
        S.solve_model(m, x0, tol=0.0)
        seq = " ".join(f"{n}:{ms*1e3:.1f}" for n, _, ms in m.profile_sequence(12))
        print(f"{str(lib):24s} {name} batch {B:3d} [{seq}]", flush=True)
        m.close()
