import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
os.environ["SPOCK_AMD_FLAT_FUSE"] = "0"
specs = S.get_server_heat_specs(10, 20, 2)
for lib in [None] + sorted(glob.glob("exp/libspock_exp*.so")):
    for B in (1, 64):
        m = S.build_model(*specs, S.SolverOptions(S.CP, None), max_iter=12, L_norm=3.28, batch=B, graph_iters=12, **({"lib_path": lib} if lib else {}))
        x0 = 0.1 * (1 + 0.01 * np.arange(B))[:, None] * np.cos(np.arange(20))[None, :]
        S.solve_model(m, x0, tol=0.0)
        seq = " ".join(f"{n}:{ms*1e3:.1f}" for n, _, ms in m.profile_sequence(12))
        print(f"{str(lib):28s} batch {B:3d} [{seq}]", flush=True)
        m.close()
