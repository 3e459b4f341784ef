import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(10, 20, 2)
for thr in ("", "1024", "512", "256", "128"):
    if thr: os.environ["SPOCK_AMD_SWEEP_THREADS"] = thr
    for B in (8, 64, 256):
        m = S.build_model(*specs, S.SolverOptions(S.CP, None), max_iter=60, L_norm=3.28, batch=B, graph_iters=60)
        x0 = 0.1 * (1 + 0.01 * np.arange(B))[:, None] * np.cos(np.arange(20))[None, :]
        S.solve_model(m, x0, tol=0.0); m.reset()
        r = S.solve_model(m, x0, tol=0.0)
        seq = " ".join(f"{n}:{ms*1e3:.1f}" for n, _, ms in m.profile_sequence(12))
        print(f"threads {thr or 'auto':5s} batch {B:3d}: {r[0]['solve_ms']/60*1e3:8.1f} us/iter [{seq}]", flush=True)
        m.close()
