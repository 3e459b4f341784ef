"""Exploration (not a test): CP iteration time, role-lane kernel (SPOCK_AMD_RLANE=1) against SUB_BA / the plain sequence (=0), over the batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

for nu in (2, 5):
    specs = S.get_server_heat_specs(13, 5, 2, nu=nu)
    for B in (1, 2, 3, 4, 6, 8, 16, 32, 64):
        row = []
        for rl in ("1", "0"):
            os.environ["SPOCK_AMD_RLANE"] = rl
            m = S.build_model(*specs, batch=B, max_iter=600, layout="instance")
            x0 = np.full((B, 5), 0.1)
            S.solve_model(m, x0, tol=0.0); m.reset()
            t = time.perf_counter(); r = S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
            row.append(dt / 600 * 1e6)
            m.close()
        print(f"nu={nu} B={B:3d}  rlane {row[0]:8.2f} us/iter   other {row[1]:8.2f} us/iter   ratio {row[1]/row[0]:.2f}", flush=True)
