import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
import spock_jl_amd as S
for nu in (2, 5):
    specs = S.get_server_heat_specs(10, 5, 2, nu=nu)
    for B in (1, 8, 16, 32, 44, 64, 96):
        row = []; seqs = []
        for rl in ("1", "0"):
            os.environ["SPOCK_AMD_RLANE"] = rl
            m = S.build_model(*specs, batch=B, max_iter=600, layout="instance")
            x0 = np.full((B, 5), 0.1) if B > 1 else np.full(5, 0.1)
            S.solve_model(m, x0, tol=0.0); m.reset()
            t = time.perf_counter(); S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
            row.append(dt / 600 * 1e6); seqs.append([n for n, _, _ in m.profile_sequence(1)][-1] + "/t0=%d" % m.sizes.split_stage)
            m.close()
        print(f"N=10 nu={nu} B={B:3d}  rlane {row[0]:8.2f} us/iter ({seqs[0]})   other {row[1]:8.2f} us/iter ({seqs[1]})  ratio {row[1]/row[0]:.2f}", flush=True)
