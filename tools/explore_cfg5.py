import sys; sys.path.insert(0, "/root/repo")
import numpy as np, time
import spock_jl_amd as S
specs = S.get_random_linear_specs(15, 64, 16)
t = time.time(); m = S.build_model(*specs, max_iter=20, L_norm="auto"); print("build", time.time() - t)
S.solve_model(m, np.full(64, 0.1), tol=0.0)
print({k: round(v[0] * 1e3, 1) for k, v in m.profile_kernels(10).items() if v[1]}, "split", m.sizes.split_stage, m.sizes.n_subtrees)
m.close()
