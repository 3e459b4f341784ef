import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
for B in (128, 256, 512, 1024, 2048):
    M = 102
    m = S.build_model(*specs, batch=B, max_iter=M, graph_iters=M)
    x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
    S.solve_model(m, x0, tol=0.0)
    r = S.solve_model(m, x0, tol=0.0)
    ms = r[0]["solve_ms"]
    print(f"B={B}: split {m.sizes.split_stage} fused {m.sizes.fused_iteration} {ms/M*1e3:8.1f} us/iter {B*M/ms*1e3:10.0f} it/s", flush=True)
    m.close()
