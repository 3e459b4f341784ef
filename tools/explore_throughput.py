"""Exploration script (not a test): timing of the CP iteration for several batch sizes / split stages."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

def run(N, nx, nu, B, split, M=400, graph=None):
    specs = S.get_server_heat_specs(N, nx, 2, nu=nu)
    kw = {}
    if graph is not None: kw["graph_iters"] = graph
    m = S.build_model(*specs, batch=B, split_stage=split, max_iter=M, **kw)
    x0 = np.full((B, nx), 0.1) * (1 + 0.01 * np.arange(B))[:, None]
    S.solve_model(m, x0, tol=0.0)           # warm-up (graph capture)
    t = time.time(); r = S.solve_model(m, x0, tol=0.0); wall = time.time() - t
    r0 = r if B == 1 else r[0]
    ms = r0["solve_ms"]
    prof = m.profile_kernels(40)
    print(f"N={N} nx={nx} nu={nu} B={B} split={m.sizes.split_stage} nsub={m.sizes.n_subtrees} G={m.sizes.graph_iters}: "
          f"{ms/M*1e3:8.2f} us/iter  {B*M/ms*1e3:12.0f} inst-it/s  wall {wall*1e3:.1f} ms | "
          + " ".join(f"{k}={v[0]*1e3:.1f}us" for k, v in prof.items() if v[1]), flush=True)
    m.close()

if __name__ == "__main__":
    for split in (5, 6, 7, 8):
        run(13, 5, 2, 1, split)
    run(13, 5, 2, 1, 6, graph=-32)
    run(13, 5, 2, 1, 6, graph=400)
    for B in (2, 8, 32, 128, 512):
        run(13, 5, 2, B, 6)
    for B in (32, 128):
        run(13, 5, 2, B, 7)
        run(13, 5, 2, B, 5)
    run(13, 5, 5, 1, 6); run(13, 5, 5, 64, 6)
    run(7, 5, 5, 1, 0); run(10, 5, 5, 1, 0); run(10, 5, 5, 1, 5); run(10, 20, 20, 1, 0)
