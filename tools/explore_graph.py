"""Exploration script (not a test): ms per CP iteration against the iterations per captured graph and the tile groups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
for tg in (1, 2, 4):
    for gi in (12, 24, 48, 96):
        os.environ["SPOCK_AMD_ILV_TGROUPS"] = str(tg)
        m = S.build_model(*specs, batch=B, max_iter=96, lib_path=lib, graph_iters=gi)
        S.solve_model(m, x0, tol=0.0)
        import time
        t = time.perf_counter(); r = S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
        print(f"tgroups {tg} graph_iters {gi:3d}: device {r[0]['solve_ms'] / 96:.3f} ms/iter, wall {dt * 1e3 / 96:.3f} ms/iter", flush=True)
        m.close()
