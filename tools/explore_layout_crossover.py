"""Exploration script (not a test): CP iterations/s of BASELINE config 2 against the batch size for both layouts of the batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
for B in (64, 128, 256, 384, 512, 768, 1024, 2048):
    x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
    row = []
    for lay in ("instance", "interleaved"):
        m = S.build_model(*specs, batch=B, max_iter=96, layout=lay)
        S.solve_model(m, x0, tol=0.0)
        r = S.solve_model(m, x0, tol=0.0)
        row.append(B * 96 / r[0]["solve_ms"] * 1e3)
        m.close()
    print(f"B={B:5d}: instance-major {row[0]:9.0f} it/s   element-major {row[1]:9.0f} it/s   ratio {row[1] / row[0]:.2f}", flush=True)
