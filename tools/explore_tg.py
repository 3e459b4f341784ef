"""Exploration script (not a test): ms per CP iteration of the element-major regime for tile-group / chunk settings (env)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
for tg in (1, 2, 4, 8):
    for ch in (8, 16, 32):
        os.environ["SPOCK_AMD_ILV_TGROUPS"] = str(tg); os.environ["SPOCK_AMD_ILV_CHUNK"] = str(ch)
        m = S.build_model(*specs, batch=B, max_iter=96, lib_path=lib)
        S.solve_model(m, x0, tol=0.0)
        r = S.solve_model(m, x0, tol=0.0)
        print(f"tgroups {tg} chunk {ch:2d}: {r[0]['solve_ms'] / 96:.3f} ms/iter  {B * 96 / r[0]['solve_ms'] * 1e3:9.0f} it/s", flush=True)
        m.close()
