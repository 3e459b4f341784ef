"""Exploration script (not a test): per-launch durations of one CP iteration in the element-major regime (spock_profile_sequence).
    python tools/explore_seq.py [lib] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
m = S.build_model(*specs, batch=B, max_iter=48, lib_path=lib)
x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
S.solve_model(m, x0, tol=0.0)
r = S.solve_model(m, x0, tol=0.0)
print("ms/iter (graph, tile groups):", r[0]["solve_ms"] / 48)
tot = 0.
for name, lv, ms in m.profile_sequence(12):
    print(f"  {name:8s} {lv:3d} {ms*1e3:9.1f} us"); tot += ms
print("  sum", tot * 1e3, "us")
m.close()
