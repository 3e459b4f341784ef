"""Soak (not a test; GPU box): the element-major kernels (crown + level-streaming role waves, double-buffered LDS exchange rows)
repeated many times on the same inputs -- every repeat must give bitwise the same iterate.  A race between the role waves of a
workgroup, or between launches, would show as a difference.     python tests/tools/soak_ilv.py [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for (N, nu, B, iters) in [(13, 2, 2048, 24), (10, 5, 130, 60), (9, 2, 700, 48)]:
    specs = S.get_server_heat_specs(N, 5, 2, nu=nu)
    x0 = 0.1 * (1.0 + 0.001 * np.arange(B))[:, None] * np.ones((B, 5))
    m = S.build_model(*specs, batch=B, layout="interleaved", max_iter=iters)
    S.solve_model(m, x0, tol=0.0)
    ref_z = m.solver_state.z.copy(); ref_v = m.solver_state.v.copy()
    n = reps if B < 1000 else max(5, reps // 10)
    for k in range(n):
        m.reset()
        S.solve_model(m, x0, tol=0.0)
        if not (np.array_equal(m.solver_state.z, ref_z) and np.array_equal(m.solver_state.v, ref_v)):
            bad += 1
            print(f"MISMATCH N={N} nu={nu} B={B} repeat {k}: max |dz| {np.max(np.abs(m.solver_state.z - ref_z)):.3e}", flush=True)
    print(f"N={N} nu={nu} B={B}: {n} repeats of {iters} iterations bitwise identical" if not bad else f"N={N} nu={nu} B={B}: {bad} mismatches", flush=True)
    m.close()
print("soak_ilv:", "OK" if not bad else f"{bad} MISMATCHES")
