"""Context row against the reference's ONLY published timing curve: the Spock branch of examples/server_heat/scaling.jl
(nx = nu = 50, d = 2, N = 3..15, TOL = 1e-3, SP, cold start, one solve per N; scaling.jl:9-17,58), whose measured wall times
sit in examples/server_heat/output/scaling.tikz:29-41 ("reference: Julia, unknown hardware" -- context, not a target).

    python tools/scaling_curve.py [N_max] > profiles/r02/scaling_curve.json
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

REFERENCE_S = {3: 0.02994837, 4: 0.060908716, 5: 0.120554163, 6: 0.28526187, 7: 0.557456879, 8: 1.540880604, 9: 2.428256747,
               10: 4.800994611, 11: 14.616935775, 12: 22.880396134, 13: 45.661248555, 14: 78.266806084, 15: 151.511085064}
N_max = int(sys.argv[1]) if len(sys.argv) > 1 else 15
nx, d, TOL = 50, 2, 1e-3
rows = []
for N in range(3, N_max + 1):
    specs = S.get_server_heat_specs(N, nx, d)
    flat = S.flatten_problem(*specs)
    try:
        t = time.perf_counter(); m = S.build_model(*specs, S.SolverOptions(S.SP, None)); tb = time.perf_counter() - t
    except S.SpockError as e:
        rows.append({"N": N, "error": str(e)}); continue
    x0 = np.full(nx, 0.1)
    S.solve_model(m, x0, tol=TOL); m.reset()          # warm-up of the code path (graph capture), then a cold solve
    t = time.perf_counter(); r = S.solve_model(m, x0, tol=TOL); wall = time.perf_counter() - t
    rows.append({"N": N, "nodes": flat.n, "nz_plus_nv": flat.nz + flat.nv, "sp_iterations": r["iterations"], "status": r["status"],
                 "gpu_solve_ms": round(r["solve_ms"], 3), "gpu_wall_ms": round(wall * 1e3, 3), "build_ms": round(tb * 1e3, 1),
                 "reference_julia_s": REFERENCE_S[N], "ratio_reference_over_gpu": round(REFERENCE_S[N] / wall, 1)})
    print(rows[-1], file=sys.stderr, flush=True)
    m.close()
print(json.dumps({"workload": "examples/server_heat/scaling.jl, Spock branch: nx = nu = 50, d = 2, SP, tol 1e-3, cold start",
                  "reference": "examples/server_heat/output/scaling.tikz:29-41 -- Julia, unknown hardware, one run (context, not a target)",
                  "rows": rows}, indent=1))
