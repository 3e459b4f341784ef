"""fp32 variant against fp64 (BASELINE config 5: "fp32 vs fp64 tolerance + HBM-roofline report").  The reference is Float64 only
(src/model.jl:304-307); libspock_amd_f32.so is this build's own single-precision compilation of the same kernels.

For each configuration: CP to tol in both precisions from the same x0 -- iterations to tolerance, ms, and the distance of the
fp32 iterate from the fp64 one (relative to max(1, |z64|_inf)); plus the attainable accuracy: the residual level at which the
fp32 run stalls (tol = 0, fixed iteration count).

    python tools/fp32_study.py [full] > profiles/r02/fp32_study.json
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

full = len(sys.argv) > 1 and sys.argv[1] == "full"
cases = [("cfg5 random linear nx=64 nu=16 N=9", S.get_random_linear_specs(9, 64, 16), "auto"),
         ("cfg1 server_heat nx=5 N=10", S.get_server_heat_specs(10, 5, 2), None),
         ("cfg4 server_heat nx=20 N=10", S.get_server_heat_specs(10, 20, 2), None)]
if full:
    cases.insert(1, ("cfg5 random linear nx=64 nu=16 N=15 (full size)", S.get_random_linear_specs(15, 64, 16), "auto"))
rows = []
for name, specs, L in cases:
    flat = S.flatten_problem(*specs)
    x0 = np.full(flat.nx, 0.1)
    row = {"config": name, "nz": flat.nz, "nv": flat.nv}
    Z = {}
    for dt in ("f64", "f32"):
        for tol in (1e-3, 1e-4):
            m = S.build_model(*specs, dtype=dt, L_norm=L, max_iter=5000)
            S.solve_model(m, x0, tol=tol); m.reset()
            r = S.solve_model(m, x0, tol=tol)
            row[f"{dt}_tol{tol:g}"] = {"iterations": r["iterations"], "status": r["status"], "ms": round(r["solve_ms"], 3),
                                       "us_per_iteration": round(r["solve_ms"] / max(1, r["iterations"]) * 1e3, 2)}
            Z[(dt, tol)] = (m.solver_state.z.copy(), m.solver_state.v.copy())
            m.close()
        m = S.build_model(*specs, dtype=dt, L_norm=L, max_iter=3000)      # residual floor: tol = 0, 3000 iterations
        r = S.solve_model(m, x0, tol=0.0)
        h = m.history()
        row[f"{dt}_floor"] = {"xi1_at_3000": float(h["xi1"][-1]), "xi2_at_3000": float(h["xi2"][-1]), "xi1_min": float(np.min(h["xi1"])), "xi2_min": float(np.min(h["xi2"]))}
        m.close()
    for tol in (1e-3, 1e-4):
        z64, v64 = Z[("f64", tol)]; z32, v32 = Z[("f32", tol)]
        row[f"rel_dist_z_tol{tol:g}"] = float(np.max(np.abs(z32 - z64)) / max(1.0, np.max(np.abs(z64))))
        row[f"rel_dist_v_tol{tol:g}"] = float(np.max(np.abs(v32 - v64)) / max(1.0, np.max(np.abs(v64))))
    rows.append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps({"study": "fp32 (libspock_amd_f32.so) vs fp64 (libspock_amd.so), CP, same x0 = 0.1, SILENT termination rule",
                  "note": "the reference has no fp32 path (src/model.jl:304-307): distances are to this build's fp64 iterates", "rows": rows}, indent=1))
