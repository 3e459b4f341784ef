"""What do the defects of the reference's run_sp! / anderson! cost?  SP as the reference has it (sp_variant 0) against the
corrected variant (sp_variant 1: least-squares Anderson, a line search that shortens the step), on the GPU: iterations and time
to tolerance, backtrack counts.  Writes profiles/<round>/sp_variant_study.json.     python tools/sp_variant_study.py [round]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
cases = [("cfg1 server_heat N=7 nx=5 nu=5", S.get_server_heat_specs(7, 5, 2, nu=5), 5, 0.1),
         ("cfg2-like N=13 nx=5 nu=5", S.get_server_heat_specs(13, 5, 2, nu=5), 5, 0.1),
         ("cfg4 server_heat N=10 nx=nu=20", S.get_server_heat_specs(10, 20, 2), 20, 0.1)]
out = []
for name, specs, nx, x0s in cases:
    for tol in (1e-3, 1e-4):
        row = {"problem": name, "tol": tol}
        for var in (0, 1):
            m = S.build_model(*specs, S.SolverOptions(S.SP, None), max_iter=3000, sp_variant=var)
            x0 = np.full(nx, x0s)
            S.solve_model(m, x0, tol=tol); m.reset()
            r = S.solve_model(m, x0, tol=tol, verbose=S.LOG)
            h = m.history(0)
            row[f"variant{var}"] = {"iterations": r["iterations"], "status": r["status"], "ms": round(r["solve_ms"], 3),
                                    "backtrack_histogram": np.bincount(h["backtrack"].astype(int)).tolist()}
            m.close()
        out.append(row)
        print(row, flush=True)
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "sp_variant_study.json"), "w"), indent=1)
