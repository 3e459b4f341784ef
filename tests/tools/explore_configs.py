"""Exploration (not a test): every BASELINE.json configuration once -- build, a few CP iterations, time per iteration,
and a spot check of the operator T against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

def run(name, specs, L=None, iters=60, check=True):
    flat = S.flatten_problem(*specs)
    t = time.time(); m = S.build_model(*specs, max_iter=iters, L_norm=("auto" if L is None else L)); tb = time.time() - t
    x0 = np.full(flat.nx, 0.1)
    S.solve_model(m, x0, tol=0.0); m.reset()
    r = S.solve_model(m, x0, tol=0.0)
    err = float("nan")
    if check:
        o = O.Oracle(flat, O.CP)
        rng = np.random.default_rng(0); z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv)
        zo, vo = m.T(z, v, 0.3, 0.3, x0); zr, vr = o.T(z, v, 0.3, 0.3, x0)
        err = max(np.max(np.abs(zo - zr)) / max(1, np.max(np.abs(zr))), np.max(np.abs(vo - vr)) / max(1, np.max(np.abs(vr))))
    print(f"{name}: n={flat.n} nz={flat.nz} nv={flat.nv} split={m.sizes.split_stage} fused={m.sizes.fused_iteration} build {tb:.2f}s  {r['solve_ms']/r['iterations']*1e3:9.1f} us/iter  T err {err:.1e}", flush=True)
    m.close()

run("cfg1 N=10 nx=5 d=2", S.get_server_heat_specs(10, 5, 2))
run("cfg2 N=13 nx=5 nu=2", S.get_server_heat_specs(13, 5, 2, nu=2))
run("cfg3 N=9 nx=20 d=3", S.get_server_heat_specs(9, 20, 3, p_ref=[0.2, 0.3, 0.5]))
run("cfg4 N=10 nx=20 d=2", S.get_server_heat_specs(10, 20, 2))
run("cfg5-small N=9 nx=64 nu=16", S.get_random_linear_specs(9, 64, 16))
if os.environ.get("BIG", "0") == "1":
    run("cfg5 N=15 nx=64 nu=16", S.get_random_linear_specs(15, 64, 16), iters=20, check=True)
