"""Exploration (not a test): BASELINE config 5 (nx = 64, nu = 16) -- MFMA sweep levels against the scalar lane-team levels:
parity of T against the oracle, time per CP iteration, per-kernel times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

kw = {"lib_path": os.environ["LIB"]} if os.environ.get("LIB") else {}
for N in [int(a) for a in sys.argv[1:]] or [9, 15]:
    specs = S.get_random_linear_specs(N, 64, 16)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, max_iter=24, L_norm="auto", **kw)
    x0 = np.full(flat.nx, 0.1)
    o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
    rng = np.random.default_rng(0); z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv)
    zo, vo = m.T(z, v, 0.3, 0.3, x0); zr, vr = o.T(z, v, 0.3, 0.3, x0)
    err = max(np.max(np.abs(zo - zr)) / max(1, np.max(np.abs(zr))), np.max(np.abs(vo - vr)) / max(1, np.max(np.abs(vr))))
    S.solve_model(m, x0, tol=0.0); m.reset()
    r = S.solve_model(m, x0, tol=0.0)
    ro = o.solve(x0, tol=0.0, max_iter=24)
    zerr = np.max(np.abs(m.solver_state.z - o.z)) / max(1, np.max(np.abs(o.z)))
    prof = m.profile_kernels(12)
    print(f"N={N} split={m.sizes.split_stage}: T err {err:.1e}  24 its z err {zerr:.1e}  {r['solve_ms']/r['iterations']*1e3:8.1f} us/it  kernels(ms) "
          + " ".join(f"{k}={v_[0]:.4f}" for k, v_ in prof.items() if v_[1]), flush=True)
    m.close()
