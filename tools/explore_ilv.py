"""Exploration (not a test): element-major vs instance-major batched CP at BASELINE config 2 -- time per iteration, kernel
times, parity of one instance against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

B = int(os.environ.get("B", "1024")); M = int(os.environ.get("M", "96")); N = int(os.environ.get("N", "13")); NU = int(os.environ.get("NU", "2"))
layouts = os.environ.get("LAYOUTS", "interleaved,instance").split(",")
specs = S.get_server_heat_specs(N, 5, 2, nu=NU)
flat = S.flatten_problem(*specs)
x0 = 0.1 * (1.0 + 0.001 * np.arange(B))[:, None] * np.ones((B, 5))
o = O.Oracle(flat, O.CP); o.solve(x0[B // 3], tol=0.0, max_iter=M)
for lay in layouts:
    kw = {}
    if os.environ.get("LIB"): kw["lib_path"] = os.environ["LIB"]
    if os.environ.get("SPLIT"): kw["split_stage"] = int(os.environ["SPLIT"])
    m = S.build_model(*specs, batch=B, layout=lay, max_iter=M, graph_iters=M, **kw)
    S.solve_model(m, x0, tol=0.0); m.reset()
    t = time.perf_counter(); r = S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
    z = m.solver_state.z.reshape(B, -1)[B // 3]
    err = np.max(np.abs(z - o.z)) / max(1, np.max(np.abs(o.z)))
    prof = m.profile_kernels(24)
    print(f"{lay:12s} ilv={m.sizes.interleaved} split={m.sizes.split_stage} nsub={m.sizes.n_subtrees}: {dt*1e3:8.2f} ms / {M} its = "
          f"{dt/M*1e6:8.1f} us/it  {B*M/dt:10.0f} it/s   err vs oracle {err:.1e}   kernels(ms) "
          + " ".join(f"{k}={v[0]:.4f}" for k, v in prof.items() if v[1]), flush=True)
    m.close()
