"""Exploration (not a test): BASELINE config 4's closed-loop legs -- 8 and 64 simulations x 20 steps, SP, tol 1e-3 -- with the steps
chained on the device against the host loop of solve calls; graph sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

specs = S.get_server_heat_specs(10, 20, 2)
steps = 20
for env in ({}, {"SPOCK_AMD_MPC_GRAPH_ITERS": "8"}, {"SPOCK_AMD_MPC_GRAPH_ITERS": "32"}, {"SPOCK_AMD_MPC_HOST": "1"}, {"SPOCK_AMD_MPC_HOST": "1", "SPOCK_AMD_FLAT_LOCAL": "0"}):
    for k in ("SPOCK_AMD_MPC_GRAPH_ITERS", "SPOCK_AMD_MPC_HOST", "SPOCK_AMD_FLAT_LOCAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for total in (8, 64, 256):
        xa = 0.1 * (1.0 + 0.01 * np.arange(total))[:, None] * np.ones((total, 20))
        wa = np.stack([np.random.default_rng(seed).integers(0, 2, size=steps) for seed in range(total)]).astype(np.int32)
        m = S.build_model(*specs, S.SolverOptions(S.SP, None), batch=total)
        S.mpc_simulation(m, xa, wa, tol=1e-3); m.reset()
        t = time.perf_counter(); xt, ut, res = S.mpc_simulation(m, xa, wa, tol=1e-3); dt = time.perf_counter() - t
        its = sum(r["iterations"] for rb in res for r in rb)
        per_sim = [sum(r["iterations"] for r in rb) for rb in res]
        print(f"{str(env):60s} sims {total:4d}: {dt*1e3:8.1f} ms  {total/dt:7.1f} sims/s  SP iterations total {its}, per simulation min {min(per_sim)} max {max(per_sim)}", flush=True)
        m.close()
