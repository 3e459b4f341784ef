"""Exploration (not a test): the flat node-local kernels (spock_flat_local.hpp) against the subtree kernels' own node-local phases on
BASELINE configs 3, 4, 5 -- time per CP iteration, per-launch breakdown, SP iteration time, batches of config 4, parity spot check.
    python tools/explore_flat.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def run(name, specs, L="auto", iters=60, batch=1, alg=None, check=True, **kw):
    flat = S.flatten_problem(*specs)
    alg = alg or S.CP
    m = S.build_model(*specs, S.SolverOptions(alg, None), max_iter=iters, L_norm=L, batch=batch, graph_iters=(iters if alg == S.CP else 0), **kw)
    x0 = 0.1 * (1 + 0.01 * np.arange(batch))[:, None] * np.cos(np.arange(flat.nx))[None, :]
    S.solve_model(m, x0, tol=0.0); m.reset()
    r = S.solve_model(m, x0, tol=0.0)
    r0 = r[0] if isinstance(r, list) else r
    err = float("nan")
    if check:
        o = O.Oracle(flat, O.CP if alg == S.CP else O.SP); o.L_norm = m.solver_state.L_norm
        o.solve(x0[0], tol=0.0, max_iter=iters)
        z = m.solver_state.z.reshape(batch, -1)[0]
        err = float(np.max(np.abs(z - o.z)) / max(1.0, np.max(np.abs(o.z))))
    seq = ""
    if alg == S.CP:
        seq = " ".join(f"{n}:{ms*1e3:.1f}" for n, _, ms in m.profile_sequence(12))
    print(f"{name:34s} batch {batch:3d} {'CP' if alg == S.CP else 'SP'} {r0['solve_ms']/r0['iterations']*1e3:9.1f} us/iter  z err {err:.1e}  [{seq}]", flush=True)
    m.close()


cfgs = [("cfg4 N=10 nx=20 d=2", S.get_server_heat_specs(10, 20, 2), 3.28),
        ("cfg3 N=9 nx=20 d=3", S.get_server_heat_specs(9, 20, 3, p_ref=[0.2, 0.3, 0.5]), "auto"),
        ("cfg5 N=15 nx=64 nu=16", S.get_random_linear_specs(15, 64, 16), "auto")]
for env in ({}, {"SPOCK_AMD_FLAT_FUSE": "0"}, {"SPOCK_AMD_FLAT_LOCAL": "0"}):
    for k in ("SPOCK_AMD_FLAT_FUSE", "SPOCK_AMD_FLAT_LOCAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    print("---- env", env, flush=True)
    for name, specs, L in cfgs:
        big = "cfg5" in name
        run(name, specs, L, iters=24 if big else 60, check=not (quick and big))
        if not quick or "cfg4" in name:
            run(name, specs, L, iters=12, alg=S.SP, check=False)
    run("cfg4 N=10 nx=20 d=2", cfgs[0][1], 3.28, iters=60, batch=8, check=False)
    run("cfg4 N=10 nx=20 d=2", cfgs[0][1], 3.28, iters=60, batch=64, check=False)
    run("cfg4 N=10 nx=20 d=2", cfgs[0][1], 3.28, iters=12, batch=64, alg=S.SP, check=False)
