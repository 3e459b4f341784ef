"""Soak (not a test, run on the GPU box): the same solves repeated many times in one process must give bitwise identical
iterates and iteration counts every time (any data race in the kernels or the device-side control logic would show up as
run-to-run differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for desc, specs, alg, B, kw in (("cfg2 nu=5 CP B=1", S.get_server_heat_specs(13, 5, 2), S.CP, 1, {}),
                                ("cfg2 nu=5 SP B=1", S.get_server_heat_specs(13, 5, 2), S.SP, 1, {}),
                                ("N=10 nu=2 CP B=40", S.get_server_heat_specs(10, 5, 2, nu=2), S.CP, 40, {"max_iter": 120}),
                                ("N=10 nu=5 SP B=6", S.get_server_heat_specs(10, 5, 2), S.SP, 6, {}),
                                ("cfg4 nx=20 SP B=1", S.get_server_heat_specs(10, 20, 2), S.SP, 1, {}),
                                ("N=7 whole-tree CP B=3", S.get_server_heat_specs(7, 5, 2), S.CP, 3, {}),
                                # round 3: flat node-local kernels + matrix-core sweep kernels (wave-local teams, shuffle sums, staged matrices)
                                ("cfg4 nx=20 CP B=9", S.get_server_heat_specs(10, 20, 2), S.CP, 9, {}),
                                ("cfg4 nx=20 SP B=9", S.get_server_heat_specs(10, 20, 2), S.SP, 9, {}),
                                ("cfg3 nx=20 d=3 N=7 CP B=2", S.get_server_heat_specs(7, 20, 3, p_ref=[0.2, 0.3, 0.5]), S.CP, 2, {"L_norm": "auto"}),
                                ("cfg5 nx=64 nu=16 N=11 CP B=2", S.get_random_linear_specs(11, 64, 16), S.CP, 2, {"L_norm": "auto", "max_iter": 150}),
                                ("scaling.jl nx=50 N=9 SP B=1", S.get_server_heat_specs(9, 50, 2), S.SP, 1, {})):
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, S.SolverOptions(alg, None), batch=B, **kw)
    x0 = np.full((B, flat.nx), 0.1) * (1 + 0.03 * np.arange(B))[:, None]
    ref = None
    n = reps if B * flat.n < 50000 else max(20, reps // 5)
    for k in range(n):
        m.reset()
        r = S.solve_model(m, x0 if B > 1 else x0[0], tol=1e-3)
        its = [q["iterations"] for q in (r if isinstance(r, list) else [r])]
        sig = (tuple(its), np.asarray(m.solver_state.z).tobytes(), np.asarray(m.solver_state.v).tobytes())
        if ref is None:
            ref = sig
        elif sig != ref:
            bad += 1; print("MISMATCH", desc, "repetition", k, its, flush=True)
    print(f"{desc}: {n} repetitions, iterations {list(ref[0])[:6]}", flush=True)
    m.close()
# closed loops with the steps chained on the device: identical trajectories every time
specs = S.get_server_heat_specs(8, 20, 2)
B, steps = 12, 10
w = np.stack([np.random.default_rng(seed).integers(0, 2, size=steps) for seed in range(B)]).astype(np.int32)
x0 = 0.1 * (1.0 + 0.05 * np.arange(B))[:, None] * np.ones((B, 20))
m = S.build_model(*specs, S.SolverOptions(S.SP, None), batch=B)
ref = None
for k in range(max(10, reps // 10)):
    m.reset()
    xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
    sig = (xt.tobytes(), ut.tobytes(), tuple(r["iterations"] for rb in res for r in rb))
    if ref is None:
        ref = sig
    elif sig != ref:
        bad += 1; print("MISMATCH closed loop, repetition", k, flush=True)
print(f"closed loop nx=20 N=8 SP B={B} x {steps} steps (chained on the device): {max(10, reps // 10)} repetitions", flush=True)
m.close()
print("soak:", "FAILED" if bad else "ok", bad)
