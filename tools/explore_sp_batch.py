"""Exploration (not a test): the Spock algorithm proper (SP) in the batched regime on the benchmark tree (nx = nu = 5, N = 13): problems/s
to tol 1e-3 and time per SP iteration against the batch size, next to CP on the same instances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

specs = S.get_server_heat_specs(13, 5, 2)
flat = S.flatten_problem(*specs)
W = 8
bytes_T = W * (4 * flat.nz + 3 * flat.nv + 2 * (flat.n * flat.nx + flat.nn * flat.nu))
bytes_SP = 2 * bytes_T + 26 * W * (flat.nz + flat.nv)
bytes_CP = W * (6 * flat.nz + 5 * flat.nv + 2 * (flat.n * flat.nx + flat.nn * flat.nu))
print(f"nz {flat.nz} nv {flat.nv}: bytes_SP {bytes_SP/1e6:.1f} MB, bytes_CP {bytes_CP/1e6:.1f} MB per instance-iteration (SURVEY 8d)", flush=True)
for B in [int(x) for x in (sys.argv[1:] or ["64", "512", "2048"])]:
    X = 0.1 * (1.0 + 0.0005 * np.arange(B))[:, None] * np.ones((B, 5))
    for name, alg in (("SP", S.SP), ("CP", S.CP)):
        m = S.build_model(*specs, S.SolverOptions(alg, None), batch=B)
        S.solve_model(m, X, tol=1e-3); m.reset()
        t = time.perf_counter(); r = S.solve_model(m, X, tol=1e-3); dt = time.perf_counter() - t
        its = [q["iterations"] for q in r]
        ok = all(q["status"] == 0 for q in r)
        ms_it = r[0]["solve_ms"] / max(its)
        bts = bytes_SP if alg == S.SP else bytes_CP
        print(f"B {B:5d} {name} layout {'ilv' if m.sizes.interleaved else 'inst'}: {dt*1e3:9.1f} ms  {B/dt:8.1f} problems/s  iterations {min(its)}..{max(its)} converged {ok}  "
              f"{ms_it:8.3f} ms per batch iteration -> {B*bts/(ms_it*1e-3)/1e12:6.2f} TB/s on SURVEY 8d bytes", flush=True)
        m.close()
