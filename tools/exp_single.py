"""Exploration (not a test): single-instance CP / SP iteration time on BASELINE config 2 under the environment's kernel choices."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 2
specs = S.get_server_heat_specs(13, 5, 2, nu=nu)
x0 = np.full(5, 0.1)
for alg, iters in ((S.CP, 1200), (S.SP, 300)):
    m = S.build_model(*specs, S.SolverOptions(alg, None), max_iter=iters)
    S.solve_model(m, x0, tol=0.0)
    m.reset()
    t = time.perf_counter(); r = S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
    seq = [n for n, _, _ in m.profile_sequence(1)] if alg == S.CP else []
    print("CP" if alg == S.CP else "SP", "split", m.sizes.split_stage, "fused", m.sizes.fused_iteration, "iters", r["iterations"], "us/iter %.2f" % (dt / r["iterations"] * 1e6), seq, flush=True)
    m.close()
