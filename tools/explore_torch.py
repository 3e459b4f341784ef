"""Exploration script (not a test): does an initialised torch.cuda context change the overlap of the tile-group streams?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode == "init":
        torch.cuda.init(); torch.cuda.synchronize()
    if mode == "tensor":
        t = torch.zeros(16, device="cuda"); torch.cuda.synchronize()
import spock_jl_amd as S
B = 2048
specs = S.get_server_heat_specs(13, 5, 2, nu=2)
x0 = np.full((B, 5), 0.1) * (1 + 0.001 * np.arange(B))[:, None]
for tg in (1, 4):
    os.environ["SPOCK_AMD_ILV_TGROUPS"] = str(tg)
    m = S.build_model(*specs, S.SolverOptions(S.CP, None), batch=B, max_iter=96, graph_iters=96)
    S.solve_model(m, x0, tol=0.0)
    for rep in range(3):
        t = time.perf_counter(); r = S.solve_model(m, x0, tol=0.0); dt = time.perf_counter() - t
        print(f"{mode}: tgroups {tg}: device {r[0]['solve_ms'] / 96:.3f} ms/iter, wall {dt * 1e3 / 96:.3f} ms/iter", flush=True)
    m.close()
print({k: v for k, v in os.environ.items() if any(s in k for s in ("HIP", "HSA", "GPU_", "ROC", "AMD"))})
