import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
specs = S.get_server_heat_specs(10, 20, 2)
xa = 0.1 * (1.0 + 0.01 * np.arange(total))[:, None] * np.ones((total, 20))
wa = np.stack([np.random.default_rng(seed).integers(0, 2, size=20) for seed in range(total)]).astype(np.int32)
m = S.build_model(*specs, S.SolverOptions(S.SP, None), batch=total)
S.mpc_simulation(m, xa, wa, tol=1e-3)
m.close()
