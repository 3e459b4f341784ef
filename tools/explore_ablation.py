"""Exploration (not a test): where does the crown / subtree kernel time go?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S

specs = S.get_server_heat_specs(13, 5, 2, nu=2)
for B in (1, 64):
    m = S.build_model(*specs, batch=B, split_stage=6, max_iter=100)
    x0 = np.full((B, 5), 0.1)
    S.solve_model(m, x0, tol=0.0)
    for name, fl in (("all", 0), ("no_bwd", 128), ("no_fwd", 256), ("no_sweeps", 2), ("no_sweeps_no_ctl", 2 | 32),
                     ("nothing(noA,noB,nosweep)", 2 | 4 | 64), ("only_A", 2 | 4), ("only_B", 2 | 64)):
        os.environ["SPOCK_AMD_DEBUG_FLAGS"] = str(fl)
        prof = m.profile_kernels(60)
        print(f"B={B} {name:28s}", " ".join(f"{k}={v[0]*1e3:7.1f}us" for k, v in prof.items() if v[1]), flush=True)
    m.close()
