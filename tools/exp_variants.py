"""Exploration script (not a test): the CP iteration at batch 256 / 1 for the default library and every exp/libspock_*.so."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import spock_jl_amd as S

def run(lib, B, M=300, nu=2):
    specs = S.get_server_heat_specs(13, 5, 2, nu=nu)
    m = S.build_model(*specs, batch=B, max_iter=M, lib_path=lib)
    x0 = np.full((B, 5), 0.1) * (1 + 0.01 * np.arange(B))[:, None]
    S.solve_model(m, x0, tol=0.0)
    r = S.solve_model(m, x0, tol=0.0)
    ms = (r if B == 1 else r[0])["solve_ms"]
    prof = m.profile_kernels(30)
    print(f"  B={B:4d} nu={nu}: {ms/M*1e3:8.2f} us/iter {B*M/ms*1e3:10.0f} it/s | " + " ".join(f"{k}={v[0]*1e3:.1f}" for k, v in prof.items() if v[1]), flush=True)
    m.close()

libs = [None] + sorted(glob.glob(os.path.join(ROOT, "exp", "libspock_*.so")))
only = sys.argv[1:]
for lib in libs:
    name = "default" if lib is None else os.path.basename(lib)
    if only and not any(o in name for o in only): continue
    print(name, flush=True)
    for nu in (2, 5):
        for B in (256, 1):
            run(lib, B, nu=nu)
