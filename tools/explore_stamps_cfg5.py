"""Exploration (not a test): in-kernel clock stamps for BASELINE config 4 (nx = nu = 20, N = 10, lane-team kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from spock_jl_amd import _lib
specs = S.get_random_linear_specs(15, 64, 16)
m = S.build_model(*specs, max_iter=100, L_norm="auto")
S.solve_model(m, 0.1 * np.cos(np.arange(64)), tol=0.0)
os.environ["SPOCK_AMD_DEBUG_FLAGS"] = "512"
prof = m.profile_kernels(40)
out = np.zeros((3, 64, 2), dtype=np.uint64)
_lib.check(m._lib, m._lib.spock_debug_stamps(m._h, out.ctypes.data))
print({k: round(v[0] * 1e3, 1) for k, v in prof.items() if v[1]}, "split", m.sizes.split_stage, "fused", m.sizes.fused_iteration)
for ki, name in enumerate(("sub_a", "crown", "sub_b")):
    st = out[ki].astype(np.int64)
    n = int(np.count_nonzero(st[:40, 1]))
    wall = np.diff(st[:n, 1]) * 10.0
    print(f"  {name}: {n} stamps, total {(st[n-1,1]-st[0,1])*10/1e3:.2f} us; ns per segment:", [int(x) for x in wall])
m.close()
