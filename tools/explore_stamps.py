"""Exploration (not a test): in-kernel clock stamps of the three kernels of one CP iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from spock_jl_amd import _lib

specs = S.get_server_heat_specs(13, 5, 2, nu=2)
for B in (1, 256):
    m = S.build_model(*specs, batch=B, split_stage=6, max_iter=100)
    S.solve_model(m, np.full((B, 5), 0.1), tol=0.0)
    os.environ["SPOCK_AMD_DEBUG_FLAGS"] = "512"
    prof = m.profile_kernels(40)
    out = np.zeros((3, 64, 2), dtype=np.uint64)
    _lib.check(m._lib, m._lib.spock_debug_stamps(m._h, out.ctypes.data))
    print(f"B={B}", {k: round(v[0] * 1e3, 1) for k, v in prof.items() if v[1]})
    for ki, name in enumerate(("sub_a", "crown", "sub_b")):
        st = out[ki].astype(np.int64)
        n = int(np.count_nonzero(st[:40, 1]))
        cyc = np.diff(st[:n, 0]); wall = np.diff(st[:n, 1]) * 10.0   # ns
        tot_c = st[n - 1, 0] - st[0, 0]; tot_w = (st[n - 1, 1] - st[0, 1]) * 10.0
        print(f"  {name}: {n} stamps, total {tot_w/1e3:.2f} us, {tot_c} cycles -> {tot_c/max(tot_w,1)*1e3:.0f} MHz")
        print("    ns per segment:", [int(x) for x in wall])
        t0 = st[0, 1]
        ws = [(int(st[40 + w, 1] - t0) * 10, int(st[50 + w, 1] - t0) * 10) for w in range(8) if st[40 + w, 1]]
        print("    per-wave (start, end) ns relative to stamp 0:", ws)
    m.close()
