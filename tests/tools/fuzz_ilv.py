"""Randomised parity sweep of the element-major regime (not a test; GPU box, or the emulation library through SPOCK_FUZZ_LIB): the
compile-time shapes, diagonal / shared dense cost roots, split stages, batch sizes around the tile boundaries, chunk sizes of the
level-streaming kernels, SILENT / LOG termination, level-streaming and bottom-group kernels; CP against the oracle on three
instances per case.     python tests/tools/fuzz_ilv.py [n_cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = os.environ.get("SPOCK_FUZZ_LIB")


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


fails = 0; worst = 0.0
for case in range(n_cases):
    shape = int(rng.integers(0, 4))
    if shape == 0:
        N = int(rng.integers(3, 11)); specs = S.get_server_heat_specs(N, 5, 2, nu=2)
    elif shape == 1:
        N = int(rng.integers(3, 10)); specs = S.get_server_heat_specs(N, 5, 2, nu=5)
    elif shape == 2:
        N = int(rng.integers(3, 11)); specs = S.get_random_linear_specs(N, 2, 1, seed=int(rng.integers(1 << 30)))
    else:
        N = int(rng.integers(3, 7)); specs = S.get_random_linear_specs(N, 3, 2, d=3, seed=int(rng.integers(1 << 30)))
    tree, cost, dyn, rms, con = specs
    nx, nu = dyn.A[0].shape[0], dyn.B[0].shape[1]
    kind = "diag"
    if rng.random() < 0.3:
        def spd(n):
            a = rng.standard_normal((n, n)); return 0.05 * (a @ a.T) / n + np.eye(n) * 0.1
        cost = S.Cost([spd(nx)], [spd(nu) + np.eye(nu)], [spd(nx)]); kind = "dense_shared"
    specs = (tree, cost, dyn, rms, con)
    split = int(rng.integers(1, N - 1)) if rng.random() < 0.7 else 0
    B = int(rng.choice([1, 3, 63, 64, 65, 70, 128, 130, 200]))
    stream = "0" if rng.random() < 0.25 else "1"
    chunk = int(rng.choice([1, 2, 3, 5, 16]))
    log = rng.random() < 0.5
    iters = int(rng.integers(7, 50))
    desc = f"shape={nx},{nu},d={tree.d} N={N} cost={kind} split={split} B={B} stream={stream} chunk={chunk} log={log} iters={iters}"
    os.environ["SPOCK_AMD_ILV_STREAM"] = stream; os.environ["SPOCK_AMD_ILV_CHUNK"] = str(chunk)
    try:
        flat = S.flatten_problem(*specs)
        if stream == "0" and split and (2 ** 0) and (tree.d ** (N - split) - 1) // (tree.d - 1) * nx * 64 * 8 > 56 * 1024:
            split = 0      # the bottom-group kernels keep a subtree's sweep slots in LDS
        kw = {"lib_path": lib} if lib else {}
        m = S.build_model(*specs, batch=B, layout="interleaved", max_iter=iters, split_stage=split, **kw)
        x0 = 0.1 * (1 + 0.37 * rng.standard_normal((B, 1))) * np.cos(np.arange(nx))[None, :]
        rs = S.solve_model(m, x0, tol=1e-5, verbose=S.LOG if log else S.SILENT)
        rs = rs if isinstance(rs, list) else [rs]
        Z = m.solver_state.z.reshape(B, -1); V = m.solver_state.v.reshape(B, -1)
        for b in sorted(set([0, B // 2, B - 1])):
            o = O.Oracle(flat, O.CP); ro = o.solve(x0[b], tol=1e-5, max_iter=iters, verbose=O.LOG if log else O.SILENT)
            e = max(relerr(Z[b], o.z), relerr(V[b], o.v)); worst = max(worst, e)
            if (rs[b]["iterations"], rs[b]["status"]) != (ro["iterations"], ro["status"]) or not e <= 1e-9:
                fails += 1; print("FAIL", desc, "instance", b, rs[b]["iterations"], ro["iterations"], rs[b]["status"], ro["status"], e, flush=True)
        m.close()
    except S.SpockError as ex:
        print("refused", desc, str(ex)[:80], flush=True)
print(f"fuzz_ilv: {n_cases} cases, {fails} failures, worst relative error {worst:.2e}")
