"""Exploration (not a test): where does the operator T at BASELINE config 5 full size deviate from the oracle?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spock_jl_amd as S
from oracle import oracle as O


def where(name, a, ref, sl, per):
    for k, s in sl.items():
        d = np.abs(a[s] - ref[s]); bad = ~(d <= 1e-9 * max(1.0, np.max(np.abs(ref[s]))))
        if bad.any():
            idx = np.nonzero(bad)[0]
            p = per.get(k, 1)
            print(f"  {name}.{k}: {bad.sum()} bad of {bad.size} (nan {np.isnan(a[s]).sum()}), first elem {idx[0]} (node {idx[0] // p}), last elem {idx[-1]} (node {idx[-1] // p})", flush=True)


for N in [int(a) for a in sys.argv[1:]] or [12, 15]:
    specs = S.get_random_linear_specs(N, 64, 16)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, max_iter=20, L_norm="auto")
    x0 = np.full(flat.nx, 0.1)
    o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
    rng = np.random.default_rng(0); z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv)
    zsl = flat.z_slices(); zper = {"x": flat.nx, "u": flat.nu, "s": 1, "tau": 1, "y": flat.ny}
    off = flat.v_offsets(); keys = ["v1", "v2", "v3", "v4", "v5", "v6", "v7", "v11", "v12", "v13", "v14"]
    starts = [0] + [off[k] for k in keys[1:]] + [flat.nv]
    vsl = {k: slice(starts[i], starts[i + 1]) for i, k in enumerate(keys)}
    vper = {"v1": flat.ny, "v3": flat.nx, "v4": flat.nu, "v7": flat.nx + flat.nu, "v11": flat.nx, "v14": flat.nx}
    print(f"N={N} n={flat.n} split={m.sizes.split_stage} nsub={m.sizes.n_subtrees}", flush=True)
    zr, vr = o.T(z, v, 0.3, 0.3, x0)
    for rep in range(2):
        zo, vo = m.T(z, v, 0.3, 0.3, x0)
        print(f" T rep {rep}: z err {np.nanmax(np.abs(zo - zr)):.2e} nan {np.isnan(zo).sum()}  v err {np.nanmax(np.abs(vo - vr)):.2e} nan {np.isnan(vo).sum()}", flush=True)
        where("zbar", zo, zr, zsl, zper); where("vbar", vo, vr, vsl, vper)
    pf = m.prox_f(z, 0.3, x0); pr = o.prox_f(z, 0.3, x0)
    print(f" prox_f err {np.nanmax(np.abs(pf - pr)):.2e} nan {np.isnan(pf).sum()}"); where("prox_f", pf, pr, zsl, zper)
    lt = m.L_transpose(v); ltr = o.Lt(v)
    print(f" Lt err {np.nanmax(np.abs(lt - ltr)):.2e} nan {np.isnan(lt).sum()}"); where("Lt", lt, ltr, zsl, zper)
    lz = m.L(z); lzr = o.L(z)
    print(f" L err {np.nanmax(np.abs(lz - lzr)):.2e} nan {np.isnan(lz).sum()}"); where("L", lz, lzr, vsl, vper)
    ph = m.prox_h_conj(v, 0.3); phr = o.prox_h_conj(v, 0.3)
    print(f" prox_h_conj err {np.nanmax(np.abs(ph - phr)):.2e} nan {np.isnan(ph).sum()}"); where("prox_h", ph, phr, vsl, vper)
    r = S.solve_model(m, x0, tol=0.0)
    ro = o.solve(x0, tol=0.0, max_iter=20)
    zz = m.solver_state.z
    print(f" 20 CP its: status {r['status']} its {r['iterations']} {r['solve_ms']/max(1,r['iterations'])*1e3:.1f} us/it  z err {np.nanmax(np.abs(zz - o.z)):.2e} nan {np.isnan(zz).sum()}", flush=True)
    m.close()
