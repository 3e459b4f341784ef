"""Host logic: the driver (spock_abi.cpp) and the kernel bodies (spock_device.hpp / spock_group.hpp), compiled
for the host by tests/emu (emulation harness = test infrastructure, never loaded by the product), against the
oracle.  This is what lets the launch geometry, the ownership maps, the ping-pong parity and the on-device
termination logic be debugged without a GPU; the real parity tests are tests/test_gpu_parity.py."""
import os
import subprocess

import numpy as np
import pytest

from conftest import relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libspock_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    return EMU


CASES = {
    "car": (lambda S: S.get_car_fixture(), {}),
    "sh7": (lambda S: S.get_server_heat_specs(7, 5, 2), {}),
    "sh7_split3": (lambda S: S.get_server_heat_specs(7, 5, 2), {"split_stage": 3}),
    "sh6_nu2_split2": (lambda S: S.get_server_heat_specs(6, 5, 2, nu=2), {"split_stage": 2}),
    "d3_dyn": (lambda S: S.get_server_heat_specs(4, 4, 3, p_ref=[0.2, 0.3, 0.5]), {"split_stage": 1}),
    "d3_static": (lambda S: S.get_server_heat_specs(4, 3, 3, nu=2, p_ref=[0.2, 0.3, 0.5]), {"split_stage": 2}),
    "rand_nx12": (lambda S: S.get_random_linear_specs(5, 12, 4), {"split_stage": 2}),
    "tv_risk": (None, {}),
}


def _specs(S, name):
    if name == "tv_risk":        # total-variation risk measure: ny = 3d, a single Nonnegatives cone (risk_measures.jl:138-147)
        tree, cost, dyn, _, con = S.get_server_heat_specs(4, 3, 2)
        return tree, cost, dyn, S.get_uniform_rms_tv_v2([0.4, 0.6], 0.2, 2, 4), con
    return CASES[name][0](S)


@pytest.mark.parametrize("name", list(CASES))
def test_emulated_operators(S, O, emu, name):
    specs = _specs(S, name)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, lib_path=emu, **CASES[name][1])
    o = O.Oracle(flat, O.CP)
    rng = np.random.default_rng(5)
    for scale in (1.0, 0.05, 5.0):
        z = rng.standard_normal(flat.nz) * scale; v = rng.standard_normal(flat.nv) * scale
        x0 = rng.standard_normal(flat.nx) * 0.1
        assert relerr(m.L(z), o.L(z)) <= 1e-13
        assert relerr(m.L_transpose(v), o.Lt(v)) <= 1e-13
        assert relerr(m.prox_f(z, 0.1, x0), o.prox_f(z, 0.1, x0)) <= 1e-12
        assert relerr(m.prox_h_conj(v, 0.3), o.prox_h_conj(v, 0.3)) <= 1e-13
        zo, vo = m.T(z, v, 0.3, 0.25, x0); zr, vr = o.T(z, v, 0.3, 0.25, x0)
        assert relerr(zo, zr) <= 1e-12 and relerr(vo, vr) <= 1e-12
        assert abs(m.spock_dot(z, v, v[:flat.nz] if flat.nv >= flat.nz else z, v, 0.3, 0.2)
                   - o.dot(z, v, v[:flat.nz] if flat.nv >= flat.nz else z, v, 0.3, 0.2)) <= 1e-10 * (1 + abs(np.dot(z, z)))
    m.close()


@pytest.mark.parametrize("name,tol,log", [("car", 1e-3, False), ("sh7_split3", 1e-3, False), ("sh7", 1e-4, True),
                                          ("d3_static", 1e-3, False), ("tv_risk", 1e-3, False)])
def test_emulated_cp_driver(S, O, emu, name, tol, log):
    specs = _specs(S, name)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, lib_path=emu, graph_iters=-6, **CASES[name][1])
    o = O.Oracle(flat, O.CP)
    x0 = np.full(flat.nx, 0.1)
    vb = S.LOG if log else S.SILENT
    r = S.solve_model(m, x0, tol=tol, verbose=vb)
    ro = o.solve(x0, tol=tol, verbose=int(vb), history=6000)
    assert (r["iterations"], r["iter_ref"], r["status"]) == (ro["iterations"], ro["iter_ref"], ro["status"])
    assert relerr(m.solver_state.z, o.z) <= 1e-9 and relerr(m.solver_state.v, o.v) <= 1e-9
    h = m.history()
    assert len(h["xi1"]) == ro["iterations"]
    assert np.max(np.abs(h["xi1"] - ro["hist_xi1"]) / ro["hist_xi1"]) <= 1e-8
    if log:
        assert np.max(np.abs(h["xi"] - ro["hist_xi"]) / ro["hist_xi"]) <= 1e-8
    # warm start + explicit z0/v0 + explicit step sizes (cpock.jl:174-183, cp.jl:198-201)
    r2 = S.solve_model(m, x0 * 1.1, tol=tol, verbose=vb); ro2 = o.solve(x0 * 1.1, tol=tol, verbose=int(vb))
    assert r2["iterations"] == ro2["iterations"] and relerr(m.solver_state.z, o.z) <= 1e-9
    z0 = np.zeros(flat.nz); v0 = np.zeros(flat.nv)
    r3 = S.solve_model(m, x0, tol=tol, verbose=vb, z0=z0, v0=v0, gamma=0.25, sigma=0.3)
    ro3 = o.solve(x0, tol=tol, verbose=int(vb), z0=z0, v0=v0, gamma=0.25, sigma=0.3)
    assert r3["iterations"] == ro3["iterations"] and relerr(m.solver_state.z, o.z) <= 1e-9
    with pytest.raises(ValueError):
        S.solve_model(m, x0, z0=z0)
    m.close()


def test_emulated_team_sweeps_match_thread_sweeps(S, O, emu, monkeypatch):
    """The lane-team S1 sweeps (default for nx > 8) and the one-thread-per-node sweeps are two implementations of the same
    recursion: force each on a small shape and on a large one and compare with the oracle."""
    for specs, kw in ((S.get_server_heat_specs(6, 5, 2, nu=2), {"split_stage": 3}), (S.get_random_linear_specs(5, 12, 4), {"split_stage": 2})):
        flat = S.flatten_problem(*specs)
        o = O.Oracle(flat, O.CP)
        rng = np.random.default_rng(11)
        z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv); x0 = rng.standard_normal(flat.nx) * 0.1
        zr, vr = o.T(z, v, 0.3, 0.25, x0)
        for team in ("0", "1"):
            monkeypatch.setenv("SPOCK_AMD_TEAM", team)
            m = S.build_model(*specs, lib_path=emu, **kw)
            zo, vo = m.T(z, v, 0.3, 0.25, x0)
            assert relerr(zo, zr) <= 1e-12 and relerr(vo, vr) <= 1e-12
            assert relerr(m.prox_f(z, 0.1, x0), o.prox_f(z, 0.1, x0)) <= 1e-12
            m.close()


def test_emulated_known_answer_trace(S, emu, kat_cp):
    m = S.build_model(*S.get_server_heat_specs(7, 5, 2), lib_path=emu, split_stage=3)
    r = S.solve_model(m, np.full(5, 0.1), tol=1e-5, verbose=S.LOG)
    h = m.history()["xi"]
    assert r["iterations"] == 3159 and np.max(np.abs(h - kat_cp) / kat_cp) <= 1e-9
    m.close()


def test_emulated_batch_and_max_iter(S, O, emu):
    specs = S.get_server_heat_specs(5, 5, 2)
    flat = S.flatten_problem(*specs)
    x0 = np.array([[0.1] * 5, [0.2] * 5, [-0.1, 0.1, 0.05, 0.0, 0.3]])
    m = S.build_model(*specs, lib_path=emu, batch=3, split_stage=2, max_iter=1050)
    rs = S.solve_model(m, x0, tol=1e-3)
    for b in range(3):
        o = O.Oracle(flat, O.CP)
        ro = o.solve(x0[b], tol=1e-3, max_iter=1050)
        assert (rs[b]["iterations"], rs[b]["status"]) == (ro["iterations"], ro["status"])
        assert relerr(m.solver_state.z[b], o.z) <= 1e-9
    assert sorted(r["status"] for r in rs) == [0, 0, 1]      # instance 1 needs 1117 iterations > 1050
    m.close()


@pytest.mark.parametrize("alg,gi", [("CP", -4), ("CP", -8), ("CP", 4), ("SP", -3)])
def test_emulated_eager_graph_iters_keep_the_buffer_rotation(S, O, emu, alg, gi):
    """graph_iters < 0 selects eager launches of |graph_iters| iterations per poll; every repetition restarts at buffer slot 0,
    so the count is rounded up to a multiple of 6 (CP: three z buffers x two v buffers) / an even number (SP) whatever its
    sign -- a negative count that is not such a multiple used to give silently wrong iterates from the second repetition on."""
    for kw in ({}, {"split_stage": 3}):
        specs = S.get_server_heat_specs(7, 5, 2)
        flat = S.flatten_problem(*specs)
        a, oa = (S.CP, O.CP) if alg == "CP" else (S.SP, O.SP)
        m = S.build_model(*specs, S.SolverOptions(a, None), lib_path=emu, max_iter=50, graph_iters=gi, **kw)
        assert m.sizes.graph_iters % (6 if alg == "CP" else 2) == 0 and (m.sizes.graph_iters < 0) == (gi < 0)
        r = S.solve_model(m, np.full(5, 0.1), tol=0.0)
        o = O.Oracle(flat, oa)
        ro = o.solve(np.full(5, 0.1), tol=0.0, max_iter=50)
        assert r["iterations"] == ro["iterations"] == 50 and r["status"] == 1
        assert relerr(m.solver_state.z, o.z) <= (1e-9 if alg == "CP" else 1e-6)
        m.close()


def test_emulated_build_products_match_oracle(S, O, emu):
    """The product's host build step (spock_host.cpp: eigen-decomposition based projector, explicit Rtilde
    inverse) against the oracle's (MGS projector, Cholesky solves): two independent routes to the same numbers."""
    for specs in (S.get_car_fixture(), S.get_server_heat_specs(5, 5, 2, nu=2), S.get_random_linear_specs(4, 6, 3)):
        flat = S.flatten_problem(*specs)
        m = S.build_model(*specs, lib_path=emu)
        bp = O.Oracle(flat, O.CP).build_products()
        assert relerr(m.build_product(0), bp["P"]) <= 1e-12
        assert relerr(m.build_product(1), bp["K"]) <= 1e-12
        assert relerr(m.build_product(3), bp["ABK"]) <= 1e-12
        assert relerr(m.build_product(4), bp["ls"]) <= 1e-13
        assert relerr(m.build_product(5), bp["sqrtQ"]) <= 1e-14
        nu, N = flat.nu, flat.N
        Rinv = m.build_product(2).reshape(N - 1, nu, nu)
        Lc = bp["Rchol"].reshape(N - 1, nu, nu).transpose(0, 2, 1)      # column-major lower factors
        for t in range(N - 1):
            assert np.allclose(Rinv[t] @ (Lc[t] @ Lc[t].T), np.eye(nu), atol=1e-12)
        m.close()


def _sp_compare(S, O, m, flat, x0, tol, vb, n_tight=30):
    o = O.Oracle(flat, O.SP)
    r = S.solve_model(m, x0, tol=tol, verbose=vb)
    ro = o.solve(x0, tol=tol, verbose=int(vb), history=3000)
    h = m.history()
    n = min(n_tight, len(h["xi1"]), len(ro["hist_xi1"]))
    # SP trajectories are chaotically sensitive to rounding beyond ~50 iterations: the first
    # iterations must agree tightly, afterwards only the qualitative behaviour is comparable
    assert np.max(np.abs(h["xi1"][:n] - ro["hist_xi1"][:n]) / ro["hist_xi1"][:n]) <= 1e-8
    assert np.max(np.abs(h["xi2"][:n] - ro["hist_xi2"][:n]) / ro["hist_xi2"][:n]) <= 1e-8
    assert np.array_equal(h["backtrack"][:n], ro["hist_backtrack"][:n])
    assert set(h["backtrack"].tolist()) <= {0, 9}
    assert r["status"] == 0 and ro["status"] == 0
    return r, ro, o


@pytest.mark.parametrize("name,tol,lop", [("car", 1e-3, False), ("sh7", 1e-3, False), ("sh7_split3", 1e-3, False), ("d3_static", 1e-3, False),
                                          ("sh7_split3", 1e-3, True), ("car", 1e-3, True)])
def test_emulated_sp_driver(S, O, emu, name, tol, lop, monkeypatch):
    """SuperMann + Anderson on the kernel code path (emulated) against the oracle's bug-compatible run_sp!.  Both ways of
    forming <r, P r>, <rt, P rt>, <rt, P d>: inside phase B of the two T evaluations (default) and with LOP kernels."""
    if lop:
        monkeypatch.setenv("SPOCK_AMD_SP_LOP", "1")
    specs = S.get_car_fixture(3.7) if name == "car" else _specs(S, name)     # main_profiling.jl:20-28 scales the costs for SP
    flat = S.flatten_problem(*specs)
    kw = dict(CASES[name][1])
    L = None if name != "d3_static" else 1.05 * S.L_norm_sq(specs[0], specs[1], specs[3])   # 3.28 is invalid for d = 3
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, graph_iters=-2, L_norm=L, **kw)
    x0 = np.full(flat.nx, 0.1)
    if L is not None:
        o = O.Oracle(flat, O.SP); o.L_norm = L
        r = S.solve_model(m, x0, tol=tol); ro = o.solve(x0, tol=tol, history=3000)
        h = m.history(); n = min(30, len(h["xi1"]), len(ro["hist_xi1"]))
        assert np.max(np.abs(h["xi1"][:n] - ro["hist_xi1"][:n]) / ro["hist_xi1"][:n]) <= 1e-8
        assert r["status"] == 0
    else:
        r, ro, o = _sp_compare(S, O, m, flat, x0, tol, S.SILENT)
        if name == "car":
            assert r["iterations"] == ro["iterations"]
        # warm start keeps the Anderson memory
        r2 = S.solve_model(m, x0 * 1.05, tol=tol); ro2 = o.solve(x0 * 1.05, tol=tol)
        assert r2["status"] == 0 and ro2["status"] == 0 and r2["iterations"] < r["iterations"]   # far fewer than cold
    m.close()


def test_emulated_sp_known_answer_trace(S, O, emu, kat_sp):
    """examples/server_heat/residuals.jl, SP branch (residuals.tikz:3195-3765): first value to 1 ulp-level, first 40
    iterations to 1e-6 (the tolerance north_star states), backtrack counts identical there, converges."""
    specs = S.get_server_heat_specs(7, 5, 2)
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu)
    r = S.solve_model(m, np.full(5, 0.1), tol=1e-5, verbose=S.LOG)
    h = m.history()
    assert abs(h["xi"][0] - kat_sp[0, 0]) <= 1e-14
    assert np.max(np.abs(h["xi"][:40] - kat_sp[:40, 0]) / kat_sp[:40, 0]) <= 1e-6
    assert np.array_equal(h["backtrack"][:40], kat_sp[:40, 1].astype(int))
    assert r["status"] == 0 and h["xi"][-1] <= 1e-5 and r["iterations"] < 1000
    # converged solution: SP and CP agree within the solve tolerance
    mc = S.build_model(*specs, S.SolverOptions(S.CP, None), lib_path=emu)
    S.solve_model(mc, np.full(5, 0.1), tol=1e-5, verbose=S.LOG)
    sl = S.flatten_problem(*specs).z_slices()
    assert abs(m.solver_state.z[sl["s"]][0] - mc.solver_state.z[sl["s"]][0]) <= 2e-3 * abs(mc.solver_state.z[sl["s"]][0])
    m.close(); mc.close()


def _mpc_reference_loop(O, flat, specs, alg, x0, w, tol):
    """examples/server_heat/mpc_simulation.jl:47-54,164-177 with the oracle: solve, apply u = z[u_inds[1:nu]], advance."""
    o = O.Oracle(flat, alg)
    A, B = specs[2].A, specs[2].B
    x = np.array(x0, dtype=np.float64)
    xs, us, its = [x.copy()], [], []
    for t in range(len(w)):
        r = o.solve(x, tol=tol)
        u = o.z[flat.z_slices()["u"]][:flat.nu].copy()
        x = A[w[t]] @ x + B[w[t]] @ u
        xs.append(x.copy()); us.append(u); its.append(r["iterations"])
    return np.array(xs), np.array(us), its


def test_emulated_mpc_closed_loop(S, O, emu):
    """the closed loop of mpc_simulation.jl as one call with the state advanced on the device."""
    specs = S.get_server_heat_specs(5, 5, 2)
    flat = S.flatten_problem(*specs)
    w = np.random.default_rng(1).integers(0, 2, size=(2, 6))
    x0 = np.array([[0.1] * 5, [0.12] * 5])
    m = S.build_model(*specs, S.SolverOptions(S.CP, None), lib_path=emu, batch=2, split_stage=2)
    xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
    for b in range(2):
        xs, us, its = _mpc_reference_loop(O, flat, specs, O.CP, x0[b], w[b], 1e-3)
        assert [r["iterations"] for r in res[b]] == its                      # warm starts behave identically
        assert np.max(np.abs(xt[b] - xs)) <= 1e-9 and np.max(np.abs(ut[b] - us)) <= 1e-9
    m.close()
    # SP: trajectories are only comparable at the level of the solve tolerance (chaotic iterates)
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, batch=2, split_stage=2)
    xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
    for b in range(2):
        xs, us, its = _mpc_reference_loop(O, flat, specs, O.SP, x0[b], w[b], 1e-3)
        assert all(r["status"] == 0 for r in res[b])
        assert np.max(np.abs(xt[b] - xs)) <= 5e-2 and np.max(np.abs(ut[b] - us)) <= 5e-2
    with pytest.raises(S.SpockError):
        S.mpc_simulation(m, x0, np.full((2, 3), 7), tol=1e-3)               # event out of range
    m.close()


def test_emulated_mpc_steps_chained_on_the_device(S, O, emu, monkeypatch):
    """SP closed loops with the steps chained on the device (SP_CTL_M: a finished instance records its step, applies u, advances x0
    and re-arms itself; no instance waits for its batch) against the host loop of solve calls (SPOCK_AMD_MPC_HOST=1: all instances
    in lockstep, one spock_solve per step): the instances are independent and run the same arithmetic, so per-step iteration
    counts, statuses and trajectories are IDENTICAL, whatever the number of iterations per graph replay."""
    specs = S.get_server_heat_specs(5, 5, 2)
    B, steps = 3, 7
    w = np.random.default_rng(4).integers(0, 2, size=(B, steps))
    x0 = np.array([[0.1] * 5, [0.12, -0.05, 0.1, 0.02, 0.2], [-0.1, 0.1, 0.05, 0.0, 0.3]])
    out = {}
    for mode, env in (("chain", {}), ("chain3", {"SPOCK_AMD_MPC_GRAPH_ITERS": "3"}), ("host", {"SPOCK_AMD_MPC_HOST": "1"})):
        for k in ("SPOCK_AMD_MPC_GRAPH_ITERS", "SPOCK_AMD_MPC_HOST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, batch=B, split_stage=2)
        xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
        out[mode] = (xt, ut, [[(r["iterations"], r["status"]) for r in rb] for rb in res], m.solver_state.z.copy())
        # a second call on the same model continues from the state left behind (warm start), as the reference's loop would
        xt2, ut2, res2 = S.mpc_simulation(m, xt[:, -1], w[:, :3], tol=1e-3)
        out[mode + "_2"] = (xt2, ut2, [[(r["iterations"], r["status"]) for r in rb] for rb in res2])
        m.close()
    its = [n for rb in out["host"][2] for n, _ in rb]
    assert len(set(its)) > 2                                                  # the instances do stop at different iterations
    for mode in ("chain", "chain3"):
        assert out[mode][2] == out["host"][2] and out[mode + "_2"][2] == out["host_2"][2]
        for k in (0, 1, 3):
            assert np.array_equal(out[mode][k], out["host"][k])
        assert np.array_equal(out[mode + "_2"][0], out["host_2"][0]) and np.array_equal(out[mode + "_2"][1], out["host_2"][1])


def test_emulated_closed_form_L_norm(S, O, emu):
    """||L||^2 in closed form (3.2746 for the d = 2 AV@R server_heat data, the value the reference
    rounds up to its hard-coded 3.28) against power iteration on the oracle's L, and as the step-size source."""
    for specs in (S.get_server_heat_specs(5, 5, 2), S.get_car_fixture(), S.get_server_heat_specs(4, 4, 3, p_ref=[0.2, 0.3, 0.5])):
        flat = S.flatten_problem(*specs)
        m = S.build_model(*specs, lib_path=emu, L_norm="auto")
        o = O.Oracle(flat, O.CP)
        z = np.random.default_rng(2).standard_normal(flat.nz); lam = 0.0
        for _ in range(400):
            z = o.Lt(o.L(z)); lam = np.linalg.norm(z); z /= lam
        assert abs(m.L_norm_closed_form - lam) <= 1e-6 * lam
        assert abs(m.solver_state.L_norm - lam) <= 1e-6 * lam
        o.L_norm = m.solver_state.L_norm
        x0 = np.full(flat.nx, 0.1)
        r = S.solve_model(m, x0, tol=1e-3); ro = o.solve(x0, tol=1e-3)
        assert r["iterations"] == ro["iterations"] and r["status"] == 0
        m.close()
    m = S.build_model(*S.get_server_heat_specs(5, 5, 2), lib_path=emu)
    assert m.solver_state.L_norm == 10.0 and abs(m.L_norm_closed_form - 3.2746211638) <= 1e-6
    m.close()


def _variant_specs(S, name):
    """Problem variants that reach the less-travelled kernel paths."""
    rng = np.random.default_rng(42)
    if name == "per_node_dense_costs":          # CM_GENERAL, compile-time shape (5, 2)
        tree, cost, dyn, rms, con = S.get_server_heat_specs(5, 5, 2, nu=2)
        def spd(n, k):
            a = rng.standard_normal((n, n)); return 0.05 * (a @ a.T) / n + np.eye(n) * (0.05 + 0.01 * (k % 3))
        cost = S.Cost([spd(5, k) for k in range(tree.n - 1)], [spd(2, k) + np.eye(2) for k in range(tree.n - 1)],
                      [spd(5, k) for k in range(tree.n_leaf_nodes)])
        return (tree, cost, dyn, rms, con), {"split_stage": 2}
    if name == "shared_dense_costs":            # CM_SHARED (not diagonal)
        tree, cost, dyn, rms, con = S.get_server_heat_specs(5, 5, 2)
        a = rng.standard_normal((5, 5)); Q = 0.05 * (a @ a.T) / 5 + 0.05 * np.eye(5)
        return (tree, S.Cost([Q], [np.eye(5) + 0.2 * Q], [Q.copy()]), dyn, rms, con), {"split_stage": 2}
    if name == "per_node_costs_large_nx":       # CM_GENERAL, run-time dimensions, lane-team sweeps with per-thread local phases
        tree, cost, dyn, rms, con = S.get_random_linear_specs(4, 10, 3)
        cost = S.Cost([np.eye(10) * (0.1 + 0.01 * k) for k in range(tree.n - 1)], [np.eye(3) * (1 + 0.05 * k) for k in range(tree.n - 1)],
                      [np.eye(10) * (0.1 + 0.02 * k) for k in range(tree.n_leaf_nodes)])
        return (tree, cost, dyn, rms, con), {"split_stage": 2}
    if name == "soc_risk_cone":                 # second-order sub-cone inside K (general dual-cone projection)
        tree, cost, dyn, _, con = S.get_server_heat_specs(4, 3, 2)
        E = np.vstack([0.95 * np.eye(2), -np.eye(2), np.ones((1, 2))]); F = np.zeros((5, 2)); b = np.array([0.3, 0.7, 0, 0, 1.0])
        rms = S.get_uniform_rms_v2(E, F, b, S.ConvexCone([(S.CONE_SOC, 3), (S.CONE_NONNEG, 1), (S.CONE_REALS, 1)]), 2, 4)
        return (tree, cost, dyn, rms, con), {}
    if name == "two_stage_tree":                # N = 2: the root and its leaves only
        return S.get_server_heat_specs(2, 5, 2, nu=2), {}
    if name == "d4_tree":                       # branching factor 4 (run-time dimensions)
        return S.get_server_heat_specs(3, 3, 4, p_ref=[0.1, 0.2, 0.3, 0.4]), {}
    if name == "nonuniform_avar":               # one AV@R per non-leaf node (risk_measures.jl:127-136): per-node b and S2 projector
        tree, cost, dyn, _, con = S.get_server_heat_specs(6, 5, 2)
        return (tree, cost, dyn, S.get_nonuniform_rms_avar_v2(2, 6, np.random.default_rng(7)), con), {"split_stage": 3}
    if name == "nonuniform_avar_large_nx":      # the same on the lane-team node-local phases
        tree, cost, dyn, _, con = S.get_random_linear_specs(5, 12, 4)
        return (tree, cost, dyn, S.get_nonuniform_rms_avar_v2(2, 5, np.random.default_rng(8)), con), {"split_stage": 2}
    if name == "tv_risk":                       # total-variation risk measure (risk_measures.jl:138-147): ny = 3 d, F != 0
        tree, cost, dyn, _, con = S.get_server_heat_specs(5, 3, 2)
        return (tree, cost, dyn, S.get_uniform_rms_tv_v2([0.4, 0.6], 0.2, 2, 5), con), {"split_stage": 2}
    if name == "nonpos_cone":                   # a Nonpositives sub-cone in K (risk_measures.jl:29-34): AV@R with its second block negated
        tree, cost, dyn, _, con = S.get_server_heat_specs(5, 5, 2)
        E = np.vstack([0.95 * np.eye(2), np.eye(2), np.ones((1, 2))]); F = np.zeros((5, 2)); b = np.array([0.3, 0.7, 0, 0, 1.0])
        rms = S.get_uniform_rms_v2(E, F, b, S.ConvexCone([(S.CONE_NONNEG, 2), (S.CONE_NONPOS, 2), (S.CONE_ZEROS, 1)]), 2, 5)
        return (tree, cost, dyn, rms, con), {"split_stage": 2}
    raise KeyError(name)


VARIANTS = ["per_node_dense_costs", "shared_dense_costs", "per_node_costs_large_nx", "soc_risk_cone", "two_stage_tree", "d4_tree",
            "nonuniform_avar", "nonuniform_avar_large_nx", "tv_risk", "nonpos_cone"]


@pytest.mark.parametrize("name", VARIANTS)
def test_emulated_variants(S, O, emu, name):
    specs, kw = _variant_specs(S, name)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, lib_path=emu, L_norm="auto", **kw)
    o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
    rng = np.random.default_rng(9)
    for scale in (1.0, 0.05, 5.0):
        z = rng.standard_normal(flat.nz) * scale; v = rng.standard_normal(flat.nv) * scale; x0 = rng.standard_normal(flat.nx) * 0.1
        assert relerr(m.L(z), o.L(z)) <= 1e-12 and relerr(m.L_transpose(v), o.Lt(v)) <= 1e-12
        assert relerr(m.prox_f(z, 0.1, x0), o.prox_f(z, 0.1, x0)) <= 1e-11
        assert relerr(m.prox_h_conj(v, 0.3), o.prox_h_conj(v, 0.3)) <= 1e-12
        zo, vo = m.T(z, v, 0.3, 0.25, x0); zr, vr = o.T(z, v, 0.3, 0.25, x0)
        assert relerr(zo, zr) <= 1e-11 and relerr(vo, vr) <= 1e-11
    x0 = np.full(flat.nx, 0.1)
    r = S.solve_model(m, x0, tol=1e-3); ro = o.solve(x0, tol=1e-3)
    assert (r["iterations"], r["status"]) == (ro["iterations"], ro["status"])
    assert relerr(m.solver_state.z, o.z) <= 1e-8
    m.close()
    if name.startswith("nonuniform"):     # the SP driver's inner products use rms[i].b as well (spock_dot, implicit_l.jl:485-516)
        ms = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, L_norm="auto", max_iter=25, **kw)
        os_ = O.Oracle(flat, O.SP); os_.L_norm = ms.solver_state.L_norm
        S.solve_model(ms, x0, tol=1e-12, verbose=S.LOG)
        ro = os_.solve(x0, tol=1e-12, verbose=O.LOG, max_iter=25, history=30)
        h = ms.history()
        assert np.max(np.abs(h["xi"] - ro["hist_xi"]) / ro["hist_xi"]) <= 1e-6 and np.array_equal(h["backtrack"], ro["hist_backtrack"])
        ms.close()


@pytest.mark.parametrize("env,ok", [({}, True), ({"SPOCK_EMU_REVERSE_BLOCKS": "1"}, True), ({"SPOCK_EMU_REVERSE": "1"}, True),
                                    ({"SPOCK_AMD_NO_FUSE": "1"}, True),
                                    ({"SPOCK_EMU_REVERSE_BLOCKS": "1", "SPOCK_EMU_NO_V2FIX": "1"}, False)])
def test_emulated_fused_iteration_is_order_independent(S, O, emu, env, ok, monkeypatch):
    """CP iteration = CROWN + SUB_BA (phase B of iteration k fused with phase A of k + 1).  The result must not depend on
    the order in which the workgroups / threads of a launch run, and must equal the unfused three-kernel sequence.  The
    last case switches off the recomputation of the subtree roots' v2 by the crown's workgroup: the emulation with the
    crown's workgroup running first must then expose the cross-workgroup read-after-write the recomputation removes."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SPOCK_AMD_RLANE", "0")   # (this test is about SUB_BA; the role-lane kernel that replaces it for few workgroups has its own)
    specs = S.get_server_heat_specs(7, 5, 2, nu=2)
    flat = S.flatten_problem(*specs)
    o = O.Oracle(flat, O.CP)
    x0 = np.full(flat.nx, 0.1)
    ro = o.solve(x0, tol=1e-3)
    for split in (1, 3, 5):
        m = S.build_model(*specs, lib_path=emu, graph_iters=-6, split_stage=split, batch=2)
        r = S.solve_model(m, np.stack([x0, 1.3 * x0]), tol=1e-3)
        good = r[0]["iterations"] == ro["iterations"] and relerr(m.solver_state.z[0], o.z) <= 1e-9 and relerr(m.solver_state.v[0], o.v) <= 1e-9
        m.close()
        assert good == ok, (split, env)


def test_emulated_group_leaf_vector_in_global_memory(S, O, emu, monkeypatch):
    """Large nx with deep groups (BASELINE config 5 at N = 15): the lane-team sweep keeps c / e / u in LDS and the vector over
    the group leaves in global memory.  Forced here on small problems; operators, CP and SP against the oracle."""
    monkeypatch.setenv("SPOCK_AMD_FORCE_XL_GLOBAL", "1")
    for specs, kw in ((S.get_random_linear_specs(5, 12, 4), {"split_stage": 2}), (S.get_server_heat_specs(6, 20, 2), {}),
                      (S.get_server_heat_specs(4, 10, 3, p_ref=[0.2, 0.3, 0.5]), {"split_stage": 1})):
        flat = S.flatten_problem(*specs)
        m = S.build_model(*specs, lib_path=emu, L_norm="auto", graph_iters=-6, **kw)
        o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
        rng = np.random.default_rng(4)
        z = rng.standard_normal(flat.nz); v = rng.standard_normal(flat.nv); x0 = np.full(flat.nx, 0.1)
        zo, vo = m.T(z, v, 0.3, 0.25, x0); zr, vr = o.T(z, v, 0.3, 0.25, x0)
        assert relerr(zo, zr) <= 1e-11 and relerr(vo, vr) <= 1e-11
        assert relerr(m.prox_f(z, 0.1, x0), o.prox_f(z, 0.1, x0)) <= 1e-11
        r = S.solve_model(m, x0, tol=1e-3); ro = o.solve(x0, tol=1e-3)
        assert (r["iterations"], r["status"]) == (ro["iterations"], ro["status"]) and relerr(m.solver_state.z, o.z) <= 1e-8
        m.close()
    specs = S.get_server_heat_specs(6, 20, 2)
    flat = S.flatten_problem(*specs)
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, graph_iters=-2)
    o = O.Oracle(flat, O.SP)
    r = S.solve_model(m, np.full(20, 0.1), tol=1e-3); ro = o.solve(np.full(20, 0.1), tol=1e-3, history=3000)
    h = m.history(); n = min(20, len(h["xi1"]), len(ro["hist_xi1"]))
    assert np.max(np.abs(h["xi1"][:n] - ro["hist_xi1"][:n]) / ro["hist_xi1"][:n]) <= 1e-8 and r["status"] == 0
    m.close()


@pytest.mark.parametrize("env", [{}, {"SPOCK_AMD_FLAT_FUSE": "0"}, {"SPOCK_EMU_REVERSE": "1", "SPOCK_EMU_REVERSE_BLOCKS": "1"}])
def test_emulated_flat_node_local_kernels(S, O, emu, monkeypatch, env):
    """Lane-team shapes (nx > 8): the node-local phases as FLAT kernels over chunks of 16 nodes (spock_flat_local.hpp), the subtree /
    crown kernels running the sweeps only -- c_i, e_i, q_leaf travel through zbar's own x / u entries.  CP runs FLAT_BA (phase B of
    iteration k + phase A of k + 1, vbar.v2 of non-leaf children recomputed by the parent's team); SPOCK_AMD_FLAT_FUSE=0 the
    separate FLAT_A / FLAT_B; SP always the separate ones with the inner products in FLAT_B_SP.  Against the oracle, in both
    thread and workgroup orders; equal to rounding to the subtree kernels' own node-local phases (SPOCK_AMD_FLAT_LOCAL=0), whose
    team-wide sums run in another order."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for specs, kw in ((S.get_random_linear_specs(5, 12, 4), {"split_stage": 2}), (S.get_server_heat_specs(6, 20, 2), {"split_stage": 3}),
                      (S.get_server_heat_specs(5, 20, 3, p_ref=[0.2, 0.3, 0.5]), {"split_stage": 2, "L_norm": "auto"})):
        flat = S.flatten_problem(*specs)
        x0 = 0.1 * np.cos(np.arange(flat.nx))
        m = S.build_model(*specs, lib_path=emu, max_iter=40, **kw)
        seq = [n for n, _, _ in m.profile_sequence(1)]
        assert seq == (["flat_a", "sub_a", "crown", "sub_b", "flat_b"] if env.get("SPOCK_AMD_FLAT_FUSE") == "0" else ["sub_a", "crown", "sub_b", "flat_ba"])
        m.reset()
        r = S.solve_model(m, x0, tol=1e-12, verbose=S.LOG)
        o = O.Oracle(flat, O.CP); o.L_norm = m.solver_state.L_norm
        ro = o.solve(x0, tol=1e-12, verbose=O.LOG, max_iter=40, history=50)
        h = m.history()
        assert r["iterations"] == ro["iterations"] == 40
        assert np.max(np.abs(h["xi"] - ro["hist_xi"]) / ro["hist_xi"]) <= 1e-11
        assert relerr(m.solver_state.z, o.z) <= 1e-11 and relerr(m.solver_state.v, o.v) <= 1e-11
        z_flat = m.solver_state.z.copy()
        m.close()
        monkeypatch.setenv("SPOCK_AMD_FLAT_LOCAL", "0")
        m0 = S.build_model(*specs, lib_path=emu, max_iter=40, **kw)
        assert [n for n, _, _ in m0.profile_sequence(1)] == ["sub_a", "crown", "sub_b"]
        m0.reset(); S.solve_model(m0, x0, tol=1e-12, verbose=S.LOG)
        assert relerr(m0.solver_state.z, z_flat) <= 1e-13
        m0.close()
        monkeypatch.delenv("SPOCK_AMD_FLAT_LOCAL")
        # SP: FLAT_A + sweeps + FLAT_B_SP per T evaluation
        ms = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, max_iter=12, **kw)
        S.solve_model(ms, x0, tol=1e-12, verbose=S.LOG)
        os_ = O.Oracle(flat, O.SP); os_.L_norm = ms.solver_state.L_norm
        rs = os_.solve(x0, tol=1e-12, verbose=O.LOG, max_iter=12, history=20)
        hs = ms.history()
        assert np.max(np.abs(hs["xi"] - rs["hist_xi"]) / rs["hist_xi"]) <= 1e-9 and np.array_equal(hs["backtrack"], rs["hist_backtrack"])
        ms.close()


def test_emulated_nan_and_state_api(S, O, emu):
    specs = S.get_server_heat_specs(4, 5, 2, nu=2)
    flat = S.flatten_problem(*specs)
    for alg in (S.CP, S.SP):
        m = S.build_model(*specs, S.SolverOptions(alg, None), lib_path=emu)
        r = S.solve_model(m, np.array([0.1, np.nan, 0.1, 0.1, 0.1]), tol=1e-3)
        assert r["status"] == 2 and r["iterations"] <= 2                       # SPOCK_NAN, stops at once
        m.reset()
        assert np.all(m.solver_state.z == 0) and np.all(m.solver_state.v == 0)
        r = S.solve_model(m, np.full(5, 0.1), tol=1e-3)
        assert r["status"] in (0, 1)
        z, v = m.solver_state.z.copy(), m.solver_state.v.copy()
        m.set_state(z * 0.5, v * 0.5)
        assert np.array_equal(m.solver_state.z, z * 0.5) and np.array_equal(m.solver_state.v, v * 0.5)
        m.close()


def test_unsupported_inputs(S, emu):
    tree, cost, dyn, rms, con = S.get_server_heat_specs(4, 3, 2)
    bad = S.generate_scenario_tree_uniform_branching_factor_v2(4, 2, 3, 3)
    bad.anc = bad.anc.copy(); bad.anc[5] = 1
    with pytest.raises(S.SpockError):
        S.build_model(bad, cost, dyn, rms, con, lib_path=emu)


@pytest.mark.parametrize("N,nu,x0s,mb,cap,tol", [(5, 5, 0.1, 0, 400, 1e-3), (7, 2, 0.3, 2, 70, 1e-4), (5, 2, 1.0, 2, 80, 1e-4)])
def test_emulated_sp_corrected_variant(S, O, emu, N, nu, x0s, mb, cap, tol):
    """spock_options_t.sp_variant = 1 (least-squares Anderson, a line search that shortens the step; no reference counterpart)
    against the oracle's statement of the same algorithm: a converging solve, and two harder problems whose line search rejects
    candidates at every depth down to the plain CP step.  Two instances per handle: a candidate accepted by one instance parks
    only that instance for the rest of the iteration."""
    specs = S.get_server_heat_specs(N, 5, 2, nu=nu)
    flat = S.flatten_problem(*specs)
    x0 = np.array([np.full(5, x0s), np.full(5, 0.7 * x0s)])
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, batch=2, max_iter=cap, sp_variant=1, sp_max_backtrack=mb)
    rs = S.solve_model(m, x0, tol=tol, verbose=S.LOG)
    seen = set()
    for b in range(2):
        o = O.Oracle(flat, O.SP, sp_variant=1, sp_max_backtrack=mb)
        ro = o.solve(x0[b], tol=tol, verbose=O.LOG, max_iter=cap, history=cap + 2)
        h = m.history(b)
        n = min(len(h["xi"]), len(ro["hist_xi"]), 40)      # (rounding differences grow along an SP trajectory: DESIGN.md section 4)
        assert np.array_equal(h["backtrack"][:n], ro["hist_backtrack"][:n])
        assert np.max(np.abs(h["xi"][:n] - ro["hist_xi"][:n]) / ro["hist_xi"][:n]) <= 1e-6
        seen |= set(int(x) for x in h["backtrack"])
        if ro["status"] == 0:
            assert (rs[b]["iterations"], rs[b]["status"]) == (ro["iterations"], 0)
            assert relerr(m.solver_state.z[b], o.z) <= 1e-6
    if mb:
        assert {0, 1, mb + 1} <= seen        # accepted at once, after one shortening, and every candidate rejected
    m.close()
    with pytest.raises(S.SpockError):
        S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, sp_variant=2)


def test_emulated_build_step_on_the_device(S, O, emu):
    """spock_update_dynamics: ricatti_offline (implicit_l.jl:46-97) and the composite sweep matrices recomputed by the device-side
    build step for new dynamics of the same shape, against a model built from scratch on the host with those dynamics: build
    products P, K, Rtilde^-1, A + B K to rounding, the same CP iterates, the closed loop advancing with the new A, B."""
    import dataclasses
    tree, cost, dyn, rms, con = S.get_server_heat_specs(6, 5, 2)
    rng = np.random.default_rng(3)
    dyn2 = dataclasses.replace(dyn, A=[0.9 * a + 0.01 * rng.standard_normal(a.shape) for a in dyn.A],
                               B=[b + 0.05 * rng.standard_normal(b.shape) for b in dyn.B])
    x0 = np.array([0.1, -0.1, 0.05, 0.2, 0.3])
    ref = S.build_model(tree, cost, dyn2, rms, con, lib_path=emu, max_iter=60)
    m = S.build_model(tree, cost, dyn, rms, con, lib_path=emu, max_iter=60)
    S.solve_model(m, x0, tol=1e-3); m.reset()          # (a solve with the old dynamics first: captured graphs are dropped)
    m.update_dynamics(dyn2)
    for which in (0, 1, 2, 3):      # P, K, Rtilde^-1, A + B K
        a, b = m.build_product(which), ref.build_product(which)
        assert a.shape == b.shape and relerr(a, b) <= 1e-12, which
    r1 = S.solve_model(m, x0, tol=1e-6); r2 = S.solve_model(ref, x0, tol=1e-6)
    assert r1["iterations"] == r2["iterations"] == 60
    assert relerr(m.solver_state.z, ref.solver_state.z) <= 1e-10
    o = O.Oracle(S.flatten_problem(tree, cost, dyn2, rms, con), O.CP); o.solve(x0, tol=1e-6, max_iter=60)
    assert relerr(m.solver_state.z, o.z) <= 1e-9
    w = np.array([[0, 1, 1]])
    m.reset(); ref.reset()
    xa, ua, _ = S.mpc_simulation(m, x0, w, tol=1e-3); xb, ub, _ = S.mpc_simulation(ref, x0, w, tol=1e-3)
    assert np.max(np.abs(xa - xb)) <= 1e-10 and np.max(np.abs(ua - ub)) <= 1e-10
    bad = dataclasses.replace(dyn, A=[np.full_like(a, np.nan) for a in dyn.A])
    with pytest.raises(S.SpockError):
        m.update_dynamics(bad)
    # a build that FAILS on the device (finite dynamics whose Riccati recursion overflows: Rtilde is then not positive definite)
    # must leave the model as it was -- the kernel rewrites the live pool in place, the driver restores its snapshot
    m.reset()
    S.solve_model(m, x0, tol=1e-6); z_before = m.solver_state.z.copy(); P_before = m.build_product(0)
    A_before = m.problem_definition.A.copy()
    boom = dataclasses.replace(dyn, A=[1e160 * np.eye(5) for _ in dyn.A])
    with pytest.raises(S.SpockError) as err:
        m.update_dynamics(boom)
    assert err.value.code == 4 and "previous dynamics" in str(err.value)
    assert np.array_equal(m.problem_definition.A, A_before) and np.array_equal(m.build_product(0), P_before)
    m.reset(); S.solve_model(m, x0, tol=1e-6)
    assert np.array_equal(m.solver_state.z, z_before)
    m.close(); ref.close()


def test_emulated_batched_draws_of_the_dynamics(S, O, emu):
    """Monte-Carlo draws of the dynamics in ONE handle (north_star: "Monte-Carlo scenario-tree draws"): a batch built
    with per_instance_dynamics, spock_update_dynamics_batch gives every instance its own A_w, B_w -- ricatti_offline (implicit_l.jl:46-97)
    and the sweep matrices are built on the device for all instances side by side -- and each instance then solves ITS problem: against one
    oracle per draw (CP iterates, SP residual log, the closed loop advancing with the instance's own A, B).  A draw whose build fails
    leaves that instance, and only that one, with its previous dynamics."""
    import dataclasses
    for N, nx in ((6, 5), (5, 12)):
        tree, cost, dyn, rms, con = S.get_server_heat_specs(N, nx, 2)
        Bn = 3
        rng = np.random.default_rng(7)
        draws = [dataclasses.replace(dyn, A=[(0.85 + 0.05 * b) * a + 0.01 * rng.standard_normal(a.shape) for a in dyn.A],
                                     B=[bb + 0.05 * rng.standard_normal(bb.shape) for bb in dyn.B]) for b in range(Bn)]
        x0 = 0.1 * (1 + 0.1 * np.arange(Bn))[:, None] * np.cos(np.arange(nx))[None, :]
        for alg, oalg in ((S.CP, O.CP), (S.SP, O.SP)):
            m = S.build_model(tree, cost, dyn, rms, con, S.SolverOptions(alg, None), batch=Bn, per_instance_dynamics=True, max_iter=30, lib_path=emu)
            assert m.sizes.interleaved == 0
            assert m.update_dynamics_batch(draws) == [0] * Bn
            rs = S.solve_model(m, x0, tol=1e-12, verbose=S.LOG)
            for b in range(Bn):
                o = O.Oracle(S.flatten_problem(tree, cost, draws[b], rms, con), oalg)
                ro = o.solve(x0[b], tol=1e-12, verbose=O.LOG, max_iter=30, history=40)
                h = m.history(b)
                assert rs[b]["iterations"] == ro["iterations"] == 30
                assert np.max(np.abs(h["xi"] - ro["hist_xi"]) / ro["hist_xi"]) <= (1e-9 if alg == S.CP else 1e-6)
                if alg == S.CP:
                    assert relerr(m.solver_state.z[b], o.z) <= 1e-9 and relerr(m.solver_state.v[b], o.v) <= 1e-9
            m.close()
        # closed loop: every instance advances with its own A_w, B_w; against one single-instance model per draw.  CP (whose
        # trajectories are stable: same per-step iteration counts) and SP chained on the device (chaotic w.r.t. the rounding
        # differences between the device build and the host build: every step converges, trajectories at the solve tolerance)
        w = np.array([[0, 1, 1, 0], [1, 1, 0, 0], [0, 0, 1, 1]], dtype=np.int32)
        for alg in (S.CP, S.SP):
            m = S.build_model(tree, cost, dyn, rms, con, S.SolverOptions(alg, None), batch=Bn, per_instance_dynamics=True, lib_path=emu)
            m.update_dynamics_batch(draws)
            xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
            for b in range(Bn):
                m1 = S.build_model(tree, cost, draws[b], rms, con, S.SolverOptions(alg, None), lib_path=emu)
                x1, u1, r1 = S.mpc_simulation(m1, x0[b], w[b:b + 1], tol=1e-3)
                assert all(r["status"] == 0 for r in res[b])
                if alg == S.CP:
                    assert [r["iterations"] for r in res[b]] == [r["iterations"] for r in r1[0]]
                    assert np.max(np.abs(xt[b] - x1[0])) <= 1e-8 and np.max(np.abs(ut[b] - u1[0])) <= 1e-8
                else:
                    assert np.max(np.abs(xt[b] - x1[0])) <= 5e-2 and np.max(np.abs(ut[b] - u1[0])) <= 5e-2
                m1.close()
            if alg == S.CP:
                m.close()
        # one bad draw: the other instances get theirs, the bad one keeps what it had
        z_before = None
        S.solve_model(m, x0, tol=1e-3); m.reset()
        rs0 = S.solve_model(m, x0, tol=1e-3); z_before = m.solver_state.z.copy()
        bad = list(draws); bad[1] = dataclasses.replace(dyn, A=[1e160 * np.eye(nx) for _ in dyn.A])
        with pytest.raises(S.SpockError) as err:
            m.update_dynamics_batch(bad)
        assert err.value.code == 4 and m.last_build_status.tolist() == [0, 4, 0]
        m.reset(); S.solve_model(m, x0, tol=1e-3)
        assert np.array_equal(m.solver_state.z, z_before)            # draws 0 and 2 are the same as before, draw 1 was rolled back
        m.close()
        # a handle without per-instance pools refuses
        m = S.build_model(tree, cost, dyn, rms, con, batch=Bn, lib_path=emu)
        with pytest.raises(S.SpockError):
            m.update_dynamics_batch(draws)
        m.close()


# ---- element-major ("interleaved") batched regime: one lane per instance, one wave per tree node (spock_igroup.hpp) -----------
@pytest.mark.parametrize("N,nu,B,kw", [(5, 5, 3, {}), (7, 2, 3, {}), (7, 5, 66, {}), (8, 2, 4, {"split_stage": 5}),
                                       (10, 5, 2, {"split_stage": 8}), (8, 2, 70, {"split_stage": 2}),
                                       (8, 5, 3, {"split_stage": 4, "stream": "0"}), (7, 5, 66, {"stream": "0"})])
def test_emulated_interleaved_layout_matches_oracle(S, O, emu, monkeypatch, N, nu, B, kw):
    """The element-major kernels against the oracle: different instances stop at different iterations (finished lanes freeze),
    partial tiles (B not a multiple of 64), two tiles, one / two / three layers of groups (split 5 of N = 8: layers [0, 1, 5];
    split 8 of N = 10: [0, 4, 8]), tile groups on separate streams, warm start, LOG-mode kernel variant, set_state / reset.
    Below the split stage the level-streaming kernels with their role waves (spock_ilevel.hpp; split 2 of N = 8: five streamed
    stages under a three-node crown); "stream": "0" = the bottom-group kernels instead."""
    kw = dict(kw)
    monkeypatch.setenv("SPOCK_AMD_ILV_STREAM", kw.pop("stream", "1"))
    monkeypatch.setenv("SPOCK_AMD_ILV_TGROUPS", "2")
    specs = S.get_server_heat_specs(N, 5, 2, nu=nu)
    flat = S.flatten_problem(*specs)
    cap = 80 if N >= 8 else (300 if nu == 2 else 1200)
    x0 = 0.1 * (1 + 0.1 * np.arange(B))[:, None] * np.ones((B, 5)); x0[B // 2] = [-0.1, 0.1, 0.05, 0.0, 0.3]
    m = S.build_model(*specs, lib_path=emu, batch=B, layout="interleaved", max_iter=cap, **kw)
    assert m.sizes.interleaved == 1
    rs = S.solve_model(m, x0, tol=1e-3)
    Z = m.solver_state.z.reshape(B, -1).copy(); V = m.solver_state.v.reshape(B, -1).copy()
    picks = sorted(set([0, B // 2, B - 1]))
    for b in picks:
        o = O.Oracle(flat, O.CP); ro = o.solve(x0[b], tol=1e-3, max_iter=cap)
        assert (rs[b]["iterations"], rs[b]["status"]) == (ro["iterations"], ro["status"])
        assert relerr(Z[b], o.z) <= 1e-9 and relerr(V[b], o.v) <= 1e-9
    if N <= 7 and nu == 5:      # (the nu = 2 < nx extension is infeasible: no instance converges)
        assert len(set(r["iterations"] for r in rs)) > 1        # the instances did stop at different iterations
    # warm start from the state left in the model, LOG-mode termination rule and residual log
    rs2 = S.solve_model(m, 0.9 * x0, tol=1e-3, verbose=S.LOG)
    o = O.Oracle(flat, O.CP); o.solve(x0[0], tol=1e-3, max_iter=cap)
    ro2 = o.solve(0.9 * x0[0], tol=1e-3, verbose=O.LOG, max_iter=cap, history=cap + 2)
    h = m.history(0)
    assert rs2[0]["iterations"] == ro2["iterations"] and len(h["xi"]) == len(ro2["hist_xi"])
    assert np.max(np.abs(h["xi"] - ro2["hist_xi"]) / ro2["hist_xi"]) <= 1e-8
    assert relerr(m.solver_state.z.reshape(B, -1)[0], o.z) <= 1e-9
    # set_state / z0, v0 / reset cross the boundary instance-major
    m.set_state(Z, V)
    assert np.array_equal(m.solver_state.z.reshape(B, -1), Z) and np.array_equal(m.solver_state.v.reshape(B, -1), V)
    rs3 = S.solve_model(m, x0, tol=1e-3, z0=np.zeros_like(Z), v0=np.zeros_like(V))
    assert [r["iterations"] for r in rs3] == [r["iterations"] for r in rs]
    assert np.array_equal(m.solver_state.z.reshape(B, -1), Z)     # bitwise repeatable
    m.reset()
    assert not m.solver_state.z.any()
    m.close()


@pytest.mark.parametrize("name", ["shared_dense_costs", "d3_tree", "nx2_nu1", "soc_risk_cone"])
def test_emulated_interleaved_shapes_and_cost_modes(S, O, emu, name):
    """The role waves of the level-streaming kernels on the other compile-time shapes and cost modes: shared dense cost roots
    (CM_SHARED), branching factor 3 (three C_c and three L_c waves per workgroup), nx = 2 / nu = 1, a second-order sub-cone in the
    risk cone; CP against the oracle in SILENT and LOG mode."""
    rng = np.random.default_rng(5)
    if name == "shared_dense_costs":
        tree, cost, dyn, rms, con = S.get_server_heat_specs(6, 5, 2)
        a = rng.standard_normal((5, 5)); Q = 0.05 * (a @ a.T) / 5 + 0.05 * np.eye(5)
        specs = (tree, S.Cost([Q], [np.eye(5) + 0.2 * Q], [Q.copy()]), dyn, rms, con)
    elif name == "d3_tree":
        specs = S.get_random_linear_specs(5, 3, 2, d=3)
    elif name == "nx2_nu1":
        specs = S.get_random_linear_specs(7, 2, 1)
    else:
        tree, cost, dyn, _, con = S.get_server_heat_specs(6, 5, 2)
        E = np.vstack([0.95 * np.eye(2), -np.eye(2), np.ones((1, 2))]); F = np.zeros((5, 2)); b = np.array([0.3, 0.7, 0, 0, 1.0])
        specs = (tree, cost, dyn, S.get_uniform_rms_v2(E, F, b, S.ConvexCone([(S.CONE_SOC, 3), (S.CONE_NONNEG, 1), (S.CONE_REALS, 1)]), 2, 6), con)
    flat = S.flatten_problem(*specs)
    B, nx = 3, flat.nx
    x0 = 0.1 * (1 + 0.3 * np.arange(B))[:, None] * np.cos(np.arange(nx))[None, :]
    m = S.build_model(*specs, lib_path=emu, batch=B, layout="interleaved", max_iter=50, split_stage=2)
    assert m.sizes.interleaved == 1
    for verbose, ov in ((S.SILENT, O.SILENT), (S.LOG, O.LOG)):
        m.reset()
        rs = S.solve_model(m, x0, tol=1e-4, verbose=verbose)
        for b in range(B):
            o = O.Oracle(flat, O.CP); ro = o.solve(x0[b], tol=1e-4, max_iter=50, verbose=ov)
            assert (rs[b]["iterations"], rs[b]["status"]) == (ro["iterations"], ro["status"])
            assert relerr(m.solver_state.z[b], o.z) <= 1e-9 and relerr(m.solver_state.v[b], o.v) <= 1e-9
    m.close()


def test_emulated_profile_sequence_lists_the_launches_of_an_iteration(S, emu):
    """spock_profile_sequence: one entry per launch of a CP iteration, in launch order -- the crown, the level kernels LB top-down,
    the bottom stage's LBA, LA bottom-up; the iterate left behind is the one `reps` iterations further on."""
    specs = S.get_server_heat_specs(8, 5, 2, nu=2)
    m = S.build_model(*specs, lib_path=emu, batch=3, layout="interleaved", max_iter=12)
    x0 = np.full((3, 5), 0.1)
    S.solve_model(m, x0, tol=0.0)
    t0 = m.sizes.split_stage
    seq = m.profile_sequence(2)
    # (default: the two bottom stages 5 and 6 as one launch, LBA2; SPOCK_AMD_ILV_FUSE2=0: lb .. 5, lba 6, la 5 ..)
    assert [(n, lv) for n, lv, _ in seq] == [("crown", 0)] + [("lb", t) for t in range(t0, 5)] + [("lba2", 5)] + [("la", t) for t in range(4, t0 - 1, -1)]
    assert all(ms >= 0.0 for _, _, ms in seq)
    z14 = m.solver_state.z.copy()
    ref = S.build_model(*specs, lib_path=emu, batch=3, layout="interleaved", max_iter=14)
    S.solve_model(ref, x0, tol=0.0)
    assert relerr(z14, ref.solver_state.z) <= 1e-12
    m.close(); ref.close()


@pytest.mark.parametrize("env", [{}, {"SPOCK_EMU_REVERSE": "1"}, {"SPOCK_EMU_REVERSE_BLOCKS": "1"}])
def test_emulated_role_lane_subtree_kernel(S, O, emu, monkeypatch, env):
    """MODE_SUB_BA_R (spock_rgroup.hpp): the fused iteration's subtree kernel with the node-local phases split over role lanes (wave =
    role, lane = node) -- chosen for compile-time shapes with few workgroups in flight.  Against the oracle (SILENT and LOG mode: the
    kernel variant with the combined residual), in both thread / workgroup orders, several splits (subtrees of one to five levels, a
    one-node crown), d = 2 and d = 3, a second-order risk cone, two instances; equal to rounding to MODE_SUB_BA (SPOCK_AMD_RLANE=0), whose sums over the children
    run in another order."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = [(S.get_server_heat_specs(7, 5, 2, nu=2), s) for s in (1, 3, 5)] + [(S.get_server_heat_specs(6, 5, 2), 2), (S.get_random_linear_specs(5, 3, 2, d=3), 2),
                                                                                (S.get_random_linear_specs(7, 2, 1), 3)]
    tree, cost, dyn, _, con = S.get_server_heat_specs(6, 5, 2, nu=2)     # a second-order sub-cone in the risk cone (role R's projection)
    E = np.vstack([0.95 * np.eye(2), -np.eye(2), np.ones((1, 2))]); F = np.zeros((5, 2)); b = np.array([0.3, 0.7, 0, 0, 1.0])
    cases.append(((tree, cost, dyn, S.get_uniform_rms_v2(E, F, b, S.ConvexCone([(S.CONE_SOC, 3), (S.CONE_NONNEG, 1), (S.CONE_REALS, 1)]), 2, 6), con), 2))
    for specs, split in cases:
        flat = S.flatten_problem(*specs)
        x0 = np.stack([0.1 * np.cos(np.arange(flat.nx)), 0.13 * np.ones(flat.nx)])
        out = {}
        for rl in ("1", "0"):
            monkeypatch.setenv("SPOCK_AMD_RLANE", rl)
            m = S.build_model(*specs, lib_path=emu, graph_iters=-6, split_stage=split, batch=2, max_iter=60)
            names = [n for n, _, _ in m.profile_sequence(1)]
            # (SPOCK_AMD_RLANE=0: SUB_BA for nx + nu <= 8, the plain three-kernel sequence for the larger shapes)
            assert names == ["crown", "sub_ba_r"] if rl == "1" else names in (["crown", "sub_ba"], ["sub_a", "crown", "sub_b_lat"], ["sub_a", "crown", "sub_b"]), names
            res = []
            for verbose, ov in ((S.SILENT, O.SILENT), (S.LOG, O.LOG)):
                m.reset()
                rs = S.solve_model(m, x0, tol=1e-5, verbose=verbose)
                res.append((m.solver_state.z.copy(), m.solver_state.v.copy(), [r["iterations"] for r in rs]))
                if rl == "1":
                    for b in range(2):
                        o = O.Oracle(flat, O.CP); ro = o.solve(x0[b], tol=1e-5, max_iter=60, verbose=ov, history=70)
                        assert (rs[b]["iterations"], rs[b]["status"]) == (ro["iterations"], ro["status"])
                        assert relerr(m.solver_state.z[b], o.z) <= 1e-9 and relerr(m.solver_state.v[b], o.v) <= 1e-9
                        if verbose == S.LOG:
                            h = m.history(b)
                            assert np.max(np.abs(h["xi"] - ro["hist_xi"]) / ro["hist_xi"]) <= 1e-8
            out[rl] = res
            m.close()
        for (z1, v1, it1), (z0, v0, it0) in zip(out["1"], out["0"]):
            assert it1 == it0 and relerr(z1, z0) <= 1e-11 and relerr(v1, v0) <= 1e-11


def test_role_lane_kernel_is_chosen_by_the_workgroups_in_flight(S, emu, monkeypatch):
    """rlane_applies (spock_host.cpp): MODE_SUB_BA_R replaces SUB_BA while batch x (subtrees + 1) <= 400 workgroups are in flight (800 for the
    shapes whose alternative is the three-launch sequence: the measured crossovers), for compile-time shapes with shared costs and groups of at most 64 nodes; SPOCK_AMD_RLANE=1 lifts the batch
    rule only, =0 switches the kernel off; SP never uses it; per-instance dynamics keep SUB_BA (one matrix pool per instance)."""
    monkeypatch.delenv("SPOCK_AMD_RLANE", raising=False)
    specs = S.get_server_heat_specs(7, 5, 2, nu=2)
    seq = lambda **kw: [n for n, _, _ in S.build_model(*specs, lib_path=emu, split_stage=3, layout="instance", **kw).profile_sequence(1)]
    assert seq(batch=44) == ["crown", "sub_ba_r"] and seq(batch=45) == ["crown", "sub_ba"]          # 9 workgroups per instance
    specs5 = S.get_server_heat_specs(7, 5, 2)         # nx + nu > 8: the alternative is the three-launch sequence -- up to 800 workgroups
    seq5 = lambda **kw: [n for n, _, _ in S.build_model(*specs5, lib_path=emu, split_stage=3, layout="instance", **kw).profile_sequence(1)]
    assert seq5(batch=88) == ["crown", "sub_ba_r"] and seq5(batch=89)[0] == "sub_a"
    assert seq(batch=2, per_instance_dynamics=True) == ["crown", "sub_ba"]
    monkeypatch.setenv("SPOCK_AMD_RLANE", "1")
    assert seq(batch=45) == ["crown", "sub_ba_r"]
    big = S.get_server_heat_specs(9, 5, 2, nu=2)      # split 1: subtrees of 255 nodes -- no lane per node
    assert [n for n, _, _ in S.build_model(*big, lib_path=emu, split_stage=1).profile_sequence(1)] == ["crown", "sub_ba"]
    monkeypatch.setenv("SPOCK_AMD_RLANE", "0")
    assert seq(batch=1) == ["crown", "sub_ba"]


@pytest.mark.parametrize("env", [{}, {"SPOCK_EMU_REVERSE": "1"}, {"SPOCK_EMU_REVERSE_BLOCKS": "1"}, {"SPOCK_AMD_ILV_CHUNK": "3"}])
def test_emulated_two_bottom_stages_in_one_launch(S, O, emu, monkeypatch, env):
    """LBA2 (spock_ilevel.hpp): LB of stage N-3, LBA of stage N-2 and LA of stage N-3 as one launch -- x and q of the children,
    c_i / e_i and the parent's [y; s_ch; tau_ch] stay in LDS rows / registers.  Same functions, same order per entry: the iterates
    must be BITWISE those of the three-launch sequence (SPOCK_AMD_ILV_FUSE2=0), in either thread / workgroup order, with chunks
    that leave a ragged last workgroup, for d = 2 and d = 3, SILENT and LOG mode (the L'xi_2 rows), and match the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = [(S.get_server_heat_specs(8, 5, 2, nu=2), {"split_stage": 2}, 70), (S.get_server_heat_specs(6, 5, 2), {"split_stage": 3}, 3),
             (S.get_random_linear_specs(5, 3, 2, d=3), {"split_stage": 2}, 3)]
    for specs, kw, B in cases:
        flat = S.flatten_problem(*specs)
        x0 = 0.1 * (1 + 0.05 * np.arange(B))[:, None] * np.cos(np.arange(flat.nx))[None, :]
        out = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("SPOCK_AMD_ILV_FUSE2", fuse)
            m = S.build_model(*specs, lib_path=emu, batch=B, layout="interleaved", max_iter=30, **kw)
            names = [n for n, _, _ in m.profile_sequence(1)]
            assert ("lba2" in names) == (fuse == "1") and ("lba" in names) == (fuse == "0")
            m.reset()
            res = []
            for verbose in (S.SILENT, S.LOG):
                m.reset()
                rs = S.solve_model(m, x0, tol=1e-6, verbose=verbose)
                res.append((m.solver_state.z.copy(), m.solver_state.v.copy(), [r["iterations"] for r in rs], m.history(0)["xi1"].copy()))
            out[fuse] = res
            if fuse == "1":
                for b in (0, B - 1):
                    o = O.Oracle(flat, O.CP); ro = o.solve(x0[b], tol=1e-6, max_iter=30, verbose=O.LOG)
                    assert rs[b]["iterations"] == ro["iterations"]
                    assert relerr(m.solver_state.z.reshape(B, -1)[b], o.z) <= 1e-9 and relerr(m.solver_state.v.reshape(B, -1)[b], o.v) <= 1e-9
            m.close()
        for (z1, v1, it1, h1), (z0, v0, it0, h0) in zip(out["1"], out["0"]):
            assert it1 == it0 and np.array_equal(z1, z0) and np.array_equal(v1, v0) and np.array_equal(h1, h0)


def test_emulated_interleaved_equals_instance_major_and_mpc(S, O, emu):
    """Same problem, both layouts: identical iteration counts and iterates to rounding; the closed loop on the device works on
    either layout (spock_mpc_simulate reads u_0 with the layout's stride)."""
    specs = S.get_server_heat_specs(6, 5, 2)
    B = 5
    x0 = 0.1 * (1 + 0.2 * np.arange(B))[:, None] * np.ones((B, 5))
    w = np.random.default_rng(1).integers(0, 2, size=(B, 4))
    out = {}
    for lay in ("instance", "interleaved"):
        m = S.build_model(*specs, lib_path=emu, batch=B, layout=lay)
        assert m.sizes.interleaved == (1 if lay == "interleaved" else 0)
        xt, ut, res = S.mpc_simulation(m, x0, w, tol=1e-3)
        out[lay] = (xt, ut, [[r["iterations"] for r in rr] for rr in res], m.solver_state.z.copy())
        m.close()
    assert out["instance"][2] == out["interleaved"][2]
    assert np.max(np.abs(out["instance"][0] - out["interleaved"][0])) <= 1e-10
    assert np.max(np.abs(out["instance"][1] - out["interleaved"][1])) <= 1e-10
    assert relerr(out["interleaved"][3], out["instance"][3]) <= 1e-10


def test_interleaved_layout_falls_back_or_refuses(S, emu):
    specs = S.get_server_heat_specs(5, 5, 2)
    # SP has no element-major kernels: chosen automatically it falls back, asked for explicitly it is refused
    m = S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, batch=1024)
    assert m.sizes.interleaved == 0
    m.close()
    with pytest.raises(S.SpockError):
        S.build_model(*specs, S.SolverOptions(S.SP, None), lib_path=emu, batch=70, layout="interleaved")
    # run-time shapes (no compile-time variant) likewise
    specs2 = S.get_server_heat_specs(4, 4, 3, p_ref=[0.2, 0.3, 0.5])
    m = S.build_model(*specs2, lib_path=emu, batch=1024)
    assert m.sizes.interleaved == 0
    m.close()
    m = S.build_model(*specs, lib_path=emu, batch=1024)    # CP, compile-time shape, >= 320 instances: element-major by default
    assert m.sizes.interleaved == 1
    m.close()
    m = S.build_model(*specs, lib_path=emu, batch=64)      # fewer: the instance-major kernels (faster there)
    assert m.sizes.interleaved == 0
    m.close()
