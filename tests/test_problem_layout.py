"""Problem-data layer: the uniform tree generator, the z / v layout and the benchmark dimensions
(scenario_tree.jl:61-109, implicit_l.jl:5-44, cpock.jl:51-60)."""
import numpy as np
import pytest


def test_tree_generator(S):
    t = S.generate_scenario_tree_uniform_branching_factor_v2(3, 2, 2, 1)
    assert (t.n, t.n_non_leaf_nodes, t.n_leaf_nodes) == (7, 3, 4)
    assert t.child_mapping == {0: [1, 2], 1: [3, 4], 2: [5, 6]}
    assert t.anc.tolist() == [-1, 0, 0, 1, 1, 2, 2]
    # 1-based w = (j % d) + 1 -> nodes 2..7 have events 1,2,1,2,1,2 -> 0-based 0,1,0,1,0,1
    assert t.w.tolist() == [-1, 0, 1, 0, 1, 0, 1]
    t3 = S.generate_scenario_tree_uniform_branching_factor_v2(3, 3, 1, 1)
    assert t3.w.tolist()[1:4] == [2, 0, 1]          # 1-based nodes 2,3,4 -> w = 3,1,2
    assert t3.min_index_per_timestep.tolist() == [0, 1, 4]
    with pytest.raises(ValueError):
        S.generate_scenario_tree_uniform_branching_factor_v2(3, 1, 1, 1)


@pytest.mark.parametrize("N,d,nx,nu,n,leaves,nz,nv", [
    (3, 2, 2, 1, 7, 4, 45, 81),
    (7, 2, 5, 5, 127, 64, 1518, 3288),
    (10, 2, 5, 5, 1023, 512, 12270, 26584),
    (13, 2, 5, 2, 8191, 4096, 86001, 176097),
    (13, 2, 5, 5, 8191, 4096, 98286, 212952),
    (9, 3, 20, 20, 9841, 6561, 305061, 846282),
    (10, 2, 20, 20, 1023, 512, 35280, 87934),
    (15, 2, 64, 16, 32767, 16384, 2506664, 6225670),
])
def test_dimension_table(S, N, d, nx, nu, n, leaves, nz, nv):
    tree = S.generate_scenario_tree_uniform_branching_factor_v2(N, d, nx, nu)
    rms = S.get_uniform_rms_avar_v2(np.full(d, 1.0 / d), 0.95, d, N)
    flat = S.FlatProblem(N=N, d=d, nx=nx, nu=nu, ny=2 * d + 1, anc=tree.anc, w=tree.w, A=np.zeros(1), B=np.zeros(1),
                         nQ=1, nR=1, nQN=1, Q=np.zeros(1), R=np.zeros(1), QN=np.zeros(1), E=rms[0].E, F=rms[0].F, b=rms[0].b,
                         cone_kind=np.zeros(2, np.int32), cone_dim=np.zeros(2, np.int32), x_min=-1, x_max=1, u_min=-1, u_max=1)
    assert (flat.n, flat.nl, flat.nz, flat.nv) == (n, leaves, nz, nv)
    assert flat.v_offsets()["end"] == nv


def test_server_heat_data(S):
    tree, cost, dyn, rms, con = S.get_server_heat_specs(7, 5, 2)
    A = dyn.A
    assert np.allclose(np.diag(A[0]), 1.0)                                  # event 1: 1 + (...)*0/d
    assert np.allclose(np.diag(A[1]), [1 + (1 + j / 5) * 0.5 for j in range(5)])
    assert A[0][0, 1] == A[0][1, 0] == 0.01 and A[0][0, 2] == 0.0
    assert np.array_equal(dyn.B[0], np.eye(5))
    rm = rms[0]
    assert np.array_equal(rm.b, [0.3, 0.7, 0, 0, 1.0])
    assert np.array_equal(rm.E, np.vstack([0.95 * np.eye(2), -np.eye(2), np.ones((1, 2))]))
    assert (con.x_min, con.x_max, con.u_min, con.u_max) == (-1.0, 1.0, -1.5, 1.5)
    flat = S.flatten_problem(tree, cost, dyn, rms, con)
    assert (flat.nQ, flat.nR, flat.nQN) == (1, 1, 1)
    # per-node costs that are all equal are recognised as shared
    cost2 = S.Cost([np.eye(5) * 0.1 for _ in range(tree.n - 1)], [np.eye(5) for _ in range(tree.n - 1)],
                   [np.eye(5) * 0.1 for _ in range(tree.n_leaf_nodes)])
    assert S.flatten_problem(tree, cost2, dyn, rms, con).nQ == 1
    cost3 = S.Cost([np.eye(5) * (0.1 + 0.001 * k) for k in range(tree.n - 1)], cost2.R, cost2.QN)
    assert S.flatten_problem(tree, cost3, dyn, rms, con).nQ == tree.n - 1
