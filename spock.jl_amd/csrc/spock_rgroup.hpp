// spock_rgroup.hpp -- ROLE-LANE subtree kernel of the single-instance (latency) regime: MODE_SUB_BA_R does what MODE_SUB_BA of
// spock_group.hpp does -- forward sweep of a subtree, phase B, phase A of the next iteration, backward sweep (cp.jl:205-228 with
// z <- zbar at lambda = 1, cp.jl:34-52) -- but splits the NODE-LOCAL phases of a node over role lanes.
//
// Why: in-kernel stamps of SUB_BA on MI355X (one instance, 4096 leaves, subtrees of 63 nodes): the 12 sweep levels take 0.35-0.5 us
// each, the node-local phases 12.1 us (phase B) + 4.7 us (next phase A) -- one thread per node works through ~3000 dependent fp64
// instructions on one SIMD while the other three idle.  The element-major kernels already split a node over ROLES X / R / C_c / L_c
// (spock_ilevel.hpp) that exchange a few partial results through LDS rows; here the same role functions run with lane = NODE (the
// <= 64 non-leaf nodes of the group) and wave = role, on the plain instance-major vectors:
//   1. forward sweep, one thread per node (fwd_node; wave 0), x_i / u_i / x_leaf -> zbar
//   2. wave 0 hands x_i, u_i, x_leaf to the role lanes through LDS rows (the roles' "given" mode: nothing is recomputed)
//   3. ONE role step for all nodes of the group at once: X (box block v7, residuals, next phase A on (x_i, u_i) -> c_i, e_i; at the
//      bottom level also the backward step with the leaves' terms -> q_i, d_i), R (risk block; bottom: S2), C_c (cone block of child c),
//      L_c (leaf child c) -- LBA's roles for the bottom level, LB's for the levels above, chosen per lane
//   4. S2 of the other levels: [y; s_ch; tau_ch] with the children's s entries from their R lanes through LDS (children in other
//      workgroups -- the crown's extra workgroup, last level: recomputed, risk_v2_only, as MODE_SUB_BA does)
//   5. backward sweep over the levels above the bottom one, one thread per node (bwd_node); q_root -> qroot, d_i -> dbuf
// The crown's extra workgroup (g == nsub) runs 2-4 on the nodes above the split stage and leaves c_i, e_i in cbuf for MODE_CROWN.
// Interfaces (dbuf, qroot, cbuf, zbar's y / s / tau, the third z buffer) are those of MODE_SUB_BA: the kernels are interchangeable
// launch by launch.  Sums over the children are formed per role and added in child order (as in the element-major kernels): the
// iterates agree with MODE_SUB_BA to rounding, not bitwise.
#pragma once
#include "spock_ilevel.hpp"

namespace spock {

template <class P>
struct RState {
    TState<P> s;
    LvRole<P> lr;
    real mx[4];
    ProxScal ps;
    real vecp[P::MAXM];   // role R, levels above the bottom one: [y; s_ch; tau_ch] of (-gamma) L'vbar + zbar
};

// X: each(f) / barrier() / phased(f) + sync() (spock_ilevel.hpp) / lds() / stash3, reduce3, red3 (spock_group.hpp)
template <class P, bool WX, class X>
SPK_DEV void rgroup_body(X &x, const KArgs &a, int inst, int g) {
    const Dims &dm = a.dm;
    const P p(dm);
    Ctl &ctl = a.ctl[inst];
    if (ctl.done) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu;
    const MatPtr M = (MatPtr)(a.mats + (long)inst * a.mats_stride);
    const real *z = a.z_in + (long)inst * a.stride_z, *v = a.v_in + (long)inst * a.stride_v;
    real *zo = a.z_out + (long)inst * a.stride_z, *vo = a.v_out + (long)inst * a.stride_v, *zn = a.z_next + (long)inst * a.stride_z;
    real *db = a.dbuf + (long)inst * dm.nn * nu;
    const bool crown_blk = (g == dm.nsub);
    const int root_stage = crown_blk ? 0 : dm.t0, root_pos = crown_blk ? 0 : g;
    const int stages = crown_blk ? dm.t0 + 1 : dm.sub_stages;
    const int nl_grp = igroup_nonleaf(d, stages);
    const int n_groups = dm.nsub + 1;
    double *partials = a.partials + (long)inst * n_groups * kPartialStride;
    const int rows = lba_exchange_rows(nx, nu, d, WX);
    const int crow = (2 + (WX ? 1 : 0)) * nxu + 1, oX = 2 * nxu, rL = d * crow;
    real *const exs = x.lds(), *const G = exs + 64 * nx, *const exr = G + (long)(nxu + d * nx) * 64, *const sown = exr + (long)rows * 64;
    const MatPtr Bm = M + dm.mB, SWb = M + dm.mSweep + (long)(dm.N - 2) * dm.sw_stride;   // (W, V of the bottom stage: role L_c)

    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        init_geometry(p, dm, tid & 63, root_stage, root_pos, stages, crown_blk, st.s);
        st.mx[0] = st.mx[1] = st.mx[2] = st.mx[3] = 0.;
        ProxScal ps;
        ps.sigma = ctl.sigma; ps.inv_sigma = 1. / ctl.sigma; ps.inv_sigma_neg = -1. / ctl.sigma;
        ps.inv_gamma_neg = -1. / ctl.gamma; ps.gamma = ctl.gamma; ps.want_xi = WX;
        st.ps = ps;
    });

    // ---- 1. x_i, u_i (and the leaves' x) of the group's nodes
    if (!crown_blk) {
        const int root_node = dm.first[root_stage] + root_pos;
        const real *xroot = zo + dm.zx + (long)root_node * nx;
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            TState<P> &s = st.s;
            if (tid < 64 && s.active) { SPK_UNROLL for (int k = 0; k < nu; ++k) s.dd[k] = db[s.node * nu + k]; }
        });
        for (int l = 0; l <= stages - 2; ++l) {
            const MatPtr SW = M + dm.mSweep + (long)(root_stage + l) * dm.sw_stride;
            x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
                if (tid < 64 && st.s.active && st.s.lvl == l) fwd_node(p, dm, SW, Bm, tid, exs, xroot, st.s);
            });
            x.barrier();
        }
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA { if (tid < 64 && st.s.active) fwd_store(p, dm, zo, st.s); });
    } else {
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            TState<P> &s = st.s;
            if (tid < 64 && s.active) {
                SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = zo[dm.zx + (long)s.node * nx + k];
                SPK_UNROLL for (int k = 0; k < nu; ++k) s.ub[k] = zo[dm.zu + (long)s.node * nu + k];
            }
        });
    }
    // ---- 2. hand them to the role lanes: rows [x_i | u_i | x_leaf of child 0 .. d-1], one column per node
    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        TState<P> &s = st.s;
        if (tid >= 64 || !s.active) return;
        const LVec gv = LVec{G + tid};
        SPK_UNROLL for (int k = 0; k < nx; ++k) gv[k] = s.xb[k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) gv[nx + k] = s.ub[k];
        if (s.kids_leaf) {
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                SPK_UNROLL for (int k = 0; k < nx; ++k) gv[nxu + c * nx + k] = s.xl[c][k];
            }
        }
    });
    x.barrier();
    // ---- 3. the role step: wave = role, lane = node
    x.phased([&](int tid, RState<P> &st, bool pre, bool post) SPK_ILAMBDA {
        TState<P> &s = st.s;
        const int lane = tid & 63, role = spk_uniform(tid >> 6);
        const bool act = s.active, bot = s.kids_leaf;
        const ProxScal &ps = st.ps;
        const LVec ex = LVec{exr + lane}, gv = LVec{G + lane};
        const int node = s.node;
        if (role == 0) {
            if (act && pre) lv_role_x<P, WX>(p, dm, SWb, Bm, z, v, zo, vo, (const real *)db, node, false, ps, st.mx, st.lr, gv, LVec{nullptr}, gv + nx);
            x.sync();
            if (act && post) {
                lv_finish_x_res<P, WX>(p, ps, ex, crow, oX, st.mx, st.lr);
                // c_i, e_i (bottom level: q_i, d_i with the leaves' terms) through the node's own given rows into the sweep thread's state
                if (bot) lv_finish_x_next<P, true, 1>(p, dm, M, ps, ex, crow, rL, node, s.stage, db, db, st.lr, gv);
                else lv_finish_x_next<P, false, 1>(p, dm, M, ps, ex, crow, rL, node, s.stage, db, db, st.lr, gv);
                SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = gv[k];
                SPK_UNROLL for (int k = 0; k < nu; ++k) s.ub[k] = gv[nx + k];
            }
        } else if (role == 1) {
            if (act && pre) lv_role_r<P, WX, true>(p, dm, M, z, v, zo, vo, node, ps, st.mx, st.lr);
            x.sync();
            if (act && post) {
                sown[lane] = st.lr.sown;
                if (bot) lv_finish_r(p, dm, M, ex, crow, rL, node, zn, st.lr);
                else {
                    SPK_UNROLL for (int k = 0; k < ny; ++k) st.vecp[k] = st.lr.vecy[k];
                    SPK_UNROLL for (int c = 0; c < d; ++c) st.vecp[ny + d + c] = ex[c * crow + crow - 1];
                }
            }
        } else if (role < 2 + d) {
            if (act && pre) lv_role_c<P, WX, true>(p, dm, M, SWb, z, v, zo, vo, (const real *)db, node, role - 2, ps, st.mx, ex, (role - 2) * crow, gv, gv + nx);
            x.sync();
        } else {
            if (act && bot && pre)
                lv_role_l<P, WX>(p, dm, M, SWb, Bm, z, v, zo, vo, (const real *)db, node, role - 2 - d, ps, st.mx, ex, rL, LVec{nullptr},
                                 gv + (nxu + (role - 2 - d) * nx));
            x.sync();
        }
    });
    x.barrier();
    // ---- 4. S2 of the levels above the bottom one (implicit_l.jl:708-726)
    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        TState<P> &s = st.s;
        if ((tid >> 6) != 1 || !s.active) return;
        if (s.node == 0) zn[dm.zs] = st.lr.sown - st.ps.gamma;   // s_1: (z - gamma L'v)[s_1] - gamma  (prox_f!, implicit_l.jl:741)
        if (s.kids_leaf) return;
        const int lane = tid & 63;
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            if (s.kids_gleaf) {   // the children sit in other workgroups of this launch (crown's workgroup, subtree roots): recomputed
                const int j = s.c0 + c;
                st.vecp[ny + c] = -st.ps.gamma * risk_v2_only(p, dm, M, z, v, zo, j, st.ps) + zo[dm.zs + j];
            } else st.vecp[ny + c] = sown[lane * d + 1 + c];
        }
        a_finish_s2(p, dm, M, zn, s, st.vecp);
    });
    // ---- 5. crown's workgroup: c_i, e_i -> cbuf;  subtrees: backward sweep above the bottom level, q_root -> qroot, d_i -> dbuf
    if (crown_blk) {
        real *cb = a.cbuf + (long)inst * igroup_nonleaf(d, dm.t0 + 1) * nxu;
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            TState<P> &s = st.s;
            if (tid >= 64 || !s.active) return;
            SPK_UNROLL for (int k = 0; k < nx; ++k) cb[tid * nx + k] = s.xb[k];
            SPK_UNROLL for (int k = 0; k < nu; ++k) cb[nl_grp * nx + tid * nu + k] = s.ub[k];
        });
    } else {
        real *q_out = a.qroot + ((long)inst * dm.nsub + g) * dm.nx;
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            TState<P> &s = st.s;
            if (tid >= 64 || !s.active || !s.kids_leaf) return;
            SPK_UNROLL for (int k = 0; k < nu; ++k) s.dd[k] = s.ub[k];
            if (s.lvl > 0) { SPK_UNROLL for (int k = 0; k < nx; ++k) exs[tid * nx + k] = s.xb[k]; }
            else { SPK_UNROLL for (int k = 0; k < nx; ++k) q_out[k] = s.xb[k]; }
        });
        x.barrier();
        for (int l = stages - 3; l >= 0; --l) {
            const MatPtr SW = M + dm.mSweep + (long)(root_stage + l) * dm.sw_stride;
            x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
                if (tid < 64 && st.s.active && st.s.lvl == l) bwd_node(p, dm, SW, tid, exs, (const real *)nullptr, q_out, st.s);
            });
            x.barrier();
        }
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            TState<P> &s = st.s;
            if (tid < 64 && s.active) { SPK_UNROLL for (int k = 0; k < nu; ++k) db[s.node * nu + k] = s.dd[k]; }
        });
    }
    // ---- residual maxima of this workgroup
    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        real m3[3];
        for (int k = 0; k < 3; ++k) m3[k] = (st.mx[3] != st.mx[3]) ? st.mx[3] : st.mx[k];
        x.stash3(tid, m3);
    });
    x.reduce3();
    x.each([&](int tid, RState<P> &) SPK_ILAMBDA { if (tid < 3) partials[(long)g * kPartialStride + tid] = x.red3(tid); });
}

}  // namespace spock
