// spock_rgroup.hpp -- ROLE-LANE subtree kernel of the single-instance (latency) regime: MODE_SUB_BA_R does what MODE_SUB_BA of
// spock_group.hpp does -- forward sweep of a subtree, phase B, phase A of the next iteration, backward sweep (cp.jl:205-228 with
// z <- zbar at lambda = 1, cp.jl:34-52) -- but splits the NODE-LOCAL phases of a node over role lanes.
//
// Why: in-kernel stamps of SUB_BA on MI355X (one instance, 4096 leaves, subtrees of 63 nodes): the 12 sweep levels take 0.35-0.5 us
// each, the node-local phases 12.1 us (phase B) + 4.7 us (next phase A) -- one thread per node works through ~3000 dependent fp64
// instructions on one SIMD while the other three idle.  The element-major kernels already split a node over ROLES X / R / C_c / L_c
// (spock_ilevel.hpp) that exchange a few partial results through LDS rows; here the same role functions run with lane = NODE (the
// <= 64 non-leaf nodes of the group) and wave = role, on the plain instance-major vectors:
//   1. forward sweep: the register-resident lane-team sweep (fwd_team_reg: T = 8 lanes per node, rows in registers), x_i / u_i /
//      x_leaf -> zbar
//   2. ... and transposed into LDS rows, one column per node, for the role lanes (the roles' "given" mode: nothing is recomputed)
//   3. ONE role step for all nodes of the group at once: X (box block v7, residuals, next phase A on (x_i, u_i) -> c_i, e_i; at the
//      bottom level also the backward step with the leaves' terms -> q_i, d_i), R (risk block; bottom: S2), C_c (cone block of child c),
//      L_c (leaf child c) -- LBA's roles for the bottom level, LB's for the levels above, chosen per lane
//   4. S2 of the other levels: [y; s_ch; tau_ch] with the children's s entries from their R lanes through LDS (children in other
//      workgroups -- the crown's extra workgroup, last level: recomputed, risk_v2_only, as MODE_SUB_BA does)
//   5. backward sweep over the levels above the bottom one (bwd_team_reg); q_root -> qroot, d_i -> dbuf
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
    // (the control block was written by the crown kernel, possibly on another XCD: its load is a cold miss.  The branch on it comes after
    // the first batch of loads below has been issued, so that all of them share one round trip)
    const int done_flag = ctl.done;
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
    // LDS: [sweep vectors XQ 64 nx | ED 64 nu | UL 64 nu | XL 64 d nx][given rows (nx + nu + d nx) 64][role exchange rows 64][s entries 64]
    TeamLds tl;
    tl.XQ = x.lds(); tl.ED = tl.XQ + 64 * nx; tl.UL = tl.ED + 64 * nu; tl.XL = tl.UL + 64 * nu;
    real *const G = tl.XL + 64 * d * nx, *const exr = G + (long)(nxu + d * nx) * 64, *const sown = exr + (long)rows * 64;
    real *const lmat = sown + 64;   // the sweep blocks [W | V | ABK | K] of this group's stages, staged for the backward sweep's rows
    const int T = 8, nthr = x.nthreads();   // register-resident lane-team sweep (bwd_team_reg / fwd_team_reg): T lanes per node, T * nodes <= threads
    auto level_geo = [&](int l, int &lfirst, int &cnt, long &gfirst) SPK_ILAMBDA {
        cnt = 1; lfirst = 0;
        for (int k = 0; k < l; ++k) { lfirst += cnt; cnt *= d; }
        gfirst = dm.first[root_stage + l] + (long)root_pos * cnt;
    };
    const MatPtr Bm = M + dm.mB, SWb = M + dm.mSweep + (long)(dm.N - 2) * dm.sw_stride;   // (W, V of the bottom stage: role L_c)

    // (debug: wall-clock stamps of workgroup 0 at the phase boundaries, SPOCK_AMD_DEBUG_FLAGS & F_DBG_STAMPS; tools/explore_stamps.py)
    int stamp_idx = 0;
    auto stamp = [&]() SPK_ILAMBDA {
        if ((a.flags & F_DBG_STAMPS) && a.dbg && inst == 0 && g == 0 && stamp_idx < 64)
            x.each([&](int tid, RState<P> &) SPK_ILAMBDA { if (tid == 0) x.clocks(a.dbg + 2 * 128 + 2 * stamp_idx); });
        ++stamp_idx;
    };
    stamp();   // 0: entry
    x.max3_init();   // (LDS accumulators of the residual maxima: zeroed here, ordered by the barriers below)
    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        init_geometry(p, dm, tid & 63, root_stage, root_pos, stages, crown_blk, st.s);
        st.mx[0] = st.mx[1] = st.mx[2] = st.mx[3] = 0.;
        ProxScal ps;
        ps.sigma = ctl.sigma; ps.inv_sigma = 1. / ctl.sigma; ps.inv_sigma_neg = -1. / ctl.sigma;
        ps.inv_gamma_neg = -1. / ctl.gamma; ps.gamma = ctl.gamma; ps.want_xi = WX;
        st.ps = ps;
    });

    // ---- 1. x_i, u_i (and the leaves' x) of the group's nodes -> LDS vectors (and zbar); 2. the role lanes' given rows
    if (!crown_blk) {
        const int root_node = dm.first[root_stage] + root_pos;
        const real *xroot = zo + dm.zx + (long)root_node * nx;
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
            team_reg_init(p, dm, M, root_stage, stages, T, tid, st.s, 2);   // this thread's rows of ABK, B, K
            if (tid < nx) tl.XQ[tid] = xroot[tid];
            for (int l = 0; l <= stages - 2; ++l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                for (int k = tid; k < cnt * nu; k += nthr) tl.ED[lfirst * nu + k] = db[gfirst * nu + k];
            }
            const MatPtr src = M + dm.mSweep + (long)root_stage * dm.sw_stride;
            for (int k = tid; k < (stages - 1) * dm.sw_stride; k += nthr) lmat[k] = src[k];
        });
        if (done_flag) return;
        x.barrier();
        stamp();   // 1: d_i, x_root in LDS
        for (int l = 0; l <= stages - 2; ++l) {
            x.each([&](int, RState<P> &st) SPK_ILAMBDA { if (st.s.tn_li >= 0 && st.s.tn_lvl == l) fwd_team_reg(p, stages, tl, st.s); });
            x.barrier();
        }
        stamp();   // 2: forward sweep
        // x_i, u_i, x_leaf: coalesced level chunks -> zbar, and transposed into the given rows [x_i | u_i | x_leaf c], one column per node
        x.each([&](int tid, RState<P> &) SPK_ILAMBDA {
            // (a store to global memory is waited for at the next barrier: x_i goes to zbar from role X's lanes inside the role step --
            // XQ is overwritten by c_i / q_i later --, u_i and x_leaf from UL / XL at the end of the kernel, where nothing waits for them)
            int lfirst, cnt; long gfirst;
            level_geo(stages - 2, lfirst, cnt, gfirst);
            for (int k = tid; k < nl_grp * nx; k += nthr) { const int li = k / nx, e = k - li * nx; G[e * 64 + li] = tl.XQ[k]; }
            for (int k = tid; k < nl_grp * nu; k += nthr) { const int li = k / nu, e = k - li * nu; G[(nx + e) * 64 + li] = tl.UL[k]; }
            for (int k = tid; k < cnt * d * nx; k += nthr) {   // leaf (pos d + c) entry e -> row nxu + c nx + e of its parent's column
                const int lf = k / nx, e = k - lf * nx, pos = lf / d, c = lf - pos * d;
                G[(nxu + c * nx + e) * 64 + lfirst + pos] = tl.XL[k];
            }
        });
    } else {
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {   // the crown's nodes are the global nodes 0 .. nl_grp - 1
            for (int k = tid; k < nl_grp * nx; k += nthr) { const int li = k / nx, e = k - li * nx; G[e * 64 + li] = zo[dm.zx + k]; }
            for (int k = tid; k < nl_grp * nu; k += nthr) { const int li = k / nu, e = k - li * nu; G[(nx + e) * 64 + li] = zo[dm.zu + k]; }
        });
        if (done_flag) return;
    }
    x.barrier();
    stamp();   // 3: zbar stored, x / u handed to the role lanes
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
                SPK_UNROLL for (int k = 0; k < nx; ++k) tl.XQ[lane * nx + k] = gv[k];
                SPK_UNROLL for (int k = 0; k < nu; ++k) tl.ED[lane * nu + k] = gv[nx + k];
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
            if (role < 2 + 2 * d && act && bot && pre)   // (the waves past the last role only take part in the sweeps)
                lv_role_l<P, WX>(p, dm, M, SWb, Bm, z, v, zo, vo, (const real *)db, node, role - 2 - d, ps, st.mx, ex, rL, LVec{nullptr},
                                 gv + (nxu + (role - 2 - d) * nx));
            x.sync();
        }
    });
    // residual maxima of this workgroup: complete once the roles' finishing parts have run; folded into LDS in front of the barrier
    x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {
        real m3[3];
        for (int k = 0; k < 3; ++k) m3[k] = (st.mx[3] != st.mx[3]) ? st.mx[3] : st.mx[k];
        x.max3_add(tid, m3);
    });
    x.barrier();
    x.each([&](int tid, RState<P> &) SPK_ILAMBDA { if (tid < 3) partials[(long)g * kPartialStride + tid] = x.max3_get(tid); });
    stamp();   // 4: role step
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
    stamp();   // 5: S2 of the upper levels
    // ---- 5. crown's workgroup: c_i, e_i -> cbuf;  subtrees: backward sweep above the bottom level, q_root -> qroot, d_i -> dbuf
    if (crown_blk) {
        real *cb = a.cbuf + (long)inst * igroup_nonleaf(d, dm.t0 + 1) * nxu;   // layout [c of every crown node | e of every crown node]
        x.each([&](int tid, RState<P> &) SPK_ILAMBDA {
            for (int k = tid; k < nl_grp * nx; k += nthr) cb[k] = tl.XQ[k];
            for (int k = tid; k < nl_grp * nu; k += nthr) cb[nl_grp * nx + k] = tl.ED[k];
        });
    } else {
        // XQ / ED hold c_i / e_i of the upper levels and q_i / d_i of the bottom level: the sweep continues above it
        x.each([&](int tid, RState<P> &st) SPK_ILAMBDA {   // this thread's rows of W, V (team_reg_init's part 1), from the staged blocks
            TState<P> &s = st.s;
            if (s.tn_li < 0) return;
            const real *SW = lmat + (long)s.tn_lvl * dm.sw_stride;
            const int r = s.tn_r;
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const real *W = SW + dm.sw_oW + c * nx * nx, *V = SW + dm.sw_oV + c * nu * nx;
                SPK_UNROLL for (int k = 0; k < nx; ++k) {
                    s.tW[c][k] = (r < nx) ? W[r + k * nx] : (real)0.;
                    s.tV[c][k] = (r < nu) ? V[r + k * nu] : (real)0.;
                }
            }
        });
        for (int l = stages - 3; l >= 0; --l) {
            x.each([&](int, RState<P> &st) SPK_ILAMBDA { if (st.s.tn_li >= 0 && st.s.tn_lvl == l) bwd_team_reg(p, stages, tl, st.s); });
            x.barrier();
        }
        real *q_out = a.qroot + ((long)inst * dm.nsub + g) * dm.nx;
        x.each([&](int tid, RState<P> &) SPK_ILAMBDA {
            if (tid < nx) q_out[tid] = tl.XQ[tid];
            for (int l = 0; l <= stages - 2; ++l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                for (int k = tid; k < cnt * nu; k += nthr) { db[gfirst * nu + k] = tl.ED[lfirst * nu + k]; zo[dm.zu + gfirst * nu + k] = tl.UL[lfirst * nu + k]; }
            }
            int lfirst, cnt; long gfirst;
            level_geo(stages - 2, lfirst, cnt, gfirst);
            const long gl_first = dm.first[root_stage + stages - 1] + (long)root_pos * cnt * d;
            for (int k = tid; k < cnt * d * nx; k += nthr) zo[dm.zx + gl_first * nx + k] = tl.XL[k];
        });
    }
    stamp();   // 6: backward sweep, q_root / d_i stored
    stamp();   // 7: end
}

}  // namespace spock
