// spock_igroup.hpp -- the BATCHED regime of the Chambolle-Pock iteration: element-major ("instance-interleaved") iterate,
// one LANE per problem instance, one WAVE per tree node.
//
// Layout: every vector of the solver state is stored element-major, z[e][Bp], v[e][Bp], d[e][Bp], ... with Bp the batch
// rounded up to a multiple of 64.  A wave works on one TILE of 64 consecutive instances and one tree node at a time: lane l
// is instance 64 tile + l.  Consequences on MI355X:
//   * every global load / store of a wave is 512 contiguous bytes (four full 128-byte lines) -- the instance-major layout
//     gives each lane its own 40- or 56-byte-strided address and leans on the 32 KiB L1 to merge them;
//   * the tree node, and with it every matrix element, offset and branch, is wave-uniform: matrices stay in scalar registers,
//     addresses are scalar base + lane offset;
//   * the S1 sweep exchanges q / x between tree levels through LDS slots [node][k][64 lanes] (2.5 KB per node at nx = 5).
// The node-local arithmetic is the SAME code as the one-thread-per-node kernels (a_local_g, b_local_g, bwd_node, fwd_node of
// spock_device.hpp, instantiated on IVec views), so parity carries over; only the orchestration below is new.
//
// Kernel sequence of one CP iteration (driver: spock_abi.cpp; reference loop: cp.jl:205-228).  The tree is tiled into a
// hierarchy of GROUPS of at most ~4 non-leaf levels (what fits the LDS slots): the bottom groups (tree leaves), layers of mid
// groups, the top group (tree root); one workgroup per group and tile.  With two layers above the bottom one (N = 13):
//   SUB_A + ICROWN_A (once per solve)  phase A + backward sweep of iteration 0 (bottom groups); phase A on the crown's nodes
//   I_UP(mid)     backward sweep over the mid groups (c_i, e_i from cbuf, q of the bottom groups' roots from cq)
//   CROWN (top)   termination test of the previous iteration (one lane = one instance); backward sweep, forward sweep, phase B
//                 and the next phase A over the top group
//   I_DOWN(mid)   forward sweep + phase B top-down, then the next iteration's phase A bottom-up (c_i, e_i to cbuf)
//   SUB_BA        the same over the bottom groups, where phase A is followed by the backward sweep step (q of the root to cq);
//                 on the last level (children = leaves) phase A and the backward step follow phase B of the same node directly
//   FINALIZE      the pending termination test after the last replay
// vbar.v2 of a group's leaves, which the next phase A of their parents needs, is recomputed by the parents' workgroup
// (risk_v2_only): the leaves' own workgroup runs in a later launch.
// A lane whose instance has finished (or is padding) is masked for the whole kernel, so its state freezes exactly like in the
// one-thread-per-node kernels.
#pragma once
#include "spock_device.hpp"

namespace spock {

#ifdef SPOCK_HOST_EMU
#define SPK_ILAMBDA
SPK_DEV int spk_uniform(int x) { return x; }
#else
#define SPK_ILAMBDA __attribute__((always_inline))
SPK_DEV int spk_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }   // wave-uniform by construction: tell the compiler
#endif

// Phase A of the NEXT iteration gathered from phase B's registers on the bottom level (ANext) instead of re-read from memory.
// Measured on MI355X (batch 2048): same kernel time, but the 26 extra live doubles spill (808 B of scratch per lane) and the raw
// HBM traffic goes UP (FETCH + WRITE 9.3 GB against 8.7 GB per launch).  Off by default; -DSPOCK_ILV_FUSE_A builds it.
#ifdef SPOCK_ILV_FUSE_A
constexpr bool kFuseNextA = true;
#else
constexpr bool kFuseNextA = false;
#endif

// role waves of the level-streaming kernels (spock_ilevel.hpp): what a role carries from the part in front of the workgroup's
// barrier to its finishing part behind it
template <class P>
struct LvRole {
    real xi[P::MAXNX], ui[P::MAXNU];                     // X: x_i, u_i of the node
    real dzx[P::MAXNX], dzu[P::MAXNU];                   // X: dz = zbar - z on them
    real nv7[P::MAXNX + P::MAXNU], dv7[P::MAXNX + P::MAXNU], lx7[P::MAXNX + P::MAXNU];   // X: vbar.v7, its dv and xi_2
    real vecy[P::MAXNY];                                 // R: the y part of (-gamma) L'vbar + zbar
    real sown;                                           // R (LBA2): the node's own s entry of its parent's [y; s_ch; tau_ch]
};
template <class P>
struct IState {
    TState<P> s;
    real mx[4];   // xi_1, xi_2, xi maxima and the NaN poison accumulator (spk_resmax)
    bool live;
    LvRole<P> lr;
    ProxScal ps;   // level kernels: the instance's step sizes and their reciprocals, formed once per kernel (three divisions)
};

SPK_DEV int igroup_nonleaf(int d, int stages) {   // non-leaf nodes of a complete d-ary tree with `stages` stages
    int cnt = 0, lvl = 1;
    for (int t = 0; t < stages - 1; ++t) { cnt += lvl; lvl *= d; }
    return cnt;
}

// X (execution policy): each(f) runs f(tid, IState&) for every thread; barrier(); lds() the q / x exchange slots;
// live_flags() 64 ints in LDS; red() [waves][3][64] doubles in LDS; nthreads()
template <class P, int MODE, bool WX, class X>
SPK_DEV void igroup_body(X &x, const KArgs &a, const real *zin_, const real *vin_, real *zout_, real *vout_,
                         real *znext_, int tile, int g) {
    const Dims &dm = a.dm;
    const P p(dm);
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu;
    const long bp = a.bp, t64 = (long)tile * 64;
    const int nW = x.nthreads() / 64;
    const int n_groups = dm.n_bslots + dm.n_crown_groups;   // partial slots: bottom part first, then the crown groups
    const MatPtr M = (MatPtr)a.mats;
    double *partials = a.partials + (long)tile * n_groups * 3 * 64;   // [group][xi1, xi2, xi][lane]

    // ---- control: which lanes (instances) are live?  CROWN / FINALIZE also run the pending termination test, one lane each
    if (MODE == MODE_CROWN || MODE == MODE_FINALIZE) {
        // the workgroups' partial maxima of the previous iteration: every wave reduces a share of the slots (four independent
        // chains of loads in flight per lane), wave 0 combines the waves' results
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            const int lane = tid & 63, wave = tid >> 6;
            double m[3][4] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}};
            int gg = wave;
            for (; gg + 3 * nW < n_groups; gg += 4 * nW)
                for (int u = 0; u < 4; ++u)
                    for (int k = 0; k < 3; ++k) m[k][u] = spk_max_nan(m[k][u], partials[((long)(gg + u * nW) * 3 + k) * 64 + lane]);
            for (; gg < n_groups; gg += nW)
                for (int k = 0; k < 3; ++k) m[k][0] = spk_max_nan(m[k][0], partials[((long)gg * 3 + k) * 64 + lane]);
            for (int k = 0; k < 3; ++k) x.red()[(wave * 3 + k) * 64 + lane] = spk_max_nan(spk_max_nan(m[k][0], m[k][1]), spk_max_nan(m[k][2], m[k][3]));
        });
        x.barrier();
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            if (tid >= 64) return;
            const int lane = tid, inst = tile * 64 + lane;
            bool fin = true;
            if (inst < a.batch) {
                Ctl &ctl = a.ctl[inst];
                const CtlSnap snap = ctl_snapshot(ctl);
                double m[3] = {0., 0., 0.};
                if (snap.pending && !snap.done)
                    for (int w = 0; w < nW; ++w)
                        for (int k = 0; k < 3; ++k) m[k] = spk_max_nan(m[k], x.red()[(w * 3 + k) * 64 + lane]);
                double *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
                fin = ctl_prologue(ctl, snap, m[0], m[1], m[2], hist, a.hist_cap, a.parity, MODE != MODE_FINALIZE);
            }
            x.live_flags()[lane] = fin ? 0 : 1;
        });
        x.barrier();
        if (MODE == MODE_FINALIZE) return;
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA { st.live = x.live_flags()[tid & 63] != 0; });
    } else {
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            const int inst = tile * 64 + (tid & 63);
            st.live = inst < a.batch && !a.ctl[inst].done;
        });
    }
    x.each([&](int, IState<P> &st) SPK_ILAMBDA { st.mx[0] = st.mx[1] = st.mx[2] = st.mx[3] = 0.; });

    // views of this tile's columns: base pointers are wave-uniform, the lane offset is added per thread (vw)
    const real *zin = zin_ + t64, *vin = vin_ + t64;
    real *zout = zout_ + t64, *vout = vout_ + t64, *znext = znext_ ? znext_ + t64 : nullptr;
    real *dbuf = a.dbuf + t64, *cbuf = a.cbuf + t64, *cq = a.cq ? a.cq + t64 : nullptr;
    const long n_cb = (long)igroup_nonleaf(d, dm.t0 + 1);
    auto vw = [&](const real *b, long n_elems, int tid) SPK_ILAMBDA { return b ? ivec_make(b, bp, tid & 63, n_elems) : ivec_null(); };
    auto lvw = [&](int tid) SPK_ILAMBDA { return LVec{x.lds() + (tid & 63)}; };
    const long n_db = (long)dm.nn * nu, n_cbuf = n_cb * nxu;

    // LOG-mode CP needs the combined residual xi (cp.jl:83-86): a property of the solve, compiled in (WX)
    constexpr bool want_xi = WX;
    auto make_ps = [&](const Ctl &ctl) SPK_ILAMBDA {
        ProxScal ps;
        ps.sigma = ctl.sigma; ps.inv_sigma = 1. / ctl.sigma; ps.inv_sigma_neg = -1. / ctl.sigma;
        ps.inv_gamma_neg = -1. / ctl.gamma; ps.gamma = ctl.gamma;
        ps.want_xi = want_xi;
        return ps;
    };
    auto level_geo = [&](int l, int &lfirst, int &cnt) SPK_ILAMBDA {
        cnt = 1; lfirst = 0;
        for (int k = 0; k < l; ++k) { lfirst += cnt; cnt *= d; }
    };
    // residual maxima of this workgroup: per lane over the nodes its wave processed, then over the waves
    auto reduce_maxima = [&](int slot) SPK_ILAMBDA {
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            for (int k = 0; k < 3; ++k) x.red()[((tid >> 6) * 3 + k) * 64 + (tid & 63)] = st.live ? (st.mx[3] != st.mx[3] ? st.mx[3] : st.mx[k]) : 0.;
        });
        x.barrier();
        x.each([&](int tid, IState<P> &) SPK_ILAMBDA {
            for (int idx = tid; idx < 64 * 3; idx += x.nthreads()) {
                const int k = idx >> 6, lane = idx & 63;
                double m = 0.;
                for (int w = 0; w < nW; ++w) m = spk_max_nan(m, x.red()[(w * 3 + k) * 64 + lane]);
                partials[((long)slot * 3 + k) * 64 + lane] = m;
            }
        });
    };

    const int crown_nl = (int)n_cb;   // non-leaf nodes of the crown = global nodes 0 .. crown_nl - 1 (breadth-first numbering)

    // ---- crown prologue (once per solve): phase A of iteration 0 on a chunk of crown nodes, c_i / e_i to cbuf ------------------
    if (MODE == MODE_ICROWN_A) {
        const int stages = dm.t0 + 1;
        const int t_lo = g * a.cchunk, t_hi = (t_lo + a.cchunk < crown_nl) ? t_lo + a.cchunk : crown_nl;
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            if (!st.live) return;
            const int wave = spk_uniform(tid >> 6);
            const real gamma = a.ctl[tile * 64 + (tid & 63)].gamma;
            const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid), cb = vw(cbuf, n_cbuf, tid);
            for (int tau = t_lo + wave; tau < t_hi; tau += nW) {
                TState<P> &s = st.s;
                init_geometry(p, dm, tau, 0, 0, stages, true, s);
                a_local_g(p, dm, M, z, v, zo, s, gamma, 0u, (const real *)nullptr);
                for (int k = 0; k < nx; ++k) cb[tau * nx + k] = s.xb[k];
                for (int k = 0; k < nu; ++k) cb[crown_nl * nx + tau * nu + k] = s.ub[k];
            }
        });
        return;
    }

    // ---- the group this workgroup works on: a complete d-ary piece of the tree rooted at (root_stage, root_pos) ---------------
    // Groups form a hierarchy (Dims::lv_root, spock_host.cpp): the BOTTOM groups (root stage t0, leaves = tree leaves), zero or
    // more layers of MID groups and the TOP group (root = tree root); the leaves of a non-bottom group are the roots of the
    // groups below it.  Inside a group q / x travel through LDS slots, between groups through cq (q, per global node) and zbar (x).
    const bool bottom_grp = (MODE == MODE_SUB_A || MODE == MODE_SUB_BA);
    const int lv = bottom_grp ? dm.n_lv - 1 : (MODE == MODE_CROWN ? 0 : a.grp_level);
    const int root_stage = dm.lv_root[lv], root_pos = g;
    const int stages = bottom_grp ? dm.N - root_stage : dm.lv_root[lv + 1] - root_stage + 1;
    const int root_node = dm.first[root_stage] + root_pos;
    int gl_span = 1;                                   // leaves of this group
    for (int k = 0; k < stages - 1; ++k) gl_span *= d;
    const long gl_first = dm.first[root_stage + stages - 1] + (long)root_pos * gl_span;   // first of them (global node)
    const long n_cqa = (long)(crown_nl + dm.nsub) * nx;                                  // rows of cq: crown nodes + subtree roots

    // d_i of node `node`: bottom groups keep it in dbuf, crown groups in the e-slot of cbuf (e_i -> d_i in place)
    auto d_slot = [&](int tid, int node) SPK_ILAMBDA {
        return bottom_grp ? vw(dbuf, n_db, tid) + (long)node * nu : vw(cbuf, n_cbuf, tid) + ((long)crown_nl * nx + (long)node * nu);
    };
    // backward step of node tau (implicit_l.jl:584-637): q_children from LDS / cq (group leaves) / registers (tree leaves)
    auto bwd_step = [&](int tid, TState<P> &s, int tau, int l) SPK_ILAMBDA {
        const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
        const IVec cqv = vw(cq, n_cqa, tid);
        bwd_node(p, dm, SW, tau, lvw(tid), cqv + gl_first * nx, root_stage > 0 ? cqv + (long)root_node * nx : ivec_null(), s);
        const IVec ds = d_slot(tid, s.node);
        for (int k = 0; k < nu; ++k) ds[k] = s.dd[k];
    };
    // phase A of node tau on the iterate (zi, vi) -> its part of the next zbar.  Crown groups: vbar.v2 of the children in registers
    // (children inside the group: written by this workgroup's phase B; group leaves: recomputed, their workgroup runs later)
    auto a_step = [&](int tid, TState<P> &s, const IVec &zi, const IVec &vi, const IVec &zn, const IVec &zprev, const IVec &vprev,
                      const ProxScal &ps, bool recompute_v2) SPK_ILAMBDA {
        if (bottom_grp || !recompute_v2) {
            a_local_g(p, dm, M, zi, vi, zn, s, ps.gamma, 0u, (const real *)nullptr);
        } else {
            V2Reg<P::MAXD> v2c;
            v2c.base = s.pos * d;
            for (int c = 0; c < d; ++c)
                v2c.v[c] = s.kids_gleaf ? risk_v2_only(p, dm, M, zprev, vprev, zi, s.c0 + c, ps) : (real)vi[dm.o2 + s.c0 + c];
            a_local_g(p, dm, M, zi, vi, zn, s, ps.gamma, 0u, v2c);
        }
    };
    auto store_ce = [&](int tid, const TState<P> &s) SPK_ILAMBDA {
        const IVec cb = vw(cbuf, n_cbuf, tid);
        for (int k = 0; k < nx; ++k) cb[(long)s.node * nx + k] = s.xb[k];
        for (int k = 0; k < nu; ++k) cb[(long)crown_nl * nx + (long)s.node * nu + k] = s.ub[k];
    };

    // ---- UP: backward sweep over a crown group (c_i, e_i from cbuf) ---------------------------------------------------------
    auto up_pass = [&]() SPK_ILAMBDA {
        for (int l = stages - 2; l >= 0; --l) {
            int lfirst, cnt;
            level_geo(l, lfirst, cnt);
            x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
                if (!st.live) return;
                const int wave = spk_uniform(tid >> 6);
                const IVec cb = vw(cbuf, n_cbuf, tid);
                for (int pos = wave; pos < cnt; pos += nW) {
                    const int tau = lfirst + pos;
                    TState<P> &s = st.s;
                    init_geometry(p, dm, tau, root_stage, root_pos, stages, true, s);
                    for (int k = 0; k < nx; ++k) s.xb[k] = cb[(long)s.node * nx + k];
                    for (int k = 0; k < nu; ++k) s.ub[k] = cb[(long)crown_nl * nx + (long)s.node * nu + k];
                    bwd_step(tid, s, tau, l);
                }
            });
            x.barrier();
        }
    };
    // ---- DOWN: forward sweep + phase B, top-down; then phase A of the NEXT iteration bottom-up (bottom groups: + backward sweep)
    auto down_pass = [&]() SPK_ILAMBDA {
        for (int l = 0; l <= stages - 2; ++l) {
            int lfirst, cnt;
            level_geo(l, lfirst, cnt);
            const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride, Bm = M + dm.mB;
            const bool last = (l == stages - 2);
            x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
                if (!st.live) return;
                const int wave = spk_uniform(tid >> 6);
                const Ctl &ctl = a.ctl[tile * 64 + (tid & 63)];
                const ProxScal ps = make_ps(ctl);
                const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid);
                const IVec zo = vw(zout, dm.nz, tid), vo = vw(vout, dm.nv, tid), zn = vw(znext, dm.nz, tid);
                const real *x0 = a.x0 + (long)(tile * 64 + (tid & 63)) * nx;
                for (int pos = wave; pos < cnt; pos += nW) {
                    const int tau = lfirst + pos;
                    TState<P> &s = st.s;
                    init_geometry(p, dm, tau, root_stage, root_pos, stages, !bottom_grp, s);
                    const IVec ds = d_slot(tid, s.node);
                    for (int k = 0; k < nu; ++k) s.dd[k] = ds[k];
                    if (root_stage == 0) fwd_node(p, dm, SW, Bm, tau, lvw(tid), x0, s);
                    else fwd_node(p, dm, SW, Bm, tau, lvw(tid), zo + (dm.zx + (long)root_node * nx), s);
                    fwd_store(p, dm, zo, s);      // x_i, u_i and the x of children that are leaves of the group
                    if (last && bottom_grp && kFuseNextA) {
                        // children are leaves: the next phase A of this node reads only what this phase B has just produced --
                        // gathered from registers (ANext), nothing is read back from memory
                        ANext<P> an;
                        b_local_g<P, 0, 0>(p, dm, M, z, v, zo, vo, s, ps, 0u, st.mx, nullptr, nullptr, nullptr, &an);
                        real xbar[P::MAXNX], ubar[P::MAXNU];
                        for (int k = 0; k < nx; ++k) xbar[k] = -ps.gamma * an.ax[k] + s.xb[k];
                        for (int k = 0; k < nu; ++k) ubar[k] = -ps.gamma * an.au[k] + s.ub[k];
                        for (int c = 0; c < d; ++c)
                            for (int k = 0; k < nx; ++k) s.xl[c][k] = an.xl[c][k];
                        a_finish(p, dm, M, zn, s, xbar, ubar, an.vec);
                        bwd_step(tid, s, tau, l);
                    } else {
                        b_local_g<P, 0, 0>(p, dm, M, z, v, zo, vo, s, ps, 0u, st.mx);
                        if (last) {   // the last level's phase A needs nothing from this workgroup's other nodes: right away
                            a_step(tid, s, zo, vo, zn, z, v, ps, true);
                            if (bottom_grp) bwd_step(tid, s, tau, l); else store_ce(tid, s);
                        }
                    }
                }
            });
            x.barrier();
        }
        for (int l = stages - 3; l >= 0; --l) {
            int lfirst, cnt;
            level_geo(l, lfirst, cnt);
            x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
                if (!st.live) return;
                const int wave = spk_uniform(tid >> 6);
                const ProxScal ps = make_ps(a.ctl[tile * 64 + (tid & 63)]);
                const IVec zo = vw(zout, dm.nz, tid), vo = vw(vout, dm.nv, tid), zn = vw(znext, dm.nz, tid);
                for (int pos = wave; pos < cnt; pos += nW) {
                    const int tau = lfirst + pos;
                    TState<P> &s = st.s;
                    init_geometry(p, dm, tau, root_stage, root_pos, stages, !bottom_grp, s);
                    a_step(tid, s, zo, vo, zn, zo, vo, ps, true);      // (no group leaves on these levels: zprev / vprev unused)
                    if (bottom_grp) bwd_step(tid, s, tau, l); else store_ce(tid, s);
                }
            });
            if (bottom_grp) x.barrier();
        }
        // partial residual maxima: bottom groups in slots 0 .. nsub - 1, crown groups behind them in the order of their launch
        int slot = g;
        if (!bottom_grp) { slot = dm.n_bslots; for (int k = 0; k < lv; ++k) { int c = 1; for (int q = 0; q < dm.lv_root[k]; ++q) c *= d; slot += c; } slot += g; }
        reduce_maxima(slot);
    };

    if (MODE == MODE_I_UP) { up_pass(); return; }
    if (MODE == MODE_CROWN) { up_pass(); down_pass(); return; }
    if (MODE == MODE_I_DOWN || MODE == MODE_SUB_BA) { down_pass(); return; }
    if (MODE == MODE_SUB_A) {   // prologue: phase A of iteration 0 + backward sweep over the bottom groups
        for (int l = stages - 2; l >= 0; --l) {
            int lfirst, cnt;
            level_geo(l, lfirst, cnt);
            x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
                if (!st.live) return;
                const int wave = spk_uniform(tid >> 6);
                const ProxScal ps = make_ps(a.ctl[tile * 64 + (tid & 63)]);
                const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid);
                for (int pos = wave; pos < cnt; pos += nW) {
                    const int tau = lfirst + pos;
                    TState<P> &s = st.s;
                    init_geometry(p, dm, tau, root_stage, root_pos, stages, false, s);
                    a_step(tid, s, z, v, zo, z, v, ps, false);
                    bwd_step(tid, s, tau, l);
                }
            });
            x.barrier();
        }
    }
}

}  // namespace spock
