// spock_ilevel.hpp -- LEVEL-STREAMING kernels of the element-major batched regime (one lane per problem instance): the stages
// t0 .. N-2 of the tree, which hold almost all of its nodes, one launch per stage.
//
// Why: the bottom-group kernels of spock_igroup.hpp keep a whole node (phase B, the next phase A and the backward step) in
// ONE wave: 255 vector registers, two waves per SIMD.  Measured on MI355X (tools/micro/stream_pattern.hip: the node access
// pattern without arithmetic), two waves per SIMD cannot draw more than 3.2-3.5 TB/s from HBM whatever their depth of loads in
// flight; four reach 4.6-4.8, eight 4.9 TB/s.  The kernels here therefore split a node over ROLE WAVES, each with a small live
// set, which hand their few partial results to one another through LDS rows of 64 doubles:
//   X    forward step of the node (u_i, x of the children), box block v7, afterwards the xi_1 residuals on (x_i, u_i) and -- bottom
//        stage -- the next phase A on (x_i, u_i), c_i / e_i and the backward step (q_i to cq, d_i to dbuf)
//   R    risk block (v1, v2); bottom stage: afterwards S2 on [y_i; s_ch; tau_ch] -> next zbar
//   C_c  the cone block of child c (v3 .. v6): its L'dv / L'vbar on (x_i, u_i), tau_c of the next phase A
//   L_c  bottom stage: leaf child c (its x from the forward step, v11 .. v14), its xbar of the next phase A as W_c q_c, V_c q_c
// A workgroup = one wave per role, working through ls_chunk consecutive nodes of the stage with ONE barrier per node (the
// exchange rows are double-buffered).  Nothing else is shared: x_i comes from zbar (stored by the parent's launch), q of the
// children from cq, so there are no sweep slots in LDS and no barriers between tree levels -- a tree level is a kernel boundary.
//   LB   (stages t0 .. N-3, top-down)  forward step + phase B + the (x_i, u_i) half of the next phase A (c_i, e_i parked in cq /
//                                      dbuf)                                      roles X, R, C_0 .. C_{d-1}
//   LBA  (stage N-2)                   forward step + phase B + next phase A + backward step   roles X, R, C_c, L_c
//   LA   (stages N-3 .. t0, bottom-up) backward step from c_i, e_i and the children's q (role AX); the [y; s_ch; tau_ch] half of the
//                                      next phase A with S2 on (zbar, vbar) (role AR)
//   LA0  (prologue, once per solve)    phase A + backward step of iteration 0 on (z, v), one wave per node (a_local_g)
//   LBA2 (stage N-3, default)          LB of stage N-3, LBA of stage N-2 and LA of stage N-3 in ONE launch: a parent, then its children,
//                                      then the parent's backward step; what the three launches handed to one another through HBM
//                                      (x and q of the children, c_i / e_i, LA's re-reads) stays in registers / LDS
// The arithmetic per entry is that of a_local_g / b_local_g / fwd_node / bwd_node (spock_device.hpp), block by block; sums over the
// children that those accumulate into one register are formed here per role and added afterwards (a different, fixed order).
// Reference: L! / L_transpose! / prox_f! / prox_h_conj! (implicit_l.jl:177-951), should_terminate! (cp.jl:54-123).
#pragma once
#include "spock_igroup.hpp"

namespace spock {

// x_i (from zbar; two-stage kernel LBA2: the children's x from the LDS rows `xs` their parent's role X filled) and d_i of a node
template <class P, class ZB, class DB>
SPK_DEV void lv_load_xd(const P &p, const Dims &dm, ZB zo, DB db, int node, real *xi, real *di, LVec xs = LVec{nullptr}) {
    const int nx = p.nx(), nu = p.nu();
    if (xs) { SPK_UNROLL for (int k = 0; k < nx; ++k) xi[k] = xs[k]; }
    else { SPK_UNROLL for (int k = 0; k < nx; ++k) xi[k] = zo[dm.zx + node * nx + k]; }
    SPK_UNROLL for (int k = 0; k < nu; ++k) di[k] = db[node * nu + k];
}
// u_i = K x_i + d_i  (fwd_node)
template <class P>
SPK_DEV void lv_u_of(const P &p, const Dims &dm, MatPtr SW, const real *xi, const real *di, real *ui) {
    const int nx = p.nx(), nu = p.nu();
    const MatPtr Kt = SW + dm.sw_oK;
    SPK_UNROLL for (int r = 0; r < nu; ++r) {
        real a = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += Kt[r + k * nu] * xi[k];
        ui[r] = a + di[r];
    }
}
// x of child slot c = ABK_c x_i + B_c d_i  (fwd_node)
template <class P>
SPK_DEV void lv_child_x(const P &p, const Dims &dm, MatPtr SW, MatPtr Bm, int c, const real *xi, const real *di, real *xc) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    const MatPtr Bw = Bm + slot_event(c, d) * nx * nu;
    const MatPtr Ab = SW + dm.sw_oABK + c * nx * nx;
    SPK_UNROLL for (int r = 0; r < nx; ++r) {
        real a = 0., b2 = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += Ab[r + k * nx] * xi[k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) b2 += Bw[r + k * nx] * di[k];
        xc[r] = a + b2;
    }
}

// ---- role X, in front of the barrier: forward step, box block v7 (implicit_l.jl:275-282, constraints.jl:38-54) ------------------
template <class P, bool WX, class ZI, class VI, class ZB, class VO, class DB>
SPK_DEV void lv_role_x(const P &p, const Dims &dm, MatPtr SW, MatPtr Bm, ZI z, VI v, ZB zo, VO vo, DB db, int node, bool store_children,
                       const ProxScal &ps, real *mx, LvRole<P> &lr, LVec xs = LVec{nullptr}, LVec xd = LVec{nullptr}, LVec gu = LVec{nullptr}) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu, i = node;
    const real sg = ps.sigma, isg = ps.inv_sigma, nis = ps.inv_sigma_neg;
    real di[P::MAXNU];
    // (gu: x_i AND u_i are given -- the forward sweep of the role-lane subtree kernel, spock_rgroup.hpp, has produced and stored them)
    if (gu) {
        SPK_UNROLL for (int k = 0; k < nx; ++k) lr.xi[k] = xs[k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) { lr.ui[k] = gu[k]; di[k] = 0.; }
        // (x_i goes to zbar from here: a store in front of the hand-off's barrier would be waited for there, ~1 us; the barrier that
        // closes the role step waits for this role's loads anyway)
        SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + i * nx + k] = lr.xi[k];
    } else lv_load_xd(p, dm, zo, db, i, lr.xi, di, xs);
    real zx_[P::MAXNX], zu_[P::MAXNU], v7[P::MAXNX + P::MAXNU];
    SPK_UNROLL for (int k = 0; k < nx; ++k) zx_[k] = z[dm.zx + i * nx + k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) zu_[k] = z[dm.zu + i * nu + k];
    SPK_UNROLL for (int k = 0; k < nxu; ++k) v7[k] = v[dm.o7 + i * nxu + k];
    if (!gu) {
        lv_u_of(p, dm, SW, lr.xi, di, lr.ui);
        SPK_UNROLL for (int k = 0; k < nu; ++k) zo[dm.zu + i * nu + k] = lr.ui[k];
    }
    if (store_children) {
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            real xc[P::MAXNX];
            lv_child_x(p, dm, SW, Bm, c, lr.xi, di, xc);
            SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + (i * d + 1 + c) * nx + k] = xc[k];
            if (xd) { SPK_UNROLL for (int k = 0; k < nx; ++k) xd[c * nx + k] = xc[k]; }   // LBA2: handed to the children's roles through LDS
        }
    }
    SPK_UNROLL for (int k = 0; k < nxu; ++k) {
        const bool isx = k < nx;
        const real xu = isx ? lr.xi[k] : lr.ui[k - nx], zi = isx ? zx_[k] : zu_[k - nx];
        const real wv = 2. * xu + -1. * zi, dzv = xu - zi;
        if (isx) lr.dzx[k] = dzv; else lr.dzu[k - nx] = dzv;
        real t = sg * wv + v7[k];
        t = t * isg;
        const real pj = isx ? spk_clamp(t, dm.x_min, dm.x_max) : spk_clamp(t, dm.u_min, dm.u_max);
        const real nvv = sg * (t - pj);
        vo[dm.o7 + i * nxu + k] = nvv;
        const real dv = nvv - v7[k];
        const real x2 = dzv + nis * dv;
        spk_resmax<IVec>(mx[1], x2, mx[3]);
        lr.nv7[k] = nvv; lr.dv7[k] = dv;
        if (WX) lr.lx7[k] = x2;
    }
}

// role X behind the barrier: xi_1 (and xi) on (x_i, u_i) with the children's L'dv (L'xi_2) from the roles C_c, added child by child
// in the order b_local_g accumulates them.  crow = rows per child block of `ex`, oX = offset of the L'xi_2 rows inside it
template <class P, bool WX>
SPK_DEV void lv_finish_x_res(const P &p, const ProxScal &ps, LVec ex, int crow, int oX, real *mx, const LvRole<P> &lr) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu;
    const real nig = ps.inv_gamma_neg;
    SPK_UNROLL for (int k = 0; k < nxu; ++k) {
        const real dz = k < nx ? lr.dzx[k] : lr.dzu[k - nx];
        real lt = lr.dv7[k], lx = WX ? lr.lx7[k] : (real)0.;
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            lt += ex[c * crow + k];
            if (WX) lx += ex[c * crow + oX + k];
        }
        const real x1 = lt + nig * dz;
        spk_resmax<IVec>(mx[0], x1, mx[3]);
        if (WX) spk_resmax<IVec>(mx[2], lx + x1, mx[3]);
    }
}

// role X behind the barrier: next phase A on (x_i, u_i) (implicit_l.jl:351-373, :399-420) with the children's L'vbar from the roles
// C_c, then c_i / e_i (a_finish_ce).  Bottom stage (BOT): the backward step with the leaves' W_c q_c, V_c q_c from the roles L_c
// (implicit_l.jl:584-637) -> q_i to cq, d_i to dbuf.  Other stages: c_i is parked in cq's row of the node and e_i in dbuf's (the
// d_i of this iteration has been consumed in front of the barrier); LA turns them into q_i, d_i when the children's q exist.
// DST (two-stage kernel LBA2): 1 = c_i, e_i to the LDS rows `qd` (the parent, whose backward step follows in the same workgroup),
// 2 = q_i to the LDS rows `qd` instead of cq (a child: only its parent's backward step reads it)
template <class P, bool BOT, int DST = 0, class DB, class CQ>
SPK_DEV void lv_finish_x_next(const P &p, const Dims &dm, MatPtr M, const ProxScal &ps, LVec ex, int crow, int rL, int node, int stage,
                              DB db, CQ cq, const LvRole<P> &lr, LVec qd = LVec{nullptr}) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu;
    const real mg = -ps.gamma;
    real xbar[P::MAXNX], ubar[P::MAXNU];
    SPK_UNROLL for (int k = 0; k < nxu; ++k) {
        real acc = lr.nv7[k];                                          // a_local_g: acc = v7, += sqrtQ v3_c / sqrtR v4_c child by child
        SPK_UNROLL for (int c = 0; c < d; ++c) acc += ex[c * crow + nxu + k];
        if (k < nx) xbar[k] = mg * acc + lr.xi[k]; else ubar[k - nx] = mg * acc + lr.ui[k - nx];
    }
    TState<P> s;
    s.stage = stage; s.node = node;
    a_finish_ce(p, dm, M, s, xbar, ubar);
    if (BOT) {
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int base = rL + c * (nxu + 1);
            SPK_UNROLL for (int r = 0; r < nx; ++r) s.xb[r] += ex[base + r];
            SPK_UNROLL for (int r = 0; r < nu; ++r) s.ub[r] -= ex[base + nx + r];
        }
    }
    if (DST == 1) {
        SPK_UNROLL for (int k = 0; k < nx; ++k) qd[k] = s.xb[k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) qd[nx + k] = s.ub[k];
        return;
    }
    SPK_UNROLL for (int k = 0; k < nu; ++k) db[node * nu + k] = s.ub[k];
    if (DST == 2) { SPK_UNROLL for (int k = 0; k < nx; ++k) qd[k] = s.xb[k]; }
    else { SPK_UNROLL for (int k = 0; k < nx; ++k) cq[node * nx + k] = s.xb[k]; }
}

// ---- role R: risk block v1, v2 of node i (implicit_l.jl:208-222, :833-851) --------------------------------------------------
template <class P, bool WX, bool NEXTA, class ZI, class VI, class ZB, class VO>
SPK_DEV void lv_role_r(const P &p, const Dims &dm, MatPtr M, ZI z, VI v, ZB zb, VO vo, int node, const ProxScal &ps, real *mx,
                       LvRole<P> &lr) {
    const int ny = p.ny(), i = node;
    const real sg = ps.sigma, isg = ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg, mgn = -ps.gamma;
    const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? i * ny : 0);
    real zby[P::MAXNY], zy[P::MAXNY], v1[P::MAXNY];
    SPK_UNROLL for (int k = 0; k < ny; ++k) { zby[k] = zb[dm.zy + i * ny + k]; zy[k] = z[dm.zy + i * ny + k]; v1[k] = v[i * ny + k]; }
    const real zbs = zb[dm.zs + i], zs_ = z[dm.zs + i], v2 = v[dm.o2 + i];
    real wy[P::MAXNY], dzy[P::MAXNY], t1[P::MAXNY], pr[P::MAXNY];
    real bw = 0., bdz = 0.;
    SPK_UNROLL for (int k = 0; k < ny; ++k) {
        wy[k] = 2. * zby[k] + -1. * zy[k];
        dzy[k] = zby[k] - zy[k];
        bw += bb[k] * wy[k];
        bdz += bb[k] * dzy[k];
    }
    const real ws = 2. * zbs + -1. * zs_, dzs = zbs - zs_;
    SPK_UNROLL for (int k = 0; k < ny; ++k) t1[k] = sg * wy[k] + v1[k];
    real t2 = sg * (ws - bw) + v2;
    SPK_UNROLL for (int k = 0; k < ny; ++k) { t1[k] = t1[k] * isg; pr[k] = t1[k]; }
    if (!dm.risk_has_soc) {
        SPK_UNROLL for (int k = 0; k < ny; ++k) {
            const int kind = (dm.elem_kind_mask >> (2 * k)) & 3;
            const real v0 = pr[k];
            pr[k] = kind == 0 ? 0. : (kind == 1 ? (v0 < 0. ? 0. : v0) : (kind == 2 ? (v0 > 0. ? 0. : v0) : v0));
        }
    } else {
        int off = 0;
        for (int cn = 0; cn < dm.n_cones; ++cn) {
            const int kind = dm.cone_kind[cn], dim = dm.cone_dim[cn];
            real t = 0., ss = 0.;
            SPK_UNROLL for (int k = 0; k < ny; ++k) {
                if (k == off) t = pr[k];
                if (k > off && k < off + dim) ss += pr[k] * pr[k];
            }
            const real x_norm = sqrt(ss);
            const bool keep = (x_norm <= t), zero = !keep && (x_norm <= -t);
            const real tq = (t + x_norm) / 2., sc = tq / x_norm;
            SPK_UNROLL for (int k = 0; k < ny; ++k) {
                if (k < off || k >= off + dim) continue;
                const real v0 = pr[k];
                real r;
                if (kind == 0) r = 0.;
                else if (kind == 1) r = v0 < 0. ? 0. : v0;
                else if (kind == 2) r = v0 > 0. ? 0. : v0;
                else if (kind == 4) r = keep ? v0 : (zero ? 0. : (k == off ? tq : sc * v0));
                else r = v0;
                pr[k] = r;
            }
            off += dim;
        }
    }
    SPK_UNROLL for (int k = 0; k < ny; ++k) pr[k] = sg * (t1[k] - pr[k]);
    t2 = t2 * isg;
    const real p2 = t2 >= 0. ? t2 : 0.;
    const real nv2 = sg * (t2 - p2);
    const real dv2 = nv2 - v2;
    const real x2 = (dzs - bdz) + nis * dv2;
    spk_resmax<IVec>(mx[1], x2, mx[3]);
    vo[dm.o2 + i] = nv2;
    SPK_UNROLL for (int k = 0; k < ny; ++k) {
        const real dv1 = pr[k] - v1[k];
        vo[i * ny + k] = pr[k];
        const real x2k = dzy[k] + nis * dv1;
        spk_resmax<IVec>(mx[1], x2k, mx[3]);
        const real x1k = (dv1 - bb[k] * dv2) + nig * dzy[k];
        spk_resmax<IVec>(mx[0], x1k, mx[3]);
        if (WX) spk_resmax<IVec>(mx[2], (x2k - bb[k] * x2) + x1k, mx[3]);
        if (NEXTA) lr.vecy[k] = mgn * (pr[k] - bb[k] * nv2) + zby[k];   // the y part of (-gamma) L'vbar + zbar (implicit_l.jl:423-429)
    }
    const real x1s = dv2 + nig * dzs;
    spk_resmax<IVec>(mx[0], x1s, mx[3]);
    if (WX) spk_resmax<IVec>(mx[2], x2 + x1s, mx[3]);
    if (NEXTA) lr.sown = mgn * nv2 + zbs;   // this node's s entry of its PARENT's [y; s_ch; tau_ch] (lv_role_ar; used by LBA2 only)
}

// role R behind the barrier, bottom stage: S2 on [y_i; s_ch; tau_ch] (implicit_l.jl:708-726) -> the next zbar
template <class P, class ZN>
SPK_DEV void lv_finish_r(const P &p, const Dims &dm, MatPtr M, LVec ex, int crow, int rL, int node, ZN zn, const LvRole<P> &lr) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu;
    real vec[P::MAXM];
    SPK_UNROLL for (int k = 0; k < ny; ++k) vec[k] = lr.vecy[k];
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        vec[ny + c] = ex[rL + c * (nxu + 1) + nxu];
        vec[ny + d + c] = ex[c * crow + crow - 1];   // tau_j: the last row of child c's block
    }
    TState<P> s;
    s.node = node; s.c0 = node * d + 1;
    a_finish_s2(p, dm, M, zn, s, vec);
}

// ---- role C_c: the cone block of child c, (v6_j; v3_j, v4_j, v5_j) in SOC(nx+nu+2) (implicit_l.jl:226-272, :854-900) and its L'
// contributions on (x_i, u_i).  Every load of the block is issued up front (one round trip to memory).  Rows of `ex` from
// `base` on: [L'dv (nxu)] [L'vbar (nxu)] [L'xi_2 (nxu), WX] [tau_j of the next phase A, TAU = bottom stage]
template <class P, bool WX, bool TAU, class ZI, class VI, class ZB, class VO, class DB>
SPK_DEV void lv_role_c(const P &p, const Dims &dm, MatPtr M, MatPtr SW, ZI z, VI v, ZB zb, VO vo, DB db, int node, int c, const ProxScal &ps,
                       real *mx, LVec ex, int base, LVec xs = LVec{nullptr}, LVec gu = LVec{nullptr}) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu, i = node;
    const real sg = ps.sigma, isg = ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg, mgn = -ps.gamma;
    const int j = i * d + 1 + c, jn = j - 1;
    real xi[P::MAXNX], di[P::MAXNU], zx_[P::MAXNX], zu_[P::MAXNU];
    if (gu) { SPK_UNROLL for (int k = 0; k < nx; ++k) xi[k] = xs[k]; }
    else lv_load_xd(p, dm, zb, db, i, xi, di, xs);
    SPK_UNROLL for (int k = 0; k < nx; ++k) zx_[k] = z[dm.zx + i * nx + k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) zu_[k] = z[dm.zu + i * nu + k];
    real vin[P::MAXNX + P::MAXNU + 2];   // [v6, v3.., v4.., v5]: the ordering of the reference's workspace
    vin[0] = v[dm.o6 + jn];
    SPK_UNROLL for (int r = 0; r < nx; ++r) vin[1 + r] = v[dm.o3 + jn * nx + r];
    SPK_UNROLL for (int r = 0; r < nu; ++r) vin[1 + nx + r] = v[dm.o4 + jn * nu + r];
    vin[1 + nxu] = v[dm.o5 + jn];
    const real zbt = zb[dm.zt + jn], zt_ = z[dm.zt + jn];
    real wxv[P::MAXNX], dzx[P::MAXNX], wuv[P::MAXNU], dzu[P::MAXNU];
    {
        real ui[P::MAXNU];
        if (gu) { SPK_UNROLL for (int k = 0; k < nu; ++k) ui[k] = gu[k]; }
        else lv_u_of(p, dm, SW, xi, di, ui);
        SPK_UNROLL for (int k = 0; k < nx; ++k) { wxv[k] = 2. * xi[k] + -1. * zx_[k]; dzx[k] = xi[k] - zx_[k]; }
        SPK_UNROLL for (int k = 0; k < nu; ++k) { wuv[k] = 2. * ui[k] + -1. * zu_[k]; dzu[k] = ui[k] - zu_[k]; }
    }
    const MatPtr SQ = M + dm.mSqQ + (P::SHARED ? 0 : jn) * nx * nx;
    const MatPtr SR = M + dm.mSqR + (P::SHARED ? 0 : jn) * nu * nu;
    const real wt = 2. * zbt + -1. * zt_, dzt = zbt - zt_;
    real tt[P::MAXNX + P::MAXNU + 2], pj[P::MAXNX + P::MAXNU + 2], lz[P::MAXNX + P::MAXNU + 2];
    sym_mv<P, false>(nx, SQ, wxv, tt + 1);
    sym_mv<P, false>(nx, SQ, dzx, lz + 1);
    sym_mv<P, false>(nu, SR, wuv, tt + 1 + nx);
    sym_mv<P, false>(nu, SR, dzu, lz + 1 + nx);
    tt[0] = sg * (0.5 * wt) + vin[0];
    lz[0] = 0.5 * dzt;
    SPK_UNROLL for (int r = 0; r < nxu; ++r) tt[1 + r] = sg * tt[1 + r] + vin[1 + r];
    tt[1 + nxu] = sg * (0.5 * wt) + vin[1 + nxu];
    lz[1 + nxu] = 0.5 * dzt;
    const int len = nxu + 2;
    SPK_UNROLL for (int k = 0; k < len; ++k) tt[k] = tt[k] * isg;
    tt[1 + nxu] -= 0.5;   // v5 -= 1/2, v6 += 1/2  (implicit_l.jl:928-933)
    tt[0] += 0.5;
    SPK_UNROLL for (int k = 0; k < len; ++k) pj[k] = tt[k];
    project_soc<P::MAXNX + P::MAXNU + 2>(pj, len);
    SPK_UNROLL for (int k = 0; k < len; ++k) pj[k] = sg * (tt[k] - pj[k]);
    vo[dm.o6 + jn] = pj[0];
    SPK_UNROLL for (int r = 0; r < nx; ++r) vo[dm.o3 + jn * nx + r] = pj[1 + r];
    SPK_UNROLL for (int r = 0; r < nu; ++r) vo[dm.o4 + jn * nu + r] = pj[1 + nx + r];
    vo[dm.o5 + jn] = pj[1 + nxu];
    int row = base + nxu;
    {   // L'vbar on x_i, u_i: the (x_i, u_i) half of the next phase A (implicit_l.jl:365-373, :412-420)
        real ltv[P::MAXNX + P::MAXNU];
        sym_mv<P, false>(nx, SQ, pj + 1, ltv);
        sym_mv<P, false>(nu, SR, pj + 1 + nx, ltv + nx);
        SPK_UNROLL for (int k = 0; k < nxu; ++k) ex[row + k] = ltv[k];
        row += nxu;
    }
    SPK_UNROLL for (int k = 0; k < len; ++k) tt[k] = pj[k] - vin[k];                 // dv
    SPK_UNROLL for (int k = 0; k < len; ++k) { lz[k] = lz[k] + nis * tt[k]; spk_resmax<IVec>(mx[1], lz[k], mx[3]); }   // xi_2
    {
        real ltd[P::MAXNX + P::MAXNU];
        sym_mv<P, false>(nx, SQ, tt + 1, ltd);
        sym_mv<P, false>(nu, SR, tt + 1 + nx, ltd + nx);
        SPK_UNROLL for (int k = 0; k < nxu; ++k) ex[base + k] = ltd[k];
    }
    if (WX) {
        real ltx[P::MAXNX + P::MAXNU];
        sym_mv<P, false>(nx, SQ, lz + 1, ltx);
        sym_mv<P, false>(nu, SR, lz + 1 + nx, ltx + nx);
        SPK_UNROLL for (int k = 0; k < nxu; ++k) ex[row + k] = ltx[k];
        row += nxu;
    }
    if (TAU) ex[row] = mgn * (0.5 * pj[1 + nxu] + 0.5 * pj[0]) + zbt;   // tau_j of the next phase A (implicit_l.jl:432-437)
    const real x1t = (0.5 * tt[1 + nxu] + 0.5 * tt[0]) + nig * dzt;   // xi_1 on tau_j (implicit_l.jl:432-437)
    spk_resmax<IVec>(mx[0], x1t, mx[3]);
    if (WX) spk_resmax<IVec>(mx[2], (0.5 * lz[1 + nxu] + 0.5 * lz[0]) + x1t, mx[3]);
}

// ---- role L_c, bottom stage: leaf child c -- its x from the forward step, (v13; v11, v12) in SOC(nx+2), v14 in the x box
// (implicit_l.jl:286-317, :752-810), its entries of the next phase A (implicit_l.jl:376-396, :445-448) and with them its part of
// the parent's backward step: rows [W_c q_c | V_c q_c | s_c] of `ex` from rL on, q_c = -xbar_c (implicit_l.jl:578-582)
template <class P, bool WX, class ZI, class VI, class ZB, class VO, class DB>
SPK_DEV void lv_role_l(const P &p, const Dims &dm, MatPtr M, MatPtr SW, MatPtr Bm, ZI z, VI v, ZB zb, VO vo, DB db, int node, int c,
                       const ProxScal &ps, real *mx, LVec ex, int rL, LVec xs = LVec{nullptr}, LVec gxl = LVec{nullptr}) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), i = node;
    const real sg = ps.sigma, isg = ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg, mgn = -ps.gamma;
    const int j = i * d + 1 + c, li = j - dm.nn;
    real xl[P::MAXNX];
    if (gxl) {   // the leaf's x is given (and stored) by the forward sweep of the role-lane subtree kernel
        SPK_UNROLL for (int k = 0; k < nx; ++k) xl[k] = gxl[k];
    } else {
        real xi[P::MAXNX], di[P::MAXNU];
        lv_load_xd(p, dm, zb, db, i, xi, di, xs);
        lv_child_x(p, dm, SW, Bm, c, xi, di, xl);
    }
    real zl[P::MAXNX], v11[P::MAXNX], v14[P::MAXNX];
    SPK_UNROLL for (int k = 0; k < nx; ++k) { zl[k] = z[dm.zx + j * nx + k]; v11[k] = v[dm.o11 + li * nx + k]; v14[k] = v[dm.o14 + li * nx + k]; }
    const real zbs = zb[dm.zs + j], zs_ = z[dm.zs + j], v12 = v[dm.o12 + li], v13 = v[dm.o13 + li];
    if (!gxl) { SPK_UNROLL for (int k = 0; k < nx; ++k) zb[dm.zx + j * nx + k] = xl[k]; }
    const MatPtr SN = M + dm.mSqQN + (P::SHARED ? 0 : li) * nx * nx;
    real wl[P::MAXNX], dl[P::MAXNX];
    SPK_UNROLL for (int k = 0; k < nx; ++k) { wl[k] = 2. * xl[k] + -1. * zl[k]; dl[k] = xl[k] - zl[k]; }
    const real ws = 2. * zbs + -1. * zs_, dzs = zbs - zs_;
    real t3[P::MAXNX + 2], p3[P::MAXNX + 2], l3[P::MAXNX + 2];   // [v13, v11.., v12]
    sym_mv<P, false>(nx, SN, wl, t3 + 1);
    sym_mv<P, false>(nx, SN, dl, l3 + 1);
    t3[0] = sg * (0.5 * ws) + v13;
    l3[0] = 0.5 * dzs;
    SPK_UNROLL for (int r = 0; r < nx; ++r) t3[1 + r] = sg * t3[1 + r] + v11[r];
    t3[1 + nx] = sg * (0.5 * ws) + v12;
    l3[1 + nx] = 0.5 * dzs;
    const int l2 = nx + 2;
    SPK_UNROLL for (int k = 0; k < l2; ++k) t3[k] = t3[k] * isg;
    t3[1 + nx] -= 0.5;   // v12 -= 1/2, v13 += 1/2 (implicit_l.jl:934-939)
    t3[0] += 0.5;
    SPK_UNROLL for (int k = 0; k < l2; ++k) p3[k] = t3[k];
    project_soc<P::MAXNX + 2>(p3, l2);
    SPK_UNROLL for (int k = 0; k < l2; ++k) p3[k] = sg * (t3[k] - p3[k]);
    vo[dm.o13 + li] = p3[0];
    SPK_UNROLL for (int r = 0; r < nx; ++r) vo[dm.o11 + li * nx + r] = p3[1 + r];
    vo[dm.o12 + li] = p3[1 + nx];
    const int nxu = nx + nu;
    ex[rL + c * (nxu + 1) + nxu] = mgn * (0.5 * p3[1 + nx] + 0.5 * p3[0]) + zbs;   // s_j of the next phase A (implicit_l.jl:445-448)
    real q11[P::MAXNX];
    sym_mv<P, false>(nx, SN, p3 + 1, q11);
    t3[0] = p3[0] - v13;
    SPK_UNROLL for (int r = 0; r < nx; ++r) t3[1 + r] = p3[1 + r] - v11[r];
    t3[1 + nx] = p3[1 + nx] - v12;
    SPK_UNROLL for (int k = 0; k < l2; ++k) { l3[k] = l3[k] + nis * t3[k]; spk_resmax<IVec>(mx[1], l3[k], mx[3]); }
    real a11[P::MAXNX], b11[P::MAXNX];
    sym_mv<P, false>(nx, SN, t3 + 1, a11);
    if (WX) sym_mv<P, false>(nx, SN, l3 + 1, b11);
    real qc[P::MAXNX];
    SPK_UNROLL for (int r = 0; r < nx; ++r) {
        real t = sg * wl[r] + v14[r];
        t = t * isg;
        const real nvv = sg * (t - spk_clamp(t, dm.x_min, dm.x_max));
        vo[dm.o14 + li * nx + r] = nvv;
        qc[r] = -(mgn * (nvv + q11[r]) + xl[r]);   // q_leaf = -xbar_leaf of the next iteration
        const real dv = nvv - v14[r];
        const real x2 = dl[r] + nis * dv;
        spk_resmax<IVec>(mx[1], x2, mx[3]);
        const real x1 = (dv + a11[r]) + nig * dl[r];
        spk_resmax<IVec>(mx[0], x1, mx[3]);
        if (WX) spk_resmax<IVec>(mx[2], (x2 + b11[r]) + x1, mx[3]);
    }
    const real x1s = (0.5 * t3[1 + nx] + 0.5 * t3[0]) + nig * dzs;   // s_leaf (implicit_l.jl:445-448)
    spk_resmax<IVec>(mx[0], x1s, mx[3]);
    if (WX) spk_resmax<IVec>(mx[2], (0.5 * l3[1 + nx] + 0.5 * l3[0]) + x1s, mx[3]);
    // the leaf's part of the parent's backward step (bwd_node)
    const MatPtr W = SW + dm.sw_oW + c * nx * nx, V = SW + dm.sw_oV + c * nu * nx;
    const int base = rL + c * (nxu + 1);
    SPK_UNROLL for (int r = 0; r < nx; ++r) {
        real a = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += W[r + k * nx] * qc[k];
        ex[base + r] = a;
    }
    SPK_UNROLL for (int r = 0; r < nu; ++r) {
        real a = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += V[r + k * nu] * qc[k];
        ex[base + nx + r] = a;
    }
}

// ---- LA roles (non-bottom stages, bottom-up): the two independent halves of a_local_g -------------------------------------
// AX: the backward step of node i (implicit_l.jl:584-637) from c_i, e_i -- parked in cq / dbuf by LB's role X -- and the children's
// q from cq: q_i = c_i + sum_c W_c q_c back to cq, d_i = e_i - sum_c V_c q_c back to dbuf
template <class P, class DB, class CQ>
SPK_DEV void lv_role_ax(const P &p, const Dims &dm, MatPtr SW, DB db, CQ cq, int node, int pos, int stage, int first_child_stage_node) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), i = node;
    TState<P> s;
    s.node = i; s.c0 = i * d + 1; s.stage = stage; s.lvl = 0; s.pos = pos; s.kids_leaf = false; s.kids_gleaf = true; s.active = true;
    SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = cq[(long)i * nx + k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) s.ub[k] = db[i * nu + k];
    bwd_node(p, dm, SW, 0, LVec{nullptr}, cq + (long)first_child_stage_node * nx, cq + (long)i * nx, s);
    SPK_UNROLL for (int k = 0; k < nu; ++k) db[i * nu + k] = s.dd[k];
}
// AR: [y_i; s_ch; tau_ch] of (-gamma) L'v + z (implicit_l.jl:423-448) and S2 on it -> zo
template <class P, class ZI, class VI, class ZO>
SPK_DEV void lv_role_ar(const P &p, const Dims &dm, MatPtr M, ZI z, VI v, ZO zo, int node, real gamma) {
    const int d = p.d(), ny = p.ny(), i = node;
    const real mg = -gamma;
    TState<P> s;
    s.node = i; s.c0 = i * d + 1;
    real vec[P::MAXM];
    const real v2i = v[dm.o2 + i];
    const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? i * ny : 0);
    SPK_UNROLL for (int k = 0; k < ny; ++k) vec[k] = mg * (v[i * ny + k] - bb[k] * v2i) + z[dm.zy + i * ny + k];
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const int j = s.c0 + c;
        const real ls = v[dm.o2 + j];   // children are non-leaf nodes here
        vec[ny + c] = mg * ls + z[dm.zs + j];
        vec[ny + d + c] = mg * (0.5 * v[dm.o5 + j - 1] + 0.5 * v[dm.o6 + j - 1]) + z[dm.zt + j - 1];
    }
    a_finish_s2(p, dm, M, zo, s, vec);
}

// ---------------------------------------------------------------------------------------------------------------------------
// X: the execution policy of spock_igroup.hpp.  g = chunk of the stage, tile = 64 instances.
template <class P, int MODE, bool WX, class X>
SPK_DEV void ilevel_body(X &x, const KArgs &a, const real *zin_, const real *vin_, real *zout_, real *vout_, real *znext_, int tile, int g) {
    const Dims &dm = a.dm;
    const P p(dm);
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nxu = nx + nu;
    const long bp = a.bp, t64 = (long)tile * 64;
    const int nW = x.nthreads() / 64;
    const int stage = a.grp_level;
    const int cnt = a.ls_cnt, first = a.ls_first, first_child = a.ls_first_child;   // (scalars: no indexing into Dims' arrays)
    const int lo = g * a.ls_chunk_k, hi = (lo + a.ls_chunk_k < cnt) ? lo + a.ls_chunk_k : cnt;
    const MatPtr M = (MatPtr)a.mats, SW = M + dm.mSweep + stage * dm.sw_stride, Bm = M + dm.mB;
    const int n_groups = dm.n_bslots + dm.n_crown_groups;
    double *partials = a.partials + (long)tile * n_groups * 3 * 64;

    x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
        const int inst = tile * 64 + (tid & 63);
        st.live = inst < a.batch && !a.ctl[inst].done;
        st.mx[0] = st.mx[1] = st.mx[2] = st.mx[3] = 0.;
    });
    const real *zin = zin_ + t64, *vin = vin_ + t64;
    real *zout = zout_ + t64, *vout = vout_ ? vout_ + t64 : nullptr, *znext = znext_ ? znext_ + t64 : nullptr;
    real *dbuf = a.dbuf + t64, *cq = a.cq + t64;
    auto vw = [&](const real *b, long n_elems, int tid) SPK_ILAMBDA { return b ? ivec_make(b, bp, tid & 63, n_elems) : ivec_null(); };
    const long n_db = (long)dm.nn * nu, n_cq = (long)dm.nn * nx;
    auto make_ps = [&](const Ctl &ctl) SPK_ILAMBDA {
        ProxScal ps;
        ps.sigma = ctl.sigma; ps.inv_sigma = 1. / ctl.sigma; ps.inv_sigma_neg = -1. / ctl.sigma;
        ps.inv_gamma_neg = -1. / ctl.gamma; ps.gamma = ctl.gamma;
        ps.want_xi = WX;
        return ps;
    };

    if (MODE == MODE_LA0) {   // prologue: phase A + backward step of iteration 0, (z, v) -> zbar; one wave per node
        const bool bottom = (stage == dm.N - 2);
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            if (!st.live) return;
            const int wave = spk_uniform(tid >> 6);
            const real gamma = a.ctl[tile * 64 + (tid & 63)].gamma;
            const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid);
            const IVec db = vw(dbuf, n_db, tid), cqv = vw(cq, n_cq, tid);
            for (int pos = lo + wave; pos < hi; pos += nW) {
                TState<P> &s = st.s;
                s.node = first + pos; s.c0 = s.node * d + 1; s.stage = stage; s.lvl = 0; s.pos = pos; s.active = true;
                s.kids_leaf = bottom; s.kids_gleaf = true;
                a_local_g(p, dm, M, z, v, zo, s, gamma, 0u, (const real *)nullptr);
                bwd_node(p, dm, SW, 0, LVec{nullptr}, cqv + (long)first_child * nx, cqv + (long)s.node * nx, s);
                SPK_UNROLL for (int k = 0; k < nu; ++k) db[s.node * nu + k] = s.dd[k];
            }
        });
        return;
    }
    if (MODE == MODE_LA) {   // next phase A + backward step on (zbar, vbar) -> znext; two independent role waves per node
        x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
            if (!st.live) return;
            const int wave = spk_uniform(tid >> 6), role = wave & 1;
            const real gamma = a.ctl[tile * 64 + (tid & 63)].gamma;
            const IVec zb = vw(zout, dm.nz, tid), vb = vw(vout, dm.nv, tid), zn = vw(znext, dm.nz, tid);
            const IVec db = vw(dbuf, n_db, tid), cqv = vw(cq, n_cq, tid);
            for (int pos = lo + (wave >> 1); pos < hi; pos += (nW >> 1)) {
                const int node = first + pos;
                if (role == 0) lv_role_ax(p, dm, SW, db, cqv, node, pos, stage, first_child);
                else lv_role_ar(p, dm, M, zb, vb, zn, node, gamma);
            }
        });
        return;
    }

    x.each([&](int tid, IState<P> &st) SPK_ILAMBDA { if (st.live) st.ps = make_ps(a.ctl[tile * 64 + (tid & 63)]); });
    if (MODE == MODE_LBA2) {
        // ---- the two bottom stages in one launch: `stage` = N-3.  Per parent: its LB step, the LBA step of each child, then its own
        // backward step and S2 (what LA does) -- everything that flowed between the three launches through HBM stays in the workgroup:
        //   x of the children       role X -> LDS rows `xrows` (and zbar, where it belongs)   instead of read back from zbar
        //   c_i, e_i of the parent  role X -> LDS rows `cerows`                                instead of parked in cq / dbuf
        //   q of the children       role X -> LDS rows `qrows`                                 instead of written to / read from cq
        //   [y; s_ch; tau_ch]       role R -> LDS rows `vprows`, piece by piece               instead of LA's re-reads of zbar / vbar
        // The parent's backward step and S2 run on the first two L_c waves -- idle during a parent's step -- behind the barrier of the
        // NEXT parent's step (cerows / vprows are double-buffered over the parents), and once more after the last parent.
        // The arithmetic per entry is that of LB / LBA / LA (the same functions), so the iterates are those of the three launches.
        const MatPtr SWc = SW + dm.sw_stride;
        const int ny = p.ny(), mrow = ny + 2 * d;
        const int rows = lba_exchange_rows(nx, nu, d, WX);
        const int crow = (2 + (WX ? 1 : 0)) * nxu + 1, oX = 2 * nxu, rL = d * crow;
        real *const xrows = x.lds() + 2L * rows * 64, *const qrows = xrows + (long)d * nx * 64;
        real *const cerows = qrows + (long)d * nx * 64, *const vprows = cerows + 2L * nxu * 64;
        // backward step (implicit_l.jl:584-637) and S2 (implicit_l.jl:708-726) of parent `node` from the LDS rows of parity `par`
        auto parent_tail = [&](int tid, int which, int node, int par) SPK_ILAMBDA {
            if (which == 0) {
                const IVec db = vw(dbuf, n_db, tid), cqv = vw(cq, n_cq, tid);
                const LVec ce = LVec{cerows + (long)par * nxu * 64 + (tid & 63)};
                TState<P> s;
                s.node = node; s.c0 = node * d + 1; s.stage = stage; s.lvl = 0; s.pos = 0; s.kids_leaf = false; s.kids_gleaf = true; s.active = true;
                SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = ce[k];
                SPK_UNROLL for (int k = 0; k < nu; ++k) s.ub[k] = ce[nx + k];
                bwd_node(p, dm, SW, 0, LVec{nullptr}, LVec{qrows + (tid & 63)}, cqv + (long)node * nx, s);
                SPK_UNROLL for (int k = 0; k < nu; ++k) db[node * nu + k] = s.dd[k];
            } else {
                const LVec vp = LVec{vprows + (long)par * mrow * 64 + (tid & 63)};
                real vec[P::MAXM];
                SPK_UNROLL for (int k = 0; k < mrow; ++k) vec[k] = vp[k];
                TState<P> s;
                s.node = node; s.c0 = node * d + 1;
                a_finish_s2(p, dm, M, vw(znext, dm.nz, tid), s, vec);
            }
        };
        int step = 0;
        for (int pos = lo; pos < hi; ++pos) {
            const int node = first + pos, par = (pos - lo) & 1;
            {
                const int buf = step & 1;
                ++step;
                x.phased([&](int tid, IState<P> &st, bool pre, bool post) SPK_ILAMBDA {
                    if (!st.live) return;
                    const int role = spk_uniform(tid >> 6);
                    const ProxScal &ps = st.ps;
                    const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid), vo = vw(vout, dm.nv, tid);
                    const IVec db = vw(dbuf, n_db, tid);
                    const LVec ex = LVec{x.lds() + (long)buf * rows * 64 + (tid & 63)};
                    if (role == 0) {
                        if (pre) lv_role_x<P, WX>(p, dm, SW, Bm, z, v, zo, vo, db, node, true, ps, st.mx, st.lr, LVec{nullptr}, LVec{xrows + (tid & 63)});
                        x.sync();
                        if (post) {
                            lv_finish_x_res<P, WX>(p, ps, ex, crow, oX, st.mx, st.lr);
                            lv_finish_x_next<P, false, 1>(p, dm, M, ps, ex, crow, rL, node, stage, db, db, st.lr,
                                                          LVec{cerows + (long)par * nxu * 64 + (tid & 63)});
                        }
                    } else if (role == 1) {
                        if (pre) lv_role_r<P, WX, true>(p, dm, M, z, v, zo, vo, node, ps, st.mx, st.lr);
                        x.sync();
                        if (post) {
                            const LVec vp = LVec{vprows + (long)par * mrow * 64 + (tid & 63)};
                            SPK_UNROLL for (int k = 0; k < ny; ++k) vp[k] = st.lr.vecy[k];
                            SPK_UNROLL for (int c = 0; c < d; ++c) vp[ny + d + c] = ex[c * crow + crow - 1];
                        }
                    } else if (role < 2 + d) {
                        if (pre) lv_role_c<P, WX, true>(p, dm, M, SW, z, v, zo, vo, db, node, role - 2, ps, st.mx, ex, (role - 2) * crow);
                        x.sync();
                    } else {
                        x.sync();
                        if (post && pos > lo && role < 4 + d) parent_tail(tid, role - 2 - d, node - 1, par ^ 1);
                    }
                });
            }
            for (int c = 0; c < d; ++c) {
                const int child = node * d + 1 + c, buf = step & 1;
                ++step;
                x.phased([&](int tid, IState<P> &st, bool pre, bool post) SPK_ILAMBDA {
                    if (!st.live) return;
                    const int role = spk_uniform(tid >> 6);
                    const ProxScal &ps = st.ps;
                    const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid), vo = vw(vout, dm.nv, tid);
                    const IVec db = vw(dbuf, n_db, tid);
                    const LVec ex = LVec{x.lds() + (long)buf * rows * 64 + (tid & 63)};
                    const LVec xs = LVec{xrows + (long)c * nx * 64 + (tid & 63)};
                    if (role == 0) {
                        if (pre) lv_role_x<P, WX>(p, dm, SWc, Bm, z, v, zo, vo, db, child, false, ps, st.mx, st.lr, xs);
                        x.sync();
                        if (post) {
                            lv_finish_x_res<P, WX>(p, ps, ex, crow, oX, st.mx, st.lr);
                            lv_finish_x_next<P, true, 2>(p, dm, M, ps, ex, crow, rL, child, stage + 1, db, db, st.lr,
                                                         LVec{qrows + (long)c * nx * 64 + (tid & 63)});
                        }
                    } else if (role == 1) {
                        if (pre) lv_role_r<P, WX, true>(p, dm, M, z, v, zo, vo, child, ps, st.mx, st.lr);
                        x.sync();
                        if (post) {
                            lv_finish_r(p, dm, M, ex, crow, rL, child, vw(znext, dm.nz, tid), st.lr);
                            const LVec vp = LVec{vprows + (long)par * mrow * 64 + (tid & 63)};
                            vp[ny + c] = st.lr.sown;
                        }
                    } else if (role < 2 + d) {
                        if (pre) lv_role_c<P, WX, true>(p, dm, M, SWc, z, v, zo, vo, db, child, role - 2, ps, st.mx, ex, (role - 2) * crow, xs);
                        x.sync();
                    } else {
                        if (pre) lv_role_l<P, WX>(p, dm, M, SWc, Bm, z, v, zo, vo, db, child, role - 2 - d, ps, st.mx, ex, rL, xs);
                        x.sync();
                    }
                });
            }
        }
        if (hi > lo) {   // the last parent's backward step and S2
            x.barrier();
            x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
                if (!st.live) return;
                const int role = spk_uniform(tid >> 6);
                if (role >= 2 + d && role < 4 + d) parent_tail(tid, role - 2 - d, first + hi - 1, (hi - 1 - lo) & 1);
            });
        }
    }

    // ---- LB / LBA: one wave per role, one node at a time, one barrier per node ---------------------------------------------
    constexpr bool BOT = (MODE == MODE_LBA);
    const int rows = BOT ? lba_exchange_rows(nx, nu, d, WX) : lb_exchange_rows(nx, nu, d, WX);
    // rows of one exchange buffer: per child c a block of `crow` rows from role C_c, [L'dv | L'vbar | L'xi_2 (LOG mode) |
    // tau_j (bottom)], then -- bottom -- per child [W_c q_c | V_c q_c | s_c] from role L_c
    const int crow = (2 + (WX ? 1 : 0)) * nxu + (BOT ? 1 : 0), oX = 2 * nxu, rL = d * crow;
    // Every role is ONE branch that contains its part in front of the barrier, the barrier and its finishing part: what a role
    // carries across the barrier (LvRole) is then live in that role's branch only.  (x.phased: the GPU policy runs the body once
    // with pre = post = true and sync() = the workgroup barrier; the host emulation runs it per thread with pre, then with post.)
    for (int pos = lo; pos < hi && MODE != MODE_LBA2; ++pos) {
        const int node = first + pos, buf = (pos - lo) & 1;
        x.phased([&](int tid, IState<P> &st, bool pre, bool post) SPK_ILAMBDA {
            if (!st.live) return;
            const int role = spk_uniform(tid >> 6);
            const ProxScal &ps = st.ps;
            const IVec z = vw(zin, dm.nz, tid), v = vw(vin, dm.nv, tid), zo = vw(zout, dm.nz, tid), vo = vw(vout, dm.nv, tid);
            const IVec db = vw(dbuf, n_db, tid);
            const LVec ex = LVec{x.lds() + (long)buf * rows * 64 + (tid & 63)};
            if (role == 0) {
                if (pre) lv_role_x<P, WX>(p, dm, SW, Bm, z, v, zo, vo, db, node, !BOT, ps, st.mx, st.lr);
                x.sync();
                if (post) {
                    lv_finish_x_res<P, WX>(p, ps, ex, crow, oX, st.mx, st.lr);
                    lv_finish_x_next<P, BOT>(p, dm, M, ps, ex, crow, rL, node, stage, db, vw(cq, n_cq, tid), st.lr);
                }
            } else if (role == 1) {
                if (pre) lv_role_r<P, WX, BOT>(p, dm, M, z, v, zo, vo, node, ps, st.mx, st.lr);
                x.sync();
                if (post && BOT) lv_finish_r(p, dm, M, ex, crow, rL, node, vw(znext, dm.nz, tid), st.lr);
            } else if (role < 2 + d) {
                if (pre) lv_role_c<P, WX, BOT>(p, dm, M, SW, z, v, zo, vo, db, node, role - 2, ps, st.mx, ex, (role - 2) * crow);
                x.sync();
            } else {
                if (pre && BOT) lv_role_l<P, WX>(p, dm, M, SW, Bm, z, v, zo, vo, db, node, role - 2 - d, ps, st.mx, ex, rL);
                x.sync();
            }
        });
    }
    // residual maxima of this workgroup: per lane over the nodes and roles (spock_igroup.hpp: reduce_maxima)
    x.barrier();
    x.each([&](int tid, IState<P> &st) SPK_ILAMBDA {
        for (int k = 0; k < 3; ++k) x.red()[((tid >> 6) * 3 + k) * 64 + (tid & 63)] = st.live ? (st.mx[3] != st.mx[3] ? st.mx[3] : st.mx[k]) : 0.;
    });
    x.barrier();
    const int slot = a.ls_slot + g;
    x.each([&](int tid, IState<P> &) SPK_ILAMBDA {
        for (int idx = tid; idx < 64 * 3; idx += x.nthreads()) {
            const int k = idx >> 6, lane = idx & 63;
            double m = 0.;
            for (int w = 0; w < nW; ++w) m = spk_max_nan(m, x.red()[(w * 3 + k) * 64 + lane]);
            partials[((long)slot * 3 + k) * 64 + lane] = m;
        }
    });
}

}  // namespace spock
