// spock_flat_local.hpp -- FLAT node-local kernels of the lane-team shapes (large nx / nu: BASELINE configs 3, 4, 5 and the
// nx = 50 trees of examples/server_heat/scaling.jl).
//
// The node-local phases of the CP operator -- phase A: zbar0 = z - gamma L'v, S2, c_i / e_i (L_transpose!, projection_S2!,
// implicit_l.jl:320-449, :688-727); phase B: L(2 zbar - z), the Moreau step around the cone / box projections, the residuals
// (L!, prox_h_conj!, implicit_l.jl:177-318, :752-951; should_terminate!, cp.jl:54-123) -- need nothing from the S1 sweep's
// subtree grouping.  Inside the one-workgroup-per-subtree kernels (spock_group.hpp) they ran as rounds of `nteams` nodes per
// workgroup, 129 workgroups for config 5, each round bound by its barrier and a cold round trip to memory: 330 of config 5's
// 616 us per iteration (in-kernel stamps).  Here they run as kernels of their own over chunks of consecutive
// non-leaf nodes -- one round per workgroup, ~1000 workgroups per instance -- with the arithmetic of spock_team_local.hpp
// unchanged (T lanes per node, lane r owns element r of every per-node vector):
//   FLAT_A    phase A of every non-leaf node on (z, v) -> zbar; c_i, e_i and the leaves' q = -xbar are parked in zbar's own
//             x_i / u_i / x_leaf entries (the forward sweep overwrites them with x_i, u_i, x_leaf later)
//   SUB_A / CROWN / SUB_B (spock_group.hpp, KArgs::flat_local)   the sweeps only: backward from those entries, forward into them
//   FLAT_B    phase B of every non-leaf node on the completed zbar; residual maxima (SP: + inner products) per workgroup
//   FLAT_BA   FLAT_B of iteration k followed by FLAT_A of iteration k + 1 on (zbar, vbar) -> the third z buffer (z <- zbar at
//             lambda = 1, cp.jl:34-52).  Everything the next phase A of a node reads was written by this node's own team, or by an
//             earlier launch, with one exception: vbar.v2 of its NON-LEAF children, which their own teams -- other workgroups of
//             this launch -- produce; the parent's team recomputes them (same arithmetic as tb_p2), like the fused subtree kernel's
//             crown workgroup does (risk_v2_only).
// X: the execution policy of spock_group.hpp (GPU: spock_kernels.hip; host emulation: tests/emu).
#pragma once
#include "spock_device.hpp"
#include "spock_team_local.hpp"
#include <type_traits>

namespace spock {

#ifndef SPK_LAMBDA_INLINE
#ifdef SPOCK_HOST_EMU
#define SPK_LAMBDA_INLINE
#else
#define SPK_LAMBDA_INLINE __attribute__((always_inline))
#endif
#endif

// the team's node in a flat kernel: global non-leaf node i (breadth-first numbering)
SPK_DEV TeamNode flat_team_node(const Dims &dm, int d, int i) {
    TeamNode t;
    t.active = i < dm.nn;
    if (!t.active) i = 0;
    int stage = 0;
    while (stage < dm.N - 2 && i >= dm.first[stage + 1]) ++stage;
    t.li = 0; t.lvl = 0; t.pos = 0;          // pos 0: the leaves' vector handed to ta_p1 starts at this node's first child
    t.node = i; t.c0 = i * d + 1; t.stage = stage;
    t.kids_leaf = (stage == dm.N - 2);
    t.kids_gleaf = true;                     // v2x, when given, holds the children of THIS node (slot pos * d + c = c)
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Phase B of ONE node by its lane team in a single pass: every global load of the node is issued up front, the team-wide sums
// (b'y, b'dz and the squared norms of the second-order-cone blocks) come from a butterfly over the team's lanes -- a team is
// T <= 64 lanes of one wave -- and nothing is loaded twice.  (tb_p1 / tb_p2 of spock_team_local.hpp publish per-lane partials in
// LDS, meet at a workgroup barrier and let every lane add up T entries serially: 256 LDS reads per lane and node at T = 64.)
// The arithmetic per entry is tb_p1 / tb_p2's, statement by statement; only the order of those sums differs.
//   PASS 0  GPU: one pass, sums by shuffles.
//   PASS 1 / 2  host emulation, whose "lanes" run one after the other: pass 1 deposits the per-lane terms in `red` and stops in
//           front of the first sum, pass 2 recomputes them (no side effect has happened yet) and takes the sums from `red`.
// red: [(2 + 2 d)][T] doubles of team scratch (passes 1 / 2 only).
template <int PASS>
SPK_DEV real flat_team_sum(real val, int slot, int r, int T, real *red) {
#ifndef SPOCK_HOST_EMU
    if (PASS == 0) {
        for (int off = T >> 1; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
        return val;
    }
#endif
    if (PASS == 1) { red[slot * T + r] = val; return 0.; }
    real s = 0.;
    for (int k = 0; k < T; ++k) s += red[slot * T + k];
    return s;
}

// The two SCALAR entries of a second-order-cone block and the cone's projection (implicit_l.jl:854-900, :752-810, :928-939;
// risk_measures.jl:47-71): child cones (t = v6_j; ..., v5_j) with q = tau_j, leaf cones (t = v13; ..., v12) with q = s_leaf.  Inputs: zbar.q,
// z.q, the two v entries, the squared norm `ssq` of the cone's VECTOR part (team sum).  sc_eff is what the vector entries are scaled
// with: pj = sigma (t - sc_eff t) -- 1 when the point is kept (1.0 * t is exact), 0 when it projects to zero, tnew / |x| otherwise.
// On the GPU ONE lane per cone evaluates this (lane q of the team owns cone q; its four inputs come from one gather each) and the
// team reads sc_eff through a shuffle -- every lane evaluating every cone cost 4 x (sqrt + division + ~60 instructions) per node.
struct ConeOut { real sc_eff, pA, pB, dvA, dvB, x2A, x2B, x1, dzq; };
SPK_DEV ConeOut cone_eval(real zbq, real zq, real vA, real vB, real ssq, bool wz, bool noprox, const ProxScal &ps) {
    const real sg = ps.sigma, isg = noprox ? 1. : ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg;
    ConeOut o;
    const real wq = wz ? zbq : 2. * zbq + -1. * zq;
    o.dzq = zbq - zq;
    real tA = (sg * (0.5 * wq) + vA) * isg, tB = (sg * (0.5 * wq) + vB) * isg;
    real keep = 1., tnew = tA, sc = 1.;      // projected (t; x) = (tnew; sc * x) unless keep
    if (!noprox) {
        tB -= 0.5; tA += 0.5;                 // v5 -= 1/2, v6 += 1/2 (implicit_l.jl:928-939)
        const real ss = tB * tB + ssq;
        const real x_norm = sqrt(ss);
        if (x_norm <= tA) { keep = 1.; }
        else if (x_norm <= -tA) { keep = 0.; tnew = 0.; sc = 0.; }
        else { keep = 0.; tnew = (tA + x_norm) / 2.; sc = tnew / x_norm; }
    }
    o.sc_eff = keep != 0. ? (real)1. : sc;
    o.pA = noprox ? tA : sg * (tA - (keep != 0. ? tA : tnew));
    o.pB = noprox ? tB : sg * (tB - (keep != 0. ? tB : sc * tB));
    o.dvA = o.pA - vA; o.dvB = o.pB - vB;
    o.x2A = 0.5 * o.dzq + nis * o.dvA; o.x2B = 0.5 * o.dzq + nis * o.dvB;
    o.x1 = (0.5 * o.dvB + 0.5 * o.dvA) + nig * o.dzq;   // xi_1 on tau_j / s_leaf
    return o;
}

template <class P, int SPM, int PASS>
SPK_DEV void fb_node(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z, const real *__restrict__ v,
                     const real *zb, real *__restrict__ vo, const TeamNode &t, int r, int T, real *red, const ProxScal &ps,
                     unsigned flags, real *mx, real *sp = nullptr, const real *dirz = nullptr, const real *dirv = nullptr) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu;
    const int i = t.node;
    const bool wz = (flags & F_W_IS_ZBAR) != 0, noprox = (flags & F_NO_PROX) != 0, wx = ps.want_xi;
    const real sg = ps.sigma, isg = noprox ? 1. : ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg;
    const MatPtr SQ = M + dm.mSqQ, SR = M + dm.mSqR, SN = M + dm.mSqQN, bb = M + dm.mb + (dm.nRm > 1 ? t.node * ny : 0);
    const bool hx = r < nx, hu = r < nu, hy = r < ny, leaf = t.kids_leaf;
    // ---- every load of the node, up front (lanes past a vector's length load nothing)
    real zby = 0., zy = 0., v1 = 0., bbr = 0.;
    if (hy) { zby = zb[dm.zy + i * ny + r]; zy = z[dm.zy + i * ny + r]; v1 = v[i * ny + r]; bbr = bb[r]; }
    const real zbs = zb[dm.zs + i], zs_ = z[dm.zs + i], v2i = v[dm.o2 + i];
    real xi = 0., zxi = 0., v7x = 0., sq_r = 0., sn_r = 0.;
    real v3[P::MAXD], xl[P::MAXD], zxl[P::MAXD], v11[P::MAXD], v14[P::MAXD];
    if (hx) {
        xi = zb[dm.zx + i * nx + r]; zxi = z[dm.zx + i * nx + r]; v7x = v[dm.o7 + i * nxu + r]; sq_r = SQ[r + r * nx];
        SPK_UNROLL for (int c = 0; c < d; ++c) v3[c] = v[dm.o3 + (t.c0 + c - 1) * nx + r];
        if (leaf) {
            sn_r = SN[r + r * nx];
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const int j = t.c0 + c, li = j - dm.nn;
                xl[c] = zb[dm.zx + j * nx + r]; zxl[c] = z[dm.zx + j * nx + r];
                v11[c] = v[dm.o11 + li * nx + r]; v14[c] = v[dm.o14 + li * nx + r];
            }
        }
    }
    real ui = 0., zui = 0., v7u = 0., sr_r = 0., v4[P::MAXD];
    if (hu) {
        ui = zb[dm.zu + i * nu + r]; zui = z[dm.zu + i * nu + r]; v7u = v[dm.o7 + i * nxu + nx + r]; sr_r = SR[r + r * nu];
        SPK_UNROLL for (int c = 0; c < d; ++c) v4[c] = v[dm.o4 + (t.c0 + c - 1) * nu + r];
    }
    // the cones' scalar entries: cone q < d is child q's (zbar.tau, z.tau, v6, v5), cone d + c leaf child c's (zbar.s, z.s, v13, v12).
    // GPU: lane q of the team loads cone q's four values (one gather per kind); host emulation: every lane reads what it needs below
    const int ncones = leaf ? 2 * d : d;
    auto cone_idx = [&](int q, int &iq, int &iA, int &iB) {
        const int cc = q < d ? q : q - d, j = t.c0 + cc;
        if (q < d) { iq = dm.zt + j - 1; iA = dm.o6 + j - 1; iB = dm.o5 + j - 1; }
        else { iq = dm.zs + j; iA = dm.o13 + (j - dm.nn); iB = dm.o12 + (j - dm.nn); }
    };
    int my_iq = 0, my_iA = 0, my_iB = 0;
    real my_zbq = 0., my_zq = 0., my_vA = 0., my_vB = 0.;
    const bool cone_owner = r < ncones;
    if (PASS == 0 && cone_owner) {
        cone_idx(r, my_iq, my_iA, my_iB);
        my_zbq = zb[my_iq]; my_zq = z[my_iq]; my_vA = v[my_iA]; my_vB = v[my_iB];
    }
    // ---- w = 2 zbar - z (cp.jl:23-24), dz = zbar - z, the pre-prox value of every owned v entry (tb_p1)
    real wy = 0., dzy = 0.;
    if (hy) { wy = wz ? zby : 2. * zby + -1. * zy; dzy = zby - zy; }
    real wxv = 0., dzx = 0., t3[P::MAXD], l3[P::MAXD], wl[P::MAXD], dl[P::MAXD], t11[P::MAXD], l11[P::MAXD], t14[P::MAXD];
    real sq[2 * P::MAXD];
    SPK_UNROLL for (int c = 0; c < 2 * d; ++c) sq[c] = 0.;
    if (hx) {
        wxv = wz ? xi : 2. * xi + -1. * zxi; dzx = xi - zxi;
        SPK_UNROLL for (int c = 0; c < d; ++c) {   // v3_j = sqrtQ x_i (implicit_l.jl:226-235)
            t3[c] = (sg * (sq_r * wxv) + v3[c]) * isg;
            l3[c] = sq_r * dzx;
            sq[c] += t3[c] * t3[c];
        }
        if (leaf)
            SPK_UNROLL for (int c = 0; c < d; ++c) {   // v11 = sqrtQN x_leaf, v14 = x_leaf (implicit_l.jl:288-317)
                wl[c] = wz ? xl[c] : 2. * xl[c] + -1. * zxl[c]; dl[c] = xl[c] - zxl[c];
                t11[c] = (sg * (sn_r * wl[c]) + v11[c]) * isg;
                l11[c] = sn_r * dl[c];
                t14[c] = (sg * wl[c] + v14[c]) * isg;
                sq[d + c] += t11[c] * t11[c];
            }
    }
    real wuv = 0., dzu = 0., t4[P::MAXD], l4[P::MAXD];
    if (hu) {
        wuv = wz ? ui : 2. * ui + -1. * zui; dzu = ui - zui;
        SPK_UNROLL for (int c = 0; c < d; ++c) {   // v4_j = sqrtR u_i (implicit_l.jl:239-248)
            t4[c] = (sg * (sr_r * wuv) + v4[c]) * isg;
            l4[c] = sr_r * dzu;
            sq[c] += t4[c] * t4[c];
        }
    }
    // ---- the team-wide sums
    const real bw = flat_team_sum<PASS>(bbr * wy, 0, r, T, red), bdz = flat_team_sum<PASS>(bbr * dzy, 1, r, T, red);
    real ssq[2 * P::MAXD];
    SPK_UNROLL for (int c = 0; c < d; ++c) ssq[c] = flat_team_sum<PASS>(sq[c], 2 + c, r, T, red);
    SPK_UNROLL for (int c = 0; c < d; ++c) ssq[d + c] = leaf ? flat_team_sum<PASS>(sq[d + c], 2 + d + c, r, T, red) : (real)0.;
    if (PASS == 1) return;
    // ---- the Moreau step (prox_h_conj!), vbar, residual maxima (tb_p2)
    real m1 = mx[0], m2 = mx[1], m3 = mx[2];
    real sp_z = 0., sp_v = 0., sp_d = 0.;
    const real gm = ps.gamma;
    auto acc_z = [&](int iz, real dz, real x1) {
        if (SPM >= 1) sp_z += dz * x1;
        if (SPM == 2) sp_d -= dirz[iz] * (dz - sg * (x1 - nig * dz));
    };
    auto acc_v = [&](int iv, real dv, real x2) {
        if (SPM >= 1) sp_v += dv * x2;
        if (SPM == 2) sp_d -= dirv[iv] * (dv - gm * (x2 - nis * dv));
    };
    // v2 (every lane, redundantly) and v1 (lane r < ny)
    const real ws = wz ? zbs : 2. * zbs + -1. * zs_, dzs = zbs - zs_;
    real t2 = (sg * (ws - bw) + v2i) * isg;
    const real nv2 = noprox ? t2 : sg * (t2 - (t2 >= 0. ? t2 : 0.));   // implicit_l.jl:849-851
    const real dv2 = nv2 - v2i;
    const real x2 = (dzs - bdz) + nis * dv2;
    if (r == 0) {
        vo[dm.o2 + i] = nv2;
        m2 = spk_max_nan(m2, fabs(x2));
        const real x1s = dv2 + nig * dzs;                            // xi_1 on s_i
        m1 = spk_max_nan(m1, fabs(x1s));
        acc_v(dm.o2 + i, dv2, x2); acc_z(dm.zs + i, dzs, x1s);
        if (wx) m3 = spk_max_nan(m3, fabs(x2 + x1s));
    }
    if (hy) {
        const real t1 = (sg * wy + v1) * isg;
        real nv1 = t1;
        if (!noprox) {
            const int kind = (dm.elem_kind_mask >> (2 * r)) & 3;       // dual-cone kind of this element (implicit_l.jl:833-845)
            const real pr = kind == 0 ? 0. : (kind == 1 ? (t1 < 0. ? 0. : t1) : (kind == 2 ? (t1 > 0. ? 0. : t1) : t1));
            nv1 = sg * (t1 - pr);
        }
        vo[i * ny + r] = nv1;
        const real dv1 = nv1 - v1;
        const real x2k = dzy + nis * dv1;
        m2 = spk_max_nan(m2, fabs(x2k));
        const real x1k = (dv1 - bbr * dv2) + nig * dzy;
        m1 = spk_max_nan(m1, fabs(x1k));
        acc_v(i * ny + r, dv1, x2k); acc_z(dm.zy + i * ny + r, dzy, x1k);
        if (wx) m3 = spk_max_nan(m3, fabs((x2k - bbr * x2) + x1k));
    }
    // v7: box on (x_i, u_i)
    real ltx = 0., lxx = 0., ltu = 0., lxu = 0.;
    if (hx) {
        const real tt = (sg * wxv + v7x) * isg;
        const real nvv = noprox ? tt : sg * (tt - spk_clamp(tt, dm.x_min, dm.x_max));
        vo[dm.o7 + i * nxu + r] = nvv;
        ltx = nvv - v7x; lxx = dzx + nis * ltx;
        m2 = spk_max_nan(m2, fabs(lxx));
        acc_v(dm.o7 + i * nxu + r, ltx, lxx);
    }
    if (hu) {
        const real tt = (sg * wuv + v7u) * isg;
        const real nvv = noprox ? tt : sg * (tt - spk_clamp(tt, dm.u_min, dm.u_max));
        vo[dm.o7 + i * nxu + nx + r] = nvv;
        ltu = nvv - v7u; lxu = dzu + nis * ltu;
        m2 = spk_max_nan(m2, fabs(lxu));
        acc_v(dm.o7 + i * nxu + nx + r, ltu, lxu);
    }
    // the cones' scalar entries and projections: one lane per cone (GPU), sc_eff of every cone to every lane
    real sc_eff[2 * P::MAXD];
    auto cone_owner_work = [&](const ConeOut &co, int iq, int iA, int iB) {
        vo[iA] = co.pA; vo[iB] = co.pB;
        m2 = spk_max_nan(m2, (real)fabs(co.x2A)); m2 = spk_max_nan(m2, (real)fabs(co.x2B));
        m1 = spk_max_nan(m1, (real)fabs(co.x1));
        acc_v(iA, co.dvA, co.x2A); acc_v(iB, co.dvB, co.x2B); acc_z(iq, co.dzq, co.x1);
        if (wx) m3 = spk_max_nan(m3, (real)fabs((0.5 * co.x2B + 0.5 * co.x2A) + co.x1));
    };
#ifndef SPOCK_HOST_EMU
    if (PASS == 0) {
        real my_ssq = ssq[0];
        SPK_UNROLL for (int q = 1; q < 2 * d; ++q) my_ssq = (r == q) ? ssq[q] : my_ssq;
        const ConeOut co = cone_eval(my_zbq, my_zq, my_vA, my_vB, my_ssq, wz, noprox, ps);
        if (cone_owner) cone_owner_work(co, my_iq, my_iA, my_iB);
        const int team_base = (int)(threadIdx.x & 63u) - r;        // first lane of this team inside its wave
        SPK_UNROLL for (int q = 0; q < 2 * d; ++q) sc_eff[q] = __shfl(co.sc_eff, team_base + q, 64);
    } else
#endif
    {
        SPK_UNROLL for (int q = 0; q < 2 * d; ++q) {
            sc_eff[q] = 1.;
            if (q >= ncones) continue;
            int iq, iA, iB;
            cone_idx(q, iq, iA, iB);
            const ConeOut co = cone_eval(zb[iq], z[iq], v[iA], v[iB], ssq[q], wz, noprox, ps);
            sc_eff[q] = co.sc_eff;
            if (r == q) cone_owner_work(co, iq, iA, iB);
        }
    }
    // per child: the vector entries of (v6; v3, v4, v5) in SOC(nx + nu + 2)   (implicit_l.jl:854-900)
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const int j = t.c0 + c, jn = j - 1;
        const real sc = sc_eff[c];
        if (hx) {
            const real tt = t3[c];
            const real pj = noprox ? tt : sg * (tt - sc * tt);
            vo[dm.o3 + jn * nx + r] = pj;
            const real dv = pj - v3[c], xx = l3[c] + nis * dv;
            m2 = spk_max_nan(m2, fabs(xx));
            acc_v(dm.o3 + jn * nx + r, dv, xx);
            ltx += sq_r * dv; lxx += sq_r * xx;
        }
        if (hu) {
            const real tt = t4[c];
            const real pj = noprox ? tt : sg * (tt - sc * tt);
            vo[dm.o4 + jn * nu + r] = pj;
            const real dv = pj - v4[c], xx = l4[c] + nis * dv;
            m2 = spk_max_nan(m2, fabs(xx));
            acc_v(dm.o4 + jn * nu + r, dv, xx);
            ltu += sr_r * dv; lxu += sr_r * xx;
        }
        // leaf child: the vector entries of (v13; v11, v12) in SOC(nx + 2), v14 in the x box   (implicit_l.jl:752-810)
        if (leaf && hx) {
            const int li = j - dm.nn;
            const real sc2 = sc_eff[d + c];
            const real tt = t11[c];
            const real p11 = noprox ? tt : sg * (tt - sc2 * tt);
            vo[dm.o11 + li * nx + r] = p11;
            const real dv11 = p11 - v11[c], x211 = l11[c] + nis * dv11;
            m2 = spk_max_nan(m2, fabs(x211));
            const real q14 = t14[c];
            const real p14 = noprox ? q14 : sg * (q14 - spk_clamp(q14, dm.x_min, dm.x_max));
            vo[dm.o14 + li * nx + r] = p14;
            const real dv14 = p14 - v14[c], x214 = dl[c] + nis * dv14;
            m2 = spk_max_nan(m2, fabs(x214));
            const real x1 = (dv14 + sn_r * dv11) + nig * dl[c];   // x_leaf = v14 + sqrtQN v11
            m1 = spk_max_nan(m1, fabs(x1));
            acc_v(dm.o11 + li * nx + r, dv11, x211); acc_v(dm.o14 + li * nx + r, dv14, x214); acc_z(dm.zx + j * nx + r, dl[c], x1);
            if (wx) m3 = spk_max_nan(m3, fabs((x214 + sn_r * x211) + x1));
        }
    }
    if (hx) {
        const real x1 = ltx + nig * dzx;
        m1 = spk_max_nan(m1, fabs(x1));
        acc_z(dm.zx + i * nx + r, dzx, x1);
        if (wx) m3 = spk_max_nan(m3, fabs(lxx + x1));
    }
    if (hu) {
        const real x1 = ltu + nig * dzu;
        m1 = spk_max_nan(m1, fabs(x1));
        acc_z(dm.zu + i * nu + r, dzu, x1);
        if (wx) m3 = spk_max_nan(m3, fabs(lxu + x1));
    }
    mx[0] = m1; mx[1] = m2; mx[2] = m3;
    if (SPM >= 1) { sp[0] += sp_z; sp[1] += sp_v; }
    if (SPM == 2) sp[2] += sp_d;
}

// vbar.v2 of NON-LEAF child c of the team's node, recomputed by the parent's team with the arithmetic -- and the summation tree -- of
// the child's own team (fb_node's v2 block; implicit_l.jl:214-222, :849-851): bitwise the value that team stores in this same launch
template <class P, int PASS>
SPK_DEV void fb_child_v2(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z, const real *__restrict__ v, const real *zb,
                         const TeamNode &t, int c, int r, int T, real *red, int slot, const ProxScal &ps, real *out) {
    const int ny = p.ny(), j = t.c0 + c;
    const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? j * ny : 0);
    real term = 0.;
    if (r < ny) {
        const real zbk = zb[dm.zy + j * ny + r], zk = z[dm.zy + j * ny + r];
        term = bb[r] * (2. * zbk + -1. * zk);
    }
    const real zbs = zb[dm.zs + j], zs_ = z[dm.zs + j], v2j = v[dm.o2 + j];
    const real bw = flat_team_sum<PASS>(term, slot, r, T, red);
    if (PASS == 1) return;
    const real ws = 2. * zbs + -1. * zs_;
    const real t2 = (ps.sigma * (ws - bw) + v2j) * ps.inv_sigma;
    if (r == 0) out[c] = ps.sigma * (t2 - (t2 >= 0. ? t2 : 0.));
}

// matrices a workgroup's phase A needs, staged in LDS behind the teams' scratch: U_t (nx x nu) and Rt^-1_t (nu x nu) of the stage of
// the workgroup's FIRST node -- consecutive nodes share a stage except across a stage boundary, where the odd node fetches its rows
// from the pool -- and the S2 projector when all nodes share one (m x m)
SPK_DEV int flat_mats_doubles(int nx, int nu, int m) { return nx * nu + nu * nu + m * m; }

// scratch of one team: [vec: m][ubar: nu][vbar.v2 of the children: d] (+ host emulation: [red: (2 + 3 d) T])
SPK_DEV int flat_scratch_doubles(int nu, int m, int d, int T) {
#ifdef SPOCK_HOST_EMU
    return m + nu + d + (2 + 3 * d) * T;
#else
    return m + nu + d;
#endif
}

// X additionally offers team_sync(): orders the LDS / global accesses of the lanes of ONE WAVE (a team is T <= 64 lanes of one
// wave, so the exchanges inside a team need no workgroup barrier); the emulation's each() calls are ordered anyway.
template <class P, int MODE_, class X>
SPK_DEV void flat_body(X &x, const KArgs &a, int inst, int g) {
    constexpr bool SP_SUMS = (MODE_ == MODE_FLAT_B_SP);
    constexpr bool DO_B = (MODE_ != MODE_FLAT_A);
    const Dims &dm = a.dm;
    const P p(dm);
    const unsigned flags = a.flags;
    Ctl &ctl = a.ctl[inst];
    if (!(flags & F_NO_CTL) && ctl.done) return;
    const real *mats = a.mats + (long)inst * a.mats_stride;   // (per-instance pools: spock_update_dynamics_batch)
    const MatPtr M = (MatPtr)mats;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nthr = x.nthreads();
    const int T = a.team, nteams = nthr / T, m = ny + 2 * d;
    const int stride = flat_scratch_doubles(nu, m, d, T);
    real *scr = x.lds();
    const real *z = a.z_in + (long)inst * a.stride_z;
    const real *v = a.v_in + (long)inst * a.stride_v;
    real *zo = a.z_out + (long)inst * a.stride_z;
    real *vo = a.v_out + (long)inst * a.stride_v;
    const real gamma = ctl.gamma, sigma = ctl.sigma;
    const int node0 = g * nteams;            // one round: team tm works on node node0 + tm

    ProxScal ps;
    ps.sigma = sigma; ps.inv_sigma = 1. / sigma; ps.inv_sigma_neg = -1. / sigma; ps.inv_gamma_neg = -1. / gamma; ps.gamma = gamma;
    ps.want_xi = (ctl.log_mode == 2 && ctl.alg == 1);
    // team scratch: S = [vec | ubar] for ta_p1 / ta_p2 (their `red` rows are not used here), v2x behind them, the emulation's red rows last
    auto S_of = [&](int tm) SPK_LAMBDA_INLINE { return scr + tm * stride; };
    auto v2x_of = [&](int tm) SPK_LAMBDA_INLINE { return scr + tm * stride + m + nu; };
    auto red_of = [&](int tm) SPK_LAMBDA_INLINE { return scr + tm * stride + m + nu + d; };

    // ---- phase A of the team's node on the iterate (zi, vi): its part of the next zbar (y_i, s / tau of the children, s_1) to zn,
    //      c_i -> zn.x_i, e_i -> zn.u_i, q_leaf = -xbar_leaf -> zn.x_leaf
    real *lmat = scr + nteams * stride;                  // [U_t | Rt^-1_t | LS] of the stage of the workgroup's first node
    const int stage0 = flat_team_node(dm, d, node0).stage;
    auto stage_mats = [&]() SPK_LAMBDA_INLINE {          // issued first thing: one coalesced copy, overlapping the node loads of part 1
        x.each([&](int tid, TState<P> &) {
            const real *sU = mats + dm.mU + (long)stage0 * nx * nu, *sR = mats + dm.mRinv + (long)stage0 * nu * nu, *sL = mats + dm.mLs;
            for (int k = tid; k < nx * nu; k += nthr) lmat[k] = sU[k];
            for (int k = tid; k < nu * nu; k += nthr) lmat[nx * nu + k] = sR[k];
            if (dm.nRm <= 1) for (int k = tid; k < m * m; k += nthr) lmat[nx * nu + nu * nu + k] = sL[k];
        });
    };
#ifndef FLAT_EXP
#define FLAT_EXP 0
#endif
    if (FLAT_EXP == 1 && MODE_ == MODE_FLAT_A) return;                       // experiment: launch + control only
    if (MODE_ == MODE_FLAT_A || MODE_ == MODE_FLAT_BA) stage_mats();
    auto phase_a = [&](const real *zi, const real *vi, real *zn, bool use_v2x) SPK_LAMBDA_INLINE {
        x.each([&](int tid, TState<P> &s) {
            const int tm = tid / T, r = tid - tm * T;
            if (tm >= nteams) return;
            const TeamNode tn = flat_team_node(dm, d, node0 + tm);
            if (!tn.active) return;
            const real *v2x = (use_v2x && !tn.kids_leaf) ? v2x_of(tm) : nullptr;
            ta_p1(p, dm, M, zi, vi, zn, tn, r, T, S_of(tm), zn + dm.zx + (long)tn.c0 * nx, gamma, flags & ~F_RAW_LT, s.tlr, v2x);
        });
        if (FLAT_EXP == 2 && MODE_ == MODE_FLAT_A) return;                   // experiment: part 1 only
        x.barrier();    // the staged matrices come from every wave of the workgroup
        x.each([&](int tid, TState<P> &s) {
            const int tm = tid / T, r = tid - tm * T;
            if (tm >= nteams) return;
            const TeamNode tn = flat_team_node(dm, d, node0 + tm);
            if (!tn.active) return;
            const bool st = (tn.stage == stage0);
            ta_p2(p, dm, M, zn, tn, r, T, S_of(tm), zn + dm.zx + (long)tn.node * nx, zn + dm.zu + (long)tn.node * nu, s.tlr,
                  st ? lmat : nullptr, st ? lmat + nx * nu : nullptr, dm.nRm <= 1 ? lmat + nx * nu + nu * nu : nullptr);
        });
    };

    if (MODE_ == MODE_FLAT_A) { phase_a(z, v, zo, false); return; }

    // ---- phase B on the completed zbar (x_i, u_i, x_leaf from zo) + the workgroup's residual maxima / sums
    double *partials = a.partials + (long)inst * a.flat_chunks * kPartialStride;
    if (DO_B) {
        x.each([&](int, TState<P> &s) { s.mx3[0] = s.mx3[1] = s.mx3[2] = 0.; s.sp3[0] = s.sp3[1] = s.sp3[2] = 0.; });
        const real *dz_ = (SP_SUMS && a.dir) ? a.dir + (long)inst * a.stride_z : nullptr;
        auto node_b = [&](auto pass_tag) SPK_LAMBDA_INLINE {
            constexpr int PASS = decltype(pass_tag)::value;
            x.each([&](int tid, TState<P> &s) {
                const int tm = tid / T, r = tid - tm * T;
                if (tm >= nteams) return;
                const TeamNode tn = flat_team_node(dm, d, node0 + tm);
                if (!tn.active) return;
                real *red = red_of(tm);
                if constexpr (SP_SUMS) {
                    if (a.sp_mode == 2) fb_node<P, 2, PASS>(p, dm, M, z, v, zo, vo, tn, r, T, red, ps, flags, s.mx3, s.sp3, dz_, dz_ + dm.nz);
                    else fb_node<P, 1, PASS>(p, dm, M, z, v, zo, vo, tn, r, T, red, ps, flags, s.mx3, s.sp3);
                } else fb_node<P, 0, PASS>(p, dm, M, z, v, zo, vo, tn, r, T, red, ps, flags, s.mx3);
                if (MODE_ == MODE_FLAT_BA && !tn.kids_leaf) {
                    SPK_UNROLL for (int c = 0; c < d; ++c) fb_child_v2<P, PASS>(p, dm, M, z, v, zo, tn, c, r, T, red, 2 + 2 * d + c, ps, v2x_of(tm));
                }
            });
        };
#ifdef SPOCK_HOST_EMU
        node_b(std::integral_constant<int, 1>());
        node_b(std::integral_constant<int, 2>());
#else
        node_b(std::integral_constant<int, 0>());
#endif
        x.each([&](int tid, TState<P> &s) { x.stash3(tid, s.mx3); if (SP_SUMS) x.stash_sum3(tid, s.sp3); });
        x.reduce3();          // (starts and ends with a barrier: vbar of this workgroup's nodes is visible to all its lanes behind it)
        x.each([&](int tid, TState<P> &) { if (tid < 3) partials[(long)g * kPartialStride + tid] = x.red3(tid); });
        if (SP_SUMS) {
            x.reduce_sum3();
            x.each([&](int tid, TState<P> &) { if (tid < 3) partials[(long)g * kPartialStride + 4 + tid] = x.sum3(tid); });
            if (a.sp_mode == 2) {   // the termination residuals of the direction d itself (K1 step: dp = d), spock_sp.hpp: sp_ctl_cd
                const real *dd_ = a.dir + (long)inst * a.stride_z;
                x.each([&](int tid, TState<P> &) {
                    double acc[4] = {0., 0., 0., 0.};
                    const int tm = tid / T, r = tid - tm * T;
                    if (tm < nteams) {
                        const TeamNode tn = flat_team_node(dm, d, node0 + tm);
                        tlop(p, dm, M, nullptr, dd_, nullptr, tn, r, gamma, sigma, acc);
                    }
                    const double mx[3] = {acc[2], acc[3], 0.};
                    x.stash3(tid, mx);
                });
                x.reduce3();
                x.each([&](int tid, TState<P> &) {
                    if (tid == 0) { partials[(long)g * kPartialStride + 3] = x.red3(0); partials[(long)g * kPartialStride + 7] = x.red3(1); }
                });
            }
        }
    }
    if (MODE_ == MODE_FLAT_BA) {
        // the first half of the NEXT iteration on (zbar, vbar): its part of the next zbar goes to the third z buffer -- other
        // workgroups of this launch still read z
        real *zn = a.z_next + (long)inst * a.stride_z;
        phase_a(zo, vo, zn, true);
    }
}

}  // namespace spock
