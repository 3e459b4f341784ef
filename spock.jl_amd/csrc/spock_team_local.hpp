// spock_team_local.hpp -- lane-team versions of the node-local phases for LARGE nx / nu (the LDS-tiled path of
// BASELINE.json configs 3-5): T lanes per node, lane r owns element r of every per-node vector (x, u, y, v3_c, v4_c,
// v11_c, v14_c, ...), so nothing per-node lives in a thread-private array.  The per-thread version (spock_device.hpp)
// keeps whole vectors per thread, which for nx = 20 or 64 means kilobytes of scratch memory per lane.
//
// Scope: shared DIAGONAL cost square roots (every shipped problem family) and risk cones without second-order parts;
// anything else stays on the per-thread path.  With diagonal costs every cost mat-vec is element-wise, so a lane needs
// other lanes' data only for
//   phase A: the S2 projector rows (all of [y; s_ch; tau_ch]) and U ubar / Rt^-1 ubar (all of ubar)   -> 1 LDS exchange
//   phase B: b'y, and the squared norms of the second-order-cone blocks                               -> 1 LDS exchange
//   LOP    : nothing (per-lane partial sums, reduced per workgroup at the end)
// Reference functions restated: L_transpose! (implicit_l.jl:320-449), projection_S2! (:688-727), L! (:177-318),
// prox_h_conj! (:911-951) with the cone / box projections (:752-909, risk_measures.jl:14-71, constraints.jl:24-54),
// should_terminate! residuals (cp.jl:54-123), spock_dot (implicit_l.jl:485-516).
#pragma once
#include "spock_device.hpp"

namespace spock {

struct TeamNode {   // the node a team works on in the current round
    int li, lvl, pos, node, c0, stage;
    bool active, kids_leaf, kids_gleaf;
};

SPK_DEV TeamNode team_node(const Dims &dm, int d, int li, int nl_grp, int root_stage, int root_pos, int stages, bool crown) {
    TeamNode t;
    t.active = li < nl_grp;
    if (!t.active) li = 0;
    int lvl = 0, first = 0, cnt = 1;
    while (li >= first + cnt) { first += cnt; cnt *= d; ++lvl; }
    t.li = li; t.lvl = lvl; t.pos = li - first;
    t.node = dm.first[root_stage + lvl] + root_pos * cnt + t.pos;
    t.c0 = t.node * d + 1;
    t.stage = root_stage + lvl;
    t.kids_gleaf = (lvl == stages - 2);
    t.kids_leaf = t.kids_gleaf && !crown && (root_stage + stages == dm.N);
    return t;
}

// team scratch in LDS (doubles): [vec: m][ub: nu][red: (2 + 2 d) * T]
SPK_DEV int team_scratch_doubles(int nu, int m, int d, int T) { return m + nu + (2 + 2 * d) * T; }

// ---------------------------------------------------------------------------------------------------
// phase A, part 1: zbar0 = (-gamma) L'v + z on the lane's elements; publish [y; s_ch; tau_ch] and ubar
template <class P>
// (S, XL restrict-qualified: what the lanes publish does not alias the iterate they read, so the loads of the u and [y; s; tau] parts
//  are not held back behind the stores of the x part)
SPK_DEV void ta_p1(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z, const real *__restrict__ v,
                   real *__restrict__ zo, const TeamNode &t, int r, int T, real *__restrict__ S, real *__restrict__ XL, real gamma, unsigned flags,
                   TLReg<P> &g, const real *v2x = nullptr) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu, m = ny + 2 * d;
    const int i = t.node;
    const real mg = -gamma;
    const bool raw = (flags & F_RAW_LT) != 0;
    const MatPtr SQ = M + dm.mSqQ, SR = M + dm.mSqR, SN = M + dm.mSqQN, bb = M + dm.mb + (dm.nRm > 1 ? t.node * p.ny() : 0);
    real *vec = S, *ubs = S + m;
    if (r < nx) {   // x_i[r] <- v7 + sum_j sqrtQ v3_j  (implicit_l.jl:351-373), diagonal sqrtQ
        real acc = v[dm.o7 + i * nxu + r];
        SPK_UNROLL for (int c = 0; c < d; ++c) acc += SQ[r + r * nx] * v[dm.o3 + (t.c0 + c - 1) * nx + r];
        g.xb = mg * acc + z[dm.zx + i * nx + r];
        if (raw) zo[dm.zx + i * nx + r] = g.xb;
        if (t.kids_leaf)   // leaves: x_j <- v14 + sqrtQN v11 (implicit_l.jl:376-396); q_leaf = -xbar_leaf goes to the sweep
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const int j = t.c0 + c, li = j - dm.nn;
                const real xl = mg * (v[dm.o14 + li * nx + r] + SN[r + r * nx] * v[dm.o11 + li * nx + r]) + z[dm.zx + j * nx + r];
                if (raw) zo[dm.zx + j * nx + r] = xl;
                else if (XL) XL[(t.pos * d + c) * nx + r] = -xl;
            }
    }
    if (r < nu) {   // u_i[r] (implicit_l.jl:399-420)
        real acc = v[dm.o7 + i * nxu + nx + r];
        SPK_UNROLL for (int c = 0; c < d; ++c) acc += SR[r + r * nu] * v[dm.o4 + (t.c0 + c - 1) * nu + r];
        g.ub = mg * acc + z[dm.zu + i * nu + r];
        if (raw) zo[dm.zu + i * nu + r] = g.ub; else ubs[r] = g.ub;
    }
    for (int e = r; e < m; e += T) {   // [y_i ; s_children ; tau_children]  (implicit_l.jl:423-448)
        // every lane first forms the indices of its three operands and loads them -- one round trip to memory for the whole vector --
        // then applies its entry's formula (as three divergent branches with their own loads the entries waited for three)
        const bool isy = e < ny, iss = !isy && e < ny + d;
        const int cc = isy ? 0 : (iss ? e - ny : e - ny - d), j = t.c0 + cc;
        const bool use_v2x = iss && v2x && t.kids_gleaf, sleaf = iss && j >= dm.nn;
        const int iA = isy ? i * ny + e : (iss ? (sleaf ? dm.o12 + (j - dm.nn) : dm.o2 + j) : dm.o5 + j - 1);
        const int iB = isy ? dm.o2 + i : (iss ? (sleaf ? dm.o13 + (j - dm.nn) : dm.o2 + j) : dm.o6 + j - 1);
        const int iZ = isy ? dm.zy + i * ny + e : (iss ? dm.zs + j : dm.zt + j - 1);
        const real a = v[iA], b = v[iB], zz = z[iZ];
        real val;
        if (isy) val = mg * (a - bb[e] * b) + zz;
        else if (iss) {
            const real ls = use_v2x ? v2x[t.pos * d + cc]   // crown workgroup of the fused B+A kernel / flat B+A: recomputed by the parent's team
                                    : (sleaf ? 0.5 * a + 0.5 * b : a);
            val = mg * ls + zz;
        } else val = mg * (0.5 * a + 0.5 * b) + zz;
        if (raw) {
            if (e < ny) zo[dm.zy + i * ny + e] = val;
            else if (e < ny + d) zo[dm.zs + t.c0 + (e - ny)] = val;
            else zo[dm.zt + t.c0 + (e - ny - d) - 1] = val;
        } else vec[e] = val;
    }
    if (i == 0 && r == 0) zo[dm.zs] = raw ? mg * v[dm.o2] + z[dm.zs] : (mg * v[dm.o2] + z[dm.zs]) - gamma;   // s_1 (implicit_l.jl:741)
}

// phase A, part 2: S2 projector rows, c_i = U ubar - xbar, e_i = Rt^-1 ubar
// lmU / lmRi / lmLS: the stage's U_t, Rt^-1_t and the S2 projector staged in LDS by the caller (flat kernels: one coalesced copy per
// workgroup), or nullptr -> rows fetched from the matrix pool element by element (eight loads in flight: without the unrolling every
// iteration of these mat-vec loops waits for its own load, nu dependent round trips to L2 per node)
template <class MP, class VP>
SPK_DEV real tl_rowdot(MP row, int stride, VP vec, int n) {
    real a = 0.;
    SPK_UNROLL8 for (int k = 0; k < n; ++k) a += row[(long)k * stride] * vec[k];
    return a;
}
template <class P>
SPK_DEV void ta_p2(const P &p, const Dims &dm, MatPtr M, real *__restrict__ zo, const TeamNode &t, int r, int T,
                   const real *S, real *cq, real *ce, const TLReg<P> &g, const real *lmU = nullptr, const real *lmRi = nullptr,
                   const real *lmLS = nullptr) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), m = ny + 2 * d;
    const int i = t.node;
    const real *vec = S, *ubs = S + m;
    const MatPtr LS = M + dm.mLs + (dm.nRm > 1 ? i * m * m : 0);
    for (int e = r; e < m; e += T) {   // implicit_l.jl:708-726
        const real a = lmLS ? tl_rowdot(lmLS + e, m, vec, m) : tl_rowdot(LS + e, m, vec, m);
        if (e < ny) zo[dm.zy + i * ny + e] = a;
        else if (e < ny + d) zo[dm.zs + t.c0 + (e - ny)] = a;
        else zo[dm.zt + t.c0 + (e - ny - d) - 1] = a;
    }
    if (r < nx) {
        const MatPtr Ut = M + dm.mU + t.stage * nx * nu;
        const real a = lmU ? tl_rowdot(lmU + r, nx, ubs, nu) : tl_rowdot(Ut + r, nx, ubs, nu);
        cq[r] = a - g.xb;
    }
    if (r < nu) {
        const MatPtr Ri = M + dm.mRinv + t.stage * nu * nu;
        ce[r] = lmRi ? tl_rowdot(lmRi + r, nu, ubs, nu) : tl_rowdot(Ri + r, nu, ubs, nu);
    }
}

// ---------------------------------------------------------------------------------------------------
// phase B, part 1: w, dz, the pre-prox value of every owned v entry; publish the partial sums of b'y and of the
// second-order-cone blocks.  xi / ui / xleaf: this lane's element of x_i, u_i and of the leaf children's x.
template <class P>
SPK_DEV void tb_p1(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z, const real *__restrict__ v,
                   const real *zb, const TeamNode &t, int r, int T, real *S, const real *xsrc, const real *usrc,
                   const real *xlsrc, int xl_stride, const ProxScal &ps, unsigned flags, TLReg<P> &g) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), m = ny + 2 * d;
    const int i = t.node;
    const bool wz = (flags & F_W_IS_ZBAR) != 0, noprox = (flags & F_NO_PROX) != 0;
    const real sg = ps.sigma, isg = noprox ? 1. : ps.inv_sigma;
    const MatPtr SQ = M + dm.mSqQ, SR = M + dm.mSqR, SN = M + dm.mSqQN, bb = M + dm.mb + (dm.nRm > 1 ? t.node * p.ny() : 0);
    real *red = S + m + nu;   // [(2 + 2 d)][T]
    real pb0 = 0., pb1 = 0.;
    if (r < ny) {
        const real zbk = zb[dm.zy + i * ny + r], zk = z[dm.zy + i * ny + r];
        g.wy = wz ? zbk : 2. * zbk + -1. * zk; g.dzy = zbk - zk;
        pb0 = bb[r] * g.wy; pb1 = bb[r] * g.dzy;
    }
    red[0 * T + r] = pb0; red[1 * T + r] = pb1;
    real sq[2 * P::MAXD];
    SPK_UNROLL for (int c = 0; c < 2 * d; ++c) sq[c] = 0.;
    if (r < nx) {
        const real xi = xsrc[r], zi = z[dm.zx + i * nx + r];
        g.wx = wz ? xi : 2. * xi + -1. * zi; g.dzx = xi - zi; g.xb = xi;
        SPK_UNROLL for (int c = 0; c < d; ++c) {   // v3_j = sqrtQ x_i (implicit_l.jl:226-235)
            const int jn = t.c0 + c - 1;
            g.t3[c] = (sg * (SQ[r + r * nx] * g.wx) + v[dm.o3 + jn * nx + r]) * isg;
            g.l3[c] = SQ[r + r * nx] * g.dzx;
            sq[c] += g.t3[c] * g.t3[c];
        }
        if (t.kids_leaf)
            SPK_UNROLL for (int c = 0; c < d; ++c) {   // v11 = sqrtQN x_leaf, v14 = x_leaf (implicit_l.jl:288-317)
                const int j = t.c0 + c, li = j - dm.nn;
                const real xl = xlsrc[c * xl_stride + r], zk = z[dm.zx + j * nx + r];
                g.wl[c] = wz ? xl : 2. * xl + -1. * zk; g.dl[c] = xl - zk;
                g.t11[c] = (sg * (SN[r + r * nx] * g.wl[c]) + v[dm.o11 + li * nx + r]) * isg;
                g.l11[c] = SN[r + r * nx] * g.dl[c];
                g.t14[c] = (sg * g.wl[c] + v[dm.o14 + li * nx + r]) * isg;
                sq[d + c] += g.t11[c] * g.t11[c];
            }
    }
    if (r < nu) {
        const real ui = usrc[r], zi = z[dm.zu + i * nu + r];
        g.wu = wz ? ui : 2. * ui + -1. * zi; g.dzu = ui - zi; g.ub = ui;
        SPK_UNROLL for (int c = 0; c < d; ++c) {   // v4_j = sqrtR u_i (implicit_l.jl:239-248)
            const int jn = t.c0 + c - 1;
            g.t4[c] = (sg * (SR[r + r * nu] * g.wu) + v[dm.o4 + jn * nu + r]) * isg;
            g.l4[c] = SR[r + r * nu] * g.dzu;
            sq[c] += g.t4[c] * g.t4[c];
        }
    }
    SPK_UNROLL for (int c = 0; c < 2 * d; ++c) red[(2 + c) * T + r] = sq[c];
}

// phase B, part 2: finish the Moreau step (prox_h_conj!), store vbar, accumulate the residual maxima
// SPM: the SP driver's inner products with the residual of this evaluation, as in b_local (spock_device.hpp)
template <class P, int SPM = 0>
SPK_DEV void tb_p2(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z, const real *__restrict__ v,
                   const real *zb, real *__restrict__ vo, const TeamNode &t, int r, int T, const real *S,
                   const ProxScal &ps, unsigned flags, const TLReg<P> &g, real *mx, real *sp = nullptr,
                   const real *dirz = nullptr, const real *dirv = nullptr) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu, m = ny + 2 * d;
    const int i = t.node;
    const bool wz = (flags & F_W_IS_ZBAR) != 0, noprox = (flags & F_NO_PROX) != 0, wx = ps.want_xi;
    const real sg = ps.sigma, isg = noprox ? 1. : ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg;
    const MatPtr SQ = M + dm.mSqQ, SR = M + dm.mSqR, SN = M + dm.mSqQN, bb = M + dm.mb + (dm.nRm > 1 ? t.node * p.ny() : 0);
    const real *red = S + m + nu;
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
    // ---- v2 (every lane, redundantly: a handful of flops) and v1 (lane r < ny)
    real bw = 0., bdz = 0.;
    for (int k = 0; k < ny; ++k) { bw += red[0 * T + k]; bdz += red[1 * T + k]; }
    const real zbs = zb[dm.zs + i], zs_ = z[dm.zs + i];
    const real ws = wz ? zbs : 2. * zbs + -1. * zs_, dzs = zbs - zs_;
    real t2 = (sg * (ws - bw) + v[dm.o2 + i]) * isg;
    const real nv2 = noprox ? t2 : sg * (t2 - (t2 >= 0. ? t2 : 0.));   // implicit_l.jl:849-851
    const real dv2 = nv2 - v[dm.o2 + i];
    const real x2 = (dzs - bdz) + nis * dv2;
    if (r == 0) {
        vo[dm.o2 + i] = nv2;
        m2 = spk_max_nan(m2, fabs(x2));
        const real x1s = dv2 + nig * dzs;                            // xi_1 on s_i
        m1 = spk_max_nan(m1, fabs(x1s));
        acc_v(dm.o2 + i, dv2, x2); acc_z(dm.zs + i, dzs, x1s);
        if (wx) m3 = spk_max_nan(m3, fabs(x2 + x1s));
    }
    if (r < ny) {
        const real t1 = (sg * g.wy + v[i * ny + r]) * isg;
        real nv1 = t1;
        if (!noprox) {
            const int kind = (dm.elem_kind_mask >> (2 * r)) & 3;       // dual-cone kind of this element (implicit_l.jl:833-845)
            const real pr = kind == 0 ? 0. : (kind == 1 ? (t1 < 0. ? 0. : t1) : (kind == 2 ? (t1 > 0. ? 0. : t1) : t1));
            nv1 = sg * (t1 - pr);
        }
        vo[i * ny + r] = nv1;
        const real dv1 = nv1 - v[i * ny + r];
        const real x2k = g.dzy + nis * dv1;
        m2 = spk_max_nan(m2, fabs(x2k));
        const real x1k = (dv1 - bb[r] * dv2) + nig * g.dzy;
        m1 = spk_max_nan(m1, fabs(x1k));
        acc_v(i * ny + r, dv1, x2k); acc_z(dm.zy + i * ny + r, g.dzy, x1k);
        if (wx) m3 = spk_max_nan(m3, fabs((x2k - bb[r] * x2) + x1k));
    }
    // ---- v7: box on (x_i, u_i)
    real ltx = 0., lxx = 0., ltu = 0., lxu = 0.;
    if (r < nx) {
        const real vin = v[dm.o7 + i * nxu + r];
        const real tt = (sg * g.wx + vin) * isg;
        const real nvv = noprox ? tt : sg * (tt - spk_clamp(tt, dm.x_min, dm.x_max));
        vo[dm.o7 + i * nxu + r] = nvv;
        ltx = nvv - vin; lxx = g.dzx + nis * ltx;
        m2 = spk_max_nan(m2, fabs(lxx));
        acc_v(dm.o7 + i * nxu + r, ltx, lxx);
    }
    if (r < nu) {
        const real vin = v[dm.o7 + i * nxu + nx + r];
        const real tt = (sg * g.wu + vin) * isg;
        const real nvv = noprox ? tt : sg * (tt - spk_clamp(tt, dm.u_min, dm.u_max));
        vo[dm.o7 + i * nxu + nx + r] = nvv;
        ltu = nvv - vin; lxu = g.dzu + nis * ltu;
        m2 = spk_max_nan(m2, fabs(lxu));
        acc_v(dm.o7 + i * nxu + nx + r, ltu, lxu);
    }
    // ---- per child: (v6; v3, v4, v5) in SOC(nx + nu + 2)   (implicit_l.jl:854-900, :928-933)
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const int j = t.c0 + c, jn = j - 1;
        const real zbt = zb[dm.zt + jn], zt_ = z[dm.zt + jn];
        const real wt = wz ? zbt : 2. * zbt + -1. * zt_, dzt = zbt - zt_;
        real t6 = (sg * (0.5 * wt) + v[dm.o6 + jn]) * isg, t5 = (sg * (0.5 * wt) + v[dm.o5 + jn]) * isg;
        real keep = 1., tnew = t6, sc = 1.;      // projected (t; x) = (tnew; sc * x) unless keep
        if (!noprox) {
            t5 -= 0.5; t6 += 0.5;
            real ss = t5 * t5;
            for (int k = 0; k < T; ++k) ss += red[(2 + c) * T + k];
            const real x_norm = sqrt(ss);
            if (x_norm <= t6) { keep = 1.; }
            else if (x_norm <= -t6) { keep = 0.; tnew = 0.; sc = 0.; }
            else { keep = 0.; tnew = (t6 + x_norm) / 2.; sc = tnew / x_norm; }
        }
        const real p6 = noprox ? t6 : sg * (t6 - (keep != 0. ? t6 : tnew));
        const real p5 = noprox ? t5 : sg * (t5 - (keep != 0. ? t5 : sc * t5));
        const real dv6 = p6 - v[dm.o6 + jn], dv5 = p5 - v[dm.o5 + jn];
        const real x26 = 0.5 * dzt + nis * dv6, x25 = 0.5 * dzt + nis * dv5;
        if (r == 0) {
            vo[dm.o6 + jn] = p6; vo[dm.o5 + jn] = p5;
            m2 = spk_max_nan(m2, fabs(x26)); m2 = spk_max_nan(m2, fabs(x25));
            const real x1t = (0.5 * dv5 + 0.5 * dv6) + nig * dzt;   // xi_1 on tau_j
            m1 = spk_max_nan(m1, fabs(x1t));
            acc_v(dm.o6 + jn, dv6, x26); acc_v(dm.o5 + jn, dv5, x25); acc_z(dm.zt + jn, dzt, x1t);
            if (wx) m3 = spk_max_nan(m3, fabs((0.5 * x25 + 0.5 * x26) + x1t));
        }
        if (r < nx) {
            const real tt = g.t3[c];
            const real pj = noprox ? tt : sg * (tt - (keep != 0. ? tt : sc * tt));
            vo[dm.o3 + jn * nx + r] = pj;
            const real dv = pj - v[dm.o3 + jn * nx + r], xx = g.l3[c] + nis * dv;
            m2 = spk_max_nan(m2, fabs(xx));
            acc_v(dm.o3 + jn * nx + r, dv, xx);
            ltx += SQ[r + r * nx] * dv; lxx += SQ[r + r * nx] * xx;
        }
        if (r < nu) {
            const real tt = g.t4[c];
            const real pj = noprox ? tt : sg * (tt - (keep != 0. ? tt : sc * tt));
            vo[dm.o4 + jn * nu + r] = pj;
            const real dv = pj - v[dm.o4 + jn * nu + r], xx = g.l4[c] + nis * dv;
            m2 = spk_max_nan(m2, fabs(xx));
            acc_v(dm.o4 + jn * nu + r, dv, xx);
            ltu += SR[r + r * nu] * dv; lxu += SR[r + r * nu] * xx;
        }
        // ---- leaf child: (v13; v11, v12) in SOC(nx + 2), v14 in the x box   (implicit_l.jl:752-810, :934-939)
        if (t.kids_leaf) {
            const int li = j - dm.nn;
            const real zbs2 = zb[dm.zs + j], zs2 = z[dm.zs + j];
            const real ws2 = wz ? zbs2 : 2. * zbs2 + -1. * zs2, dzs2 = zbs2 - zs2;
            real t13 = (sg * (0.5 * ws2) + v[dm.o13 + li]) * isg, t12 = (sg * (0.5 * ws2) + v[dm.o12 + li]) * isg;
            real keep2 = 1., tnew2 = t13, sc2 = 1.;
            if (!noprox) {
                t12 -= 0.5; t13 += 0.5;
                real ss = t12 * t12;
                for (int k = 0; k < T; ++k) ss += red[(2 + d + c) * T + k];
                const real x_norm = sqrt(ss);
                if (x_norm <= t13) { keep2 = 1.; }
                else if (x_norm <= -t13) { keep2 = 0.; tnew2 = 0.; sc2 = 0.; }
                else { keep2 = 0.; tnew2 = (t13 + x_norm) / 2.; sc2 = tnew2 / x_norm; }
            }
            const real p13 = noprox ? t13 : sg * (t13 - (keep2 != 0. ? t13 : tnew2));
            const real p12 = noprox ? t12 : sg * (t12 - (keep2 != 0. ? t12 : sc2 * t12));
            const real dv13 = p13 - v[dm.o13 + li], dv12 = p12 - v[dm.o12 + li];
            const real x213 = 0.5 * dzs2 + nis * dv13, x212 = 0.5 * dzs2 + nis * dv12;
            if (r == 0) {
                vo[dm.o13 + li] = p13; vo[dm.o12 + li] = p12;
                m2 = spk_max_nan(m2, fabs(x213)); m2 = spk_max_nan(m2, fabs(x212));
                const real x1s2 = (0.5 * dv12 + 0.5 * dv13) + nig * dzs2;   // xi_1 on s_leaf
                m1 = spk_max_nan(m1, fabs(x1s2));
                acc_v(dm.o13 + li, dv13, x213); acc_v(dm.o12 + li, dv12, x212); acc_z(dm.zs + j, dzs2, x1s2);
                if (wx) m3 = spk_max_nan(m3, fabs((0.5 * x212 + 0.5 * x213) + x1s2));
            }
            if (r < nx) {
                const real tt = g.t11[c];
                const real p11 = noprox ? tt : sg * (tt - (keep2 != 0. ? tt : sc2 * tt));
                vo[dm.o11 + li * nx + r] = p11;
                const real dv11 = p11 - v[dm.o11 + li * nx + r], x211 = g.l11[c] + nis * dv11;
                m2 = spk_max_nan(m2, fabs(x211));
                const real t14 = g.t14[c];
                const real p14 = noprox ? t14 : sg * (t14 - spk_clamp(t14, dm.x_min, dm.x_max));
                vo[dm.o14 + li * nx + r] = p14;
                const real dv14 = p14 - v[dm.o14 + li * nx + r], x214 = g.dl[c] + nis * dv14;
                m2 = spk_max_nan(m2, fabs(x214));
                const real x1 = (dv14 + SN[r + r * nx] * dv11) + nig * g.dl[c];   // x_leaf = v14 + sqrtQN v11
                m1 = spk_max_nan(m1, fabs(x1));
                acc_v(dm.o11 + li * nx + r, dv11, x211); acc_v(dm.o14 + li * nx + r, dv14, x214); acc_z(dm.zx + j * nx + r, g.dl[c], x1);
                if (wx) m3 = spk_max_nan(m3, fabs((x214 + SN[r + r * nx] * x211) + x1));
            }
        }
    }
    if (r < nx) {
        const real x1 = ltx + nig * g.dzx;
        m1 = spk_max_nan(m1, fabs(x1));
        acc_z(dm.zx + i * nx + r, g.dzx, x1);
        if (wx) m3 = spk_max_nan(m3, fabs(lxx + x1));
    }
    if (r < nu) {
        const real x1 = ltu + nig * g.dzu;
        m1 = spk_max_nan(m1, fabs(x1));
        acc_z(dm.zu + i * nu + r, g.dzu, x1);
        if (wx) m3 = spk_max_nan(m3, fabs(lxu + x1));
    }
    mx[0] = m1; mx[1] = m2; mx[2] = m3;
    if (SPM >= 1) { sp[0] += sp_z; sp[1] += sp_v; }
    if (SPM == 2) sp[2] += sp_d;
}

// ---------------------------------------------------------------------------------------------------
// LOP for lane teams (see lop_local in spock_sp.hpp): acc[0] += <a, P b>, acc[1] += <a, P b2>, acc[2], acc[3] maxima
template <class P>
SPK_DEV void tlop(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ A, const real *__restrict__ Bv,
                  const real *__restrict__ B2, const TeamNode &t, int r, real gamma, real sigma, double *acc) {
    if (!t.active) return;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny(), nxu = nx + nu, nz = dm.nz;
    const int i = t.node;
    const real nig = -1. / gamma, nis = -1. / sigma;
    const MatPtr SQ = M + dm.mSqQ, SR = M + dm.mSqR, SN = M + dm.mSqQN, bb = M + dm.mb + (dm.nRm > 1 ? t.node * p.ny() : 0);
    const int nb = B2 ? 2 : 1;
    for (int which = 0; which < nb; ++which) {
        const real *b = which ? B2 : Bv, *bz = b, *bv = b + nz;
        const real *az = A, *av = A ? A + nz : nullptr;
        real sum = 0., m1 = acc[2], m2 = acc[3];
        if (r < nx) {
            const real bx = bz[dm.zx + i * nx + r], b7 = bv[dm.o7 + i * nxu + r];
            real lt = b7;
            if (A) sum += av[dm.o7 + i * nxu + r] * (b7 - sigma * bx);
            m2 = spk_max_nan(m2, fabs(bx + nis * b7));
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const int jn = t.c0 + c - 1;
                const real b3 = bv[dm.o3 + jn * nx + r], l3 = SQ[r + r * nx] * bx;
                lt += SQ[r + r * nx] * b3;
                if (A) sum += av[dm.o3 + jn * nx + r] * (b3 - sigma * l3);
                m2 = spk_max_nan(m2, fabs(l3 + nis * b3));
                if (t.kids_leaf) {
                    const int j = t.c0 + c, li = j - dm.nn;
                    const real bxl = bz[dm.zx + j * nx + r], b11 = bv[dm.o11 + li * nx + r], b14 = bv[dm.o14 + li * nx + r];
                    const real l11 = SN[r + r * nx] * bxl, ltl = b14 + SN[r + r * nx] * b11;
                    if (A) sum += az[dm.zx + j * nx + r] * (bxl - gamma * ltl) + av[dm.o11 + li * nx + r] * (b11 - sigma * l11) + av[dm.o14 + li * nx + r] * (b14 - sigma * bxl);
                    m1 = spk_max_nan(m1, fabs(ltl + nig * bxl));
                    m2 = spk_max_nan(m2, fabs(l11 + nis * b11));
                    m2 = spk_max_nan(m2, fabs(bxl + nis * b14));
                }
            }
            if (A) sum += az[dm.zx + i * nx + r] * (bx - gamma * lt);
            m1 = spk_max_nan(m1, fabs(lt + nig * bx));
        }
        if (r < nu) {
            const real bu = bz[dm.zu + i * nu + r], b7 = bv[dm.o7 + i * nxu + nx + r];
            real lt = b7;
            if (A) sum += av[dm.o7 + i * nxu + nx + r] * (b7 - sigma * bu);
            m2 = spk_max_nan(m2, fabs(bu + nis * b7));
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const int jn = t.c0 + c - 1;
                const real b4 = bv[dm.o4 + jn * nu + r], l4 = SR[r + r * nu] * bu;
                lt += SR[r + r * nu] * b4;
                if (A) sum += av[dm.o4 + jn * nu + r] * (b4 - sigma * l4);
                m2 = spk_max_nan(m2, fabs(l4 + nis * b4));
            }
            if (A) sum += az[dm.zu + i * nu + r] * (bu - gamma * lt);
            m1 = spk_max_nan(m1, fabs(lt + nig * bu));
        }
        const real b2v = bv[dm.o2 + i];
        if (r < ny) {
            const real by = bz[dm.zy + i * ny + r], b1 = bv[i * ny + r];
            const real lty = b1 - bb[r] * b2v;
            if (A) sum += az[dm.zy + i * ny + r] * (by - gamma * lty) + av[i * ny + r] * (b1 - sigma * by);
            m1 = spk_max_nan(m1, fabs(lty + nig * by));
            m2 = spk_max_nan(m2, fabs(by + nis * b1));
        }
        if (r == 0) {   // scalars of the node: s_i <-> v2, tau_j <-> v5, v6, s_leaf <-> v12, v13
            const real bs = bz[dm.zs + i];
            real bdot = 0.;
            for (int k = 0; k < ny; ++k) bdot += bb[k] * bz[dm.zy + i * ny + k];
            const real l2 = bs - bdot;
            if (A) sum += az[dm.zs + i] * (bs - gamma * b2v) + av[dm.o2 + i] * (b2v - sigma * l2);
            m1 = spk_max_nan(m1, fabs(b2v + nig * bs));
            m2 = spk_max_nan(m2, fabs(l2 + nis * b2v));
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                const int j = t.c0 + c, jn = j - 1;
                const real bt = bz[dm.zt + jn], b5 = bv[dm.o5 + jn], b6 = bv[dm.o6 + jn];
                const real lt = 0.5 * b5 + 0.5 * b6, l56 = 0.5 * bt;
                if (A) sum += az[dm.zt + jn] * (bt - gamma * lt) + av[dm.o5 + jn] * (b5 - sigma * l56) + av[dm.o6 + jn] * (b6 - sigma * l56);
                m1 = spk_max_nan(m1, fabs(lt + nig * bt));
                m2 = spk_max_nan(m2, fabs(l56 + nis * b5)); m2 = spk_max_nan(m2, fabs(l56 + nis * b6));
                if (t.kids_leaf) {
                    const int li = j - dm.nn;
                    const real bsl = bz[dm.zs + j], b12 = bv[dm.o12 + li], b13 = bv[dm.o13 + li];
                    const real lts = 0.5 * b12 + 0.5 * b13, l1213 = 0.5 * bsl;
                    if (A) sum += az[dm.zs + j] * (bsl - gamma * lts) + av[dm.o12 + li] * (b12 - sigma * l1213) + av[dm.o13 + li] * (b13 - sigma * l1213);
                    m1 = spk_max_nan(m1, fabs(lts + nig * bsl));
                    m2 = spk_max_nan(m2, fabs(l1213 + nis * b12)); m2 = spk_max_nan(m2, fabs(l1213 + nis * b13));
                }
            }
        }
        acc[which] += sum;
        if (which == 0) { acc[2] = m1; acc[3] = m2; }
    }
}

}  // namespace spock
