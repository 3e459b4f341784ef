// spock_sp.hpp -- device pieces of the SuperMann-accelerated CP driver ("SP" = Spock):
//   run_sp!            /root/reference/src/model_algorithms/sp.jl:358-469
//   anderson!          /root/reference/src/model_algorithms/qnewton_directions/anderson.jl:1-123
//   spock_dot / norm   /root/reference/src/model_dynamics/implicit_l.jl:485-557
//
// One SP iteration on the device (driver: spock_abi.cpp; every kernel exits at once for a finished instance):
//   T(Z) -> ZB     the three CP kernels; phase B also accumulates <r, P r> (b_local / tb_p2 with SPM = 1)      (sp.jl:34-51)
//   FLAT1   r = Z - ZB, dr = r - r_old, new Anderson columns, Gram sums      (anderson.jl:13-65)
//   CTL_A   rnorm, Gram-Schmidt coefficients r2, r3, shift of R; keeps T(Z)'s CP residuals   (anderson.jl:46-50,76-88)
//   FLAT2   Q1 = dr - r2 Q2 - r3 Q3, its norm, Q'r                           (anderson.jl:88-110)
//   CTL_B   R11, gamma = R^-1 Q'r                                            (anderson.jl:94-113)
//   FLAT3   d = -r - MP gamma, candidate W = Z + d (Q1 /= rho: where it is read)  (anderson.jl:116-122, sp.jl:118-131)
//   T(W) -> WB     phase B also accumulates <rt, P rt>, <rt, P d> (SPM = 2) and forms the termination residuals of d
//   CTL_CD  K1 / K2 / fall-back decision, r_safe, backtrack count (sp.jl:411-446); termination test, history,
//           counters (sp.jl:295-344, :464) with the residuals of dp taken by linearity from the two evaluations
//   FLAT5   Z <- W | Z - coef (W - WB) | ZB ;  dp = Z - Z_old                (sp.jl:166-222, :270-281)
// = 13 launches.  The first version (SPOCK_AMD_SP_LOP=1, still built and tested) used separate passes instead:
//   ... T(Z), FLAT1, LOP <r, P r>, CTL_A, FLAT2, CTL_B, FLAT3, T(W), FLAT4 rt = W - WB, LOP <rt, P rt>, <rt, P d>,
//   CTL_C, FLAT5, LOP xi_1, xi_2 of dp (sp.jl:283-293), CTL_D = 18 launches.
//
// Differences from the reference that leave the iterates unchanged: the Anderson buffers are circular
// (no column shifting), MR is not stored (it only ever feeds Q), a rejected candidate is
// evaluated once instead of nine identical times (tau never changes: the line search result is discarded,
// sp.jl:439; the logged backtrack count is still 9).  Differences at rounding level: parallel summation order
// in the inner products, and the Gram-Schmidt coefficient against the third column is formed from Gram
// sums, (<Q3,Q1> - r2 <Q3,Q2>) / |Q3|, instead of a pass over the updated column.
#pragma once
#include "spock_device.hpp"

namespace spock {

SPK_DEV int sp_col(const SpCtl &c, int logical /*0 = newest*/) { return (c.head + logical) % 3; }

// ---------------------------------------------------------------------------------------------------
// flat element-wise steps; `e` is the element index inside the instance's nr-vector, sums[] accumulates
// rho2, rho3: norms of the Q columns at logical positions 1 and 2 (SpCtl::crho, read by the caller from the control block itself: the
// columns are stored unnormalised and divided where they are read -- bitwise the value FLAT3 used to store -- so that FLAT3 neither
// reads nor rewrites the newest one)
SPK_DEV void sp_flat_elem(const SpArgs &a, const SpCtl &c, int inst, long e, double *sums, double rho2, double rho3) {
    const long o = (long)inst * a.nr + e;
    const long m3 = (long)inst * 3 * a.nr;
    real *q1 = a.Q + m3 + (long)sp_col(c, 0) * a.nr, *q2 = a.Q + m3 + (long)sp_col(c, 1) * a.nr,
           *q3 = a.Q + m3 + (long)sp_col(c, 2) * a.nr;
    real *p1 = a.MP + m3 + (long)sp_col(c, 0) * a.nr, *p2 = a.MP + m3 + (long)sp_col(c, 1) * a.nr,
           *p3 = a.MP + m3 + (long)sp_col(c, 2) * a.nr;
    switch (a.op) {
        case SP_FLAT1: {   // head already points at the slot of the new column (set by CTL_D / solve start)
            const real z = a.Z[o], r = z - a.ZB[o];
            // r_old of update_sy! (sp.jl:232-255) and r live in ONE buffer: from here to the next FLAT1 both are this iteration's r
            // (a.ROLD aliases a.R) -- the old value is read, the new one written in place: one vector less written per iteration
            const real dr = r - a.R[o];
            a.R[o] = r;
            p1[e] = a.DP[o] - dr;                 // MP[:,1] = delta_p - delta_r   (anderson.jl:25-32)
            q1[e] = dr;                           // Q[:,1]  = MR[:,1] = delta_r   (anderson.jl:63-65)
            // (z_old <- z of update_sy! needs no copy either: see FLAT5)
            const real b = (real)(q2[e] / rho2), cc = (real)(q3[e] / rho3);
            sums[0] += dr * dr; sums[1] += b * dr; sums[2] += cc * dr; sums[3] += b * b; sums[4] += cc * cc; sums[5] += b * cc;
            sums[6] += dr * r; sums[7] += b * r; sums[8] += cc * r;      // MR'r: right-hand side of the corrected variant's normal equations
        } break;
        case SP_FLATW:                                                   // corrected variant: candidate with the shortened step
            a.W[o] = a.Z[o] + (real)a.tau * a.D[o];
            break;
        case SP_FLAT2: {
            const real b = (real)(q2[e] / rho2), cc = (real)(q3[e] / rho3);
            const real v = q1[e] - c.r2 * b - c.r3 * cc;   // anderson.jl:88
            q1[e] = v;                                         // stays unnormalised: its norm goes to crho[head] (CTL_B, anderson.jl:94-98)
            const real r = a.R[o];
            sums[0] += v * v; sums[1] += v * r; sums[2] += b * r; sums[3] += cc * r;
        } break;
        case SP_FLAT3: {
            const real w = (p1[e] * c.g[0] + p2[e] * c.g[1]) + p3[e] * c.g[2];   // MP gamma (anderson.jl:116)
            const real d = -a.R[o] - w;                            // anderson.jl:117-122
            a.D[o] = d;
            a.W[o] = a.Z[o] + 1. * d;                                // tau = 1 (sp.jl:118-131)
        } break;
        case SP_FLAT4:
            a.RW[o] = a.W[o] - a.WB[o];                              // sp.jl:155-160
            break;
        case SP_FLAT5: {
            // z_old of update_sy! (sp.jl:232-255) is the iterate this iteration started from -- still in Z here, nothing has written Z since
            // FLAT1 -- so delta_z = z_new - z_old needs no stored copy (one vector written and one read less per iteration)
            const real z_old = a.Z[o];
            real z = z_old;
            if (c.accept == 1) z = a.W[o];                           // K1 (sp.jl:166-191)
            else if (c.accept == 2) z -= c.coef * (a.fused_lop ? a.W[o] - a.WB[o] : a.RW[o]);   // K2 (sp.jl:193-222); rt = w - wbar (:155-160)
            else z = a.ZB[o];                                        // plain CP step (sp.jl:443-446)
            a.Z[o] = z;
            a.DP[o] = z - z_old;                                     // delta_z / delta_v (sp.jl:279-284)
        } break;
        default: break;
    }
}

// ---------------------------------------------------------------------------------------------------
// node-local application of L / L' to a vector pair b (and optionally b2) on the entries owned by node i:
//   acc[0] += <a, P b>   acc[1] += <a, P b2>      P = [[I, -gamma L'], [-sigma L, I]]   (implicit_l.jl:485-516)
//   acc[2] = max |L' b_v - b_z / gamma|           acc[3] = max |L b_z - b_v / sigma|    (sp.jl:283-293)
template <class P>
SPK_DEV void lop_local(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ A, const real *__restrict__ Bv,
                       const real *__restrict__ B2, const TState<P> &s, real gamma, real sigma, double *acc) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny();
    const int i = s.node, nxu = nx + nu, nz = dm.nz;
    const real nig = -1. / gamma, nis = -1. / sigma;
    const int nb = B2 ? 2 : 1;
    for (int which = 0; which < nb; ++which) {
        const real *__restrict__ b = which ? B2 : Bv;
        const real *bz = b, *bv = b + nz;
        const real *az = A, *av = A ? A + nz : nullptr;
        real sum = 0., m1 = acc[2], m2 = acc[3];
        const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? i * ny : 0);
        // ---- z entries x_i, u_i and v entries v7, v3_c, v4_c
        real bx[P::MAXNX], bu[P::MAXNU], ltx[P::MAXNX], ltu[P::MAXNU];
        SPK_UNROLL for (int k = 0; k < nx; ++k) { bx[k] = bz[dm.zx + i * nx + k]; ltx[k] = bv[dm.o7 + i * nxu + k]; }
        SPK_UNROLL for (int k = 0; k < nu; ++k) { bu[k] = bz[dm.zu + i * nu + k]; ltu[k] = bv[dm.o7 + i * nxu + nx + k]; }
        // v7 = (x_i, u_i)
        SPK_UNROLL for (int k = 0; k < nx; ++k) {
            const real lb = bx[k], vb = ltx[k];
            if (A) sum += av[dm.o7 + i * nxu + k] * (vb - sigma * lb);
            m2 = spk_max_nan(m2, fabs(lb + nis * vb));
        }
        SPK_UNROLL for (int k = 0; k < nu; ++k) {
            const real lb = bu[k], vb = ltu[k];
            if (A) sum += av[dm.o7 + i * nxu + nx + k] * (vb - sigma * lb);
            m2 = spk_max_nan(m2, fabs(lb + nis * vb));
        }
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int j = s.c0 + c, jn = j - 1;
            const MatPtr SQ = M + dm.mSqQ + (P::SHARED ? 0 : jn) * nx * nx;
            const MatPtr SR = M + dm.mSqR + (P::SHARED ? 0 : jn) * nu * nu;
            real l3[P::MAXNX], l4[P::MAXNU];
            sym_mv<P, false>(nx, SQ, bx, l3);                 // (L b_z) on v3_j
            sym_mv<P, false>(nu, SR, bu, l4);                 // (L b_z) on v4_j
            sym_mv<P, true>(nx, SQ, bv + dm.o3 + jn * nx, ltx);   // L' contribution to x_i
            sym_mv<P, true>(nu, SR, bv + dm.o4 + jn * nu, ltu);   // L' contribution to u_i
            SPK_UNROLL for (int k = 0; k < nx; ++k) {
                const real vb = bv[dm.o3 + jn * nx + k];
                if (A) sum += av[dm.o3 + jn * nx + k] * (vb - sigma * l3[k]);
                m2 = spk_max_nan(m2, fabs(l3[k] + nis * vb));
            }
            SPK_UNROLL for (int k = 0; k < nu; ++k) {
                const real vb = bv[dm.o4 + jn * nu + k];
                if (A) sum += av[dm.o4 + jn * nu + k] * (vb - sigma * l4[k]);
                m2 = spk_max_nan(m2, fabs(l4[k] + nis * vb));
            }
            // tau_j <-> v5_j, v6_j
            const real bt = bz[dm.zt + jn], b5 = bv[dm.o5 + jn], b6 = bv[dm.o6 + jn];
            const real lt = 0.5 * b5 + 0.5 * b6, l56 = 0.5 * bt;
            if (A) sum += az[dm.zt + jn] * (bt - gamma * lt) + av[dm.o5 + jn] * (b5 - sigma * l56) + av[dm.o6 + jn] * (b6 - sigma * l56);
            m1 = spk_max_nan(m1, fabs(lt + nig * bt));
            m2 = spk_max_nan(m2, fabs(l56 + nis * b5));
            m2 = spk_max_nan(m2, fabs(l56 + nis * b6));
            if (s.kids_leaf) {
                const int li = j - dm.nn;
                const MatPtr SN = M + dm.mSqQN + (P::SHARED ? 0 : li) * nx * nx;
                real bxl[P::MAXNX], l11[P::MAXNX], lt11[P::MAXNX];
                SPK_UNROLL for (int k = 0; k < nx; ++k) bxl[k] = bz[dm.zx + j * nx + k];
                sym_mv<P, false>(nx, SN, bxl, l11);
                sym_mv<P, false>(nx, SN, bv + dm.o11 + li * nx, lt11);
                SPK_UNROLL for (int k = 0; k < nx; ++k) {
                    const real v11 = bv[dm.o11 + li * nx + k], v14 = bv[dm.o14 + li * nx + k];
                    const real ltxl = v14 + lt11[k];
                    if (A) sum += az[dm.zx + j * nx + k] * (bxl[k] - gamma * ltxl) + av[dm.o11 + li * nx + k] * (v11 - sigma * l11[k])
                                  + av[dm.o14 + li * nx + k] * (v14 - sigma * bxl[k]);
                    m1 = spk_max_nan(m1, fabs(ltxl + nig * bxl[k]));
                    m2 = spk_max_nan(m2, fabs(l11[k] + nis * v11));
                    m2 = spk_max_nan(m2, fabs(bxl[k] + nis * v14));
                }
                const real bs = bz[dm.zs + j], b12 = bv[dm.o12 + li], b13 = bv[dm.o13 + li];
                const real lts = 0.5 * b12 + 0.5 * b13, l1213 = 0.5 * bs;
                if (A) sum += az[dm.zs + j] * (bs - gamma * lts) + av[dm.o12 + li] * (b12 - sigma * l1213) + av[dm.o13 + li] * (b13 - sigma * l1213);
                m1 = spk_max_nan(m1, fabs(lts + nig * bs));
                m2 = spk_max_nan(m2, fabs(l1213 + nis * b12));
                m2 = spk_max_nan(m2, fabs(l1213 + nis * b13));
            }
        }
        SPK_UNROLL for (int k = 0; k < nx; ++k) {
            if (A) sum += az[dm.zx + i * nx + k] * (bx[k] - gamma * ltx[k]);
            m1 = spk_max_nan(m1, fabs(ltx[k] + nig * bx[k]));
        }
        SPK_UNROLL for (int k = 0; k < nu; ++k) {
            if (A) sum += az[dm.zu + i * nu + k] * (bu[k] - gamma * ltu[k]);
            m1 = spk_max_nan(m1, fabs(ltu[k] + nig * bu[k]));
        }
        // ---- y_i, s_i <-> v1_i, v2_i
        const real bs = bz[dm.zs + i], b2v = bv[dm.o2 + i];
        real bdot = 0.;
        SPK_UNROLL for (int k = 0; k < ny; ++k) {
            const real by = bz[dm.zy + i * ny + k], b1 = bv[i * ny + k];
            bdot += bb[k] * by;
            const real lty = b1 - bb[k] * b2v;                     // (L' b_v) on y
            if (A) sum += az[dm.zy + i * ny + k] * (by - gamma * lty) + av[i * ny + k] * (b1 - sigma * by);
            m1 = spk_max_nan(m1, fabs(lty + nig * by));
            m2 = spk_max_nan(m2, fabs(by + nis * b1));
        }
        const real l2 = bs - bdot;                                  // (L b_z) on v2
        if (A) sum += az[dm.zs + i] * (bs - gamma * b2v) + av[dm.o2 + i] * (b2v - sigma * l2);
        m1 = spk_max_nan(m1, fabs(b2v + nig * bs));
        m2 = spk_max_nan(m2, fabs(l2 + nis * b2v));
        acc[which] += sum;
        if (which == 0) { acc[2] = m1; acc[3] = m2; }
    }
}

// ---------------------------------------------------------------------------------------------------
// control steps, ONE thread per instance; `sums` are the reduced partial sums of the preceding kernel(s)
SPK_DEV void sp_ctl_a(SpCtl &c, const double *gram, double pr, double xi1 = 0., double xi2 = 0.) {
    c.xi_fb[0] = xi1; c.xi_fb[1] = xi2;
    c.rnorm = sqrt(pr);                                              // sp.jl:395, implicit_l.jl:545-557
    c.k += 1;                                                        // generate_qnewton_direction! (models/spock.jl:212-224)
    // shift R diagonally (anderson.jl:46-50); reads below the diagonal of the UpperTriangular give 0
    double *R = c.R;
    for (int j = 1; j >= 0; --j)
        for (int i = 1; i >= 0; --i) R[(i + 1) + (j + 1) * 3] = (i <= j) ? R[i + j * 3] : 0.;
    c.r2 = 0.; c.r3 = 0.;
    if (c.k >= 2) { c.r2 = gram[1] / sqrt(gram[3]); R[0 + 1 * 3] = c.r2; }                       // anderson.jl:77-88, j = 2
    if (c.k >= 3) { c.r3 = (gram[2] - c.r2 * gram[5]) / sqrt(gram[4]); R[0 + 2 * 3] = c.r3; }    // j = 3
}
// Corrected variant (no reference counterpart; oracle: anderson_ls): gamma = argmin |r - MR gamma| over the min(k, 3) columns
// present, by the normal equations with the Gram sums of FLAT1 (the column buffers then hold the raw delta_r columns: FLAT2 /
// CTL_B are not launched).  A column whose Cholesky pivot falls below 1e-14 of its diagonal entry is dropped.
SPK_DEV void sp_ctl_a_ls(SpCtl &c, const double *f, double pr, double xi1, double xi2) {
    c.xi_fb[0] = xi1; c.xi_fb[1] = xi2;
    c.rnorm = sqrt(pr);
    c.k += 1;
    const int m = c.k < 3 ? c.k : 3;
    const double G[9] = {f[0], f[1], f[2], f[1], f[3], f[5], f[2], f[5], f[4]}, rhs[3] = {f[6], f[7], f[8]};
    double L[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, y[3] = {0., 0., 0.}, g[3] = {0., 0., 0.};
    bool valid[3] = {false, false, false};
    for (int j = 0; j < m; ++j) {
        double s = G[j + 3 * j];
        for (int k = 0; k < j; ++k) if (valid[k]) s -= L[j + 3 * k] * L[j + 3 * k];
        if (!(s > 1e-14 * G[j + 3 * j])) continue;
        valid[j] = true;
        L[j + 3 * j] = sqrt(s);
        for (int i = j + 1; i < m; ++i) {
            double t = G[i + 3 * j];
            for (int k = 0; k < j; ++k) if (valid[k]) t -= L[i + 3 * k] * L[j + 3 * k];
            L[i + 3 * j] = t / L[j + 3 * j];
        }
    }
    for (int j = 0; j < m; ++j) {
        if (!valid[j]) continue;
        double t = rhs[j];
        for (int k = 0; k < j; ++k) if (valid[k]) t -= L[j + 3 * k] * y[k];
        y[j] = t / L[j + 3 * j];
    }
    for (int j = 2; j >= 0; --j) {
        if (j >= m || !valid[j]) continue;
        double t = y[j];
        for (int k = j + 1; k < m; ++k) if (valid[k]) t -= L[k + 3 * j] * g[k];
        g[j] = t / L[j + 3 * j];
    }
    c.g[0] = g[0]; c.g[1] = g[1]; c.g[2] = g[2];
    c.rho = 1.; c.r2 = 0.; c.r3 = 0.;     // the normalisation of the newest column becomes the identity
    c.crho[sp_col(c, 0)] = 1.;
}
SPK_DEV void sp_ctl_b(SpCtl &c, const double *s4) {
    c.rho = sqrt(s4[0]);                                             // anderson.jl:70-75 / :93-99
    c.crho[sp_col(c, 0)] = c.rho;                                    // Q[:, 1] /= rho happens where the column is read
    c.R[0] = c.rho;
    double g[3] = {s4[1] / c.rho, s4[2], s4[3]};                     // gamma = Q' r with Q1 normalised (anderson.jl:110)
    const double *R = c.R;
    for (int j = 2; j >= 0; --j) {                                   // ldiv!(UpperTriangular(R), gamma) (anderson.jl:113)
        g[j] = g[j] / R[j + j * 3];
        for (int i = j - 1; i >= 0; --i) g[i] -= g[j] * R[i + j * 3];
    }
    c.g[0] = g[0]; c.g[1] = g[1]; c.g[2] = g[2];
}
SPK_DEV void sp_ctl_c(SpCtl &c, const Ctl &ctl, double s_rt, double s_rtd) {
    const double c1 = 0.99, q = 0.99, sigma_k2 = 0.1, lambda_sp = 1.;   // run_sp! keyword defaults (sp.jl:364-372)
    c.rtilde = sqrt(s_rt);
    c.accept = 0; c.backtrack = 9; c.coef = 0.;                      // rejected candidate: 9 identical re-evaluations, then CP step
    if (c.rnorm <= c.r_safe && c.rtilde <= c1 * c.rnorm) {           // K1 (sp.jl:166-174)
        c.accept = 1; c.backtrack = 0;
        c.r_safe = c.rtilde + pow(q, (double)ctl.iter);              // sp.jl:187
        return;
    }
    const double rho = s_rt - 1. * s_rtd;                            // <rt, P (rt - tau d)>, tau = 1 (sp.jl:420-434)
    c.rho_dot = rho;
    if (rho >= sigma_k2 * c.rnorm * c.rtilde) {                      // K2 (sp.jl:193-201)
        c.accept = 2; c.backtrack = 0;
        c.coef = lambda_sp * (rho / (c.rtilde * c.rtilde));          // sp.jl:211-218
    }
}
// `fin`: the value `done` takes when the solve ends here.  The merged control step CTL_CD runs BEFORE the state update
// FLAT5 of its iteration, so it ends a solve with done = 2 ("finishing": FLAT5 still applies the accepted step, every
// other kernel already exits); the next CTL_A turns 2 into 1.  The host treats any non-zero value as finished.
SPK_DEV void sp_ctl_d(SpCtl &c, Ctl &ctl, double x1, double x2, double *hist, int hist_cap, int fin = 1) {
    c.head = (c.head + 2) % 3;     // the next anderson! call writes its new column into the slot of the oldest one
    const bool log = (ctl.log_mode == 2);
    const double xi = log ? (x1 > x2 ? x1 : x2) : NAN;               // sp.jl:308
    ctl.xi1 = x1; ctl.xi2 = x2; ctl.xi = xi;
    if (hist && ctl.evals < hist_cap) {
        double *h = hist + (long)ctl.evals * kHistStride;
        h[0] = xi; h[1] = x1; h[2] = x2; h[3] = (double)c.backtrack;
    }
    ctl.evals += 1;
    bool term;
    if (log) term = (xi <= ctl.tol);
    else {
        const double t1 = ctl.tol * ctl.res0[0] > ctl.tol ? ctl.tol * ctl.res0[0] : ctl.tol;
        const double t2 = ctl.tol * ctl.res0[1] > ctl.tol ? ctl.tol * ctl.res0[1] : ctl.tol;
        term = (x1 <= t1 && x2 <= t2);
    }
    if (ctl.res0[0] == -INFINITY) ctl.res0[0] = x1;
    if (ctl.res0[1] == -INFINITY) ctl.res0[1] = x2;
    if (x1 != x1 || x2 != x2) { ctl.done = fin; ctl.status = 2; return; }   // the reference would spin to MAX_ITER on NaNs
    if (term) { ctl.done = fin; ctl.status = 0; return; }
    ctl.iter += 1;
    if (ctl.iter >= ctl.max_iter) { ctl.done = fin; ctl.status = 1; return; }
}

// CTL_C and CTL_D in one step (fused sequence).  The termination residuals xi_1 = |L' dp_v - dp_z / gamma|_inf,
// xi_2 = |L dp_z - dp_v / sigma|_inf of dp = z_new - z_old (sp.jl:283-293) need no pass of their own: by linearity
//   K1: dp = w - z = d          -> the residuals of d, formed next to phase B of T(w)      (ld1, ld2)
//   K2: dp = -coef rt           -> coef times the CP residuals of the evaluation T(w)      (xw1, xw2)
//   CP: dp = zbar - z           -> the CP residuals of the evaluation T(z)                 (c.xi_fb)
// (dp differs from these by the rounding of z_new - z_old, relative size eps |z| / |dp|.)
SPK_DEV void sp_ctl_cd(SpCtl &c, Ctl &ctl, double s_rt, double s_rtd, double xw1, double xw2, double ld1, double ld2,
                       double *hist, int hist_cap) {
    sp_ctl_c(c, ctl, s_rt, s_rtd);
    double x1, x2;
    if (c.accept == 1) { x1 = ld1; x2 = ld2; }
    else if (c.accept == 2) { const double f = fabs(c.coef); x1 = f * xw1; x2 = f * xw2; }
    else { x1 = c.xi_fb[0]; x2 = c.xi_fb[1]; }
    sp_ctl_d(c, ctl, x1, x2, hist, hist_cap, 2);
}

// Corrected variant, candidate `trial` of the line search with step tau = beta^trial (run_sp!'s loop sp.jl:411-441 with the
// result of perform_linesearch! kept).  An accepted candidate (or the last rejected one: plain CP step) closes the iteration --
// termination test, history, counters as in CTL_CD -- and parks the instance (done = 3) so that the kernels of the remaining
// candidates exit at once; SP_CTL_E at the end of the iteration re-opens it.
SPK_DEV void sp_ctl_trial(SpCtl &c, Ctl &ctl, double s_rt, double s_rtd, double xw1, double xw2, double ld1, double ld2,
                          double tau, int trial, int last, double *hist, int hist_cap) {
    const double c1 = 0.99, q = 0.99, sigma_k2 = 0.1, lambda_sp = 1.;
    c.rtilde = sqrt(s_rt);
    c.accept = 0; c.coef = 0.; c.backtrack = trial;
    if (c.rnorm <= c.r_safe && c.rtilde <= c1 * c.rnorm) {           // K1
        c.accept = 1;
        c.r_safe = c.rtilde + pow(q, (double)ctl.iter);
    } else {
        const double rho = s_rt - tau * s_rtd;                       // <rt, P (rt - tau d)>
        c.rho_dot = rho;
        if (rho >= sigma_k2 * c.rnorm * c.rtilde) {                  // K2
            c.accept = 2;
            c.coef = lambda_sp * (rho / (c.rtilde * c.rtilde));
        }
    }
    if (c.accept == 0) {
        if (!last) return;                                           // rejected: the next candidate follows
        c.backtrack = trial + 1;                                     // all candidates rejected: plain CP step
    }
    double x1, x2;
    if (c.accept == 1) { x1 = tau * ld1; x2 = tau * ld2; }           // dp = tau d
    else if (c.accept == 2) { const double f = fabs(c.coef); x1 = f * xw1; x2 = f * xw2; }
    else { x1 = c.xi_fb[0]; x2 = c.xi_fb[1]; }
    sp_ctl_d(c, ctl, x1, x2, hist, hist_cap, 2);
    if (!ctl.done && !last) ctl.done = 3;
}

// Closed loop on the device: what run_sp!'s caller does between two warm-started solves (mpc_simulation.jl:47-54,164-177), for ONE
// instance whose solve has just ended.  Thread 0 of the control kernel, after the rows of x0 <- A_w x0 + B_w u have been formed:
// the record of step t, then either the control blocks of a fresh solve (what solve_model! / run_sp! reset: sp.jl:383-389,
// models/spock.jl:231-251 -- the Anderson memory, R and z_old / r_old persist, that IS the warm start) or done = 1 after the last step.
SPK_DEV void sp_mpc_next(Ctl &ctl, SpCtl &c, int *t_ptr, int steps, double *rec) {
    const int t = *t_ptr;
    double *r = rec + (long)t * kMpcRec;
    r[0] = (double)ctl.status; r[1] = (double)ctl.evals; r[2] = (double)ctl.iter; r[3] = ctl.xi1; r[4] = ctl.xi2; r[5] = ctl.xi;
    *t_ptr = t + 1;
    if (t + 1 >= steps) { ctl.done = 1; return; }
    ctl.status = 0; ctl.evals = 0; ctl.iter = 0; ctl.pending = 0; ctl.result_parity = 0;
    ctl.res0[0] = -INFINITY; ctl.res0[1] = -INFINITY;
    ctl.xi1 = NAN; ctl.xi2 = NAN; ctl.xi = NAN;
    c.k = 0; c.r_safe = INFINITY; c.rnorm = INFINITY; c.rtilde = INFINITY; c.accept = 0; c.backtrack = 0;
    ctl.done = 0;
}

}  // namespace spock
