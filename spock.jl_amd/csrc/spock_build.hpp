// spock_build.hpp -- the dynamics-dependent part of the build step ON THE DEVICE: ricatti_offline
// (/root/reference/src/model_dynamics/implicit_l.jl:46-97) and the composite matrices of the restructured S1 recursion
// (spock_host.cpp), recomputed in the handle's matrix pool when the dynamics A_w, B_w change (spock_update_dynamics) -- e.g. a model
// re-linearised between the steps of a closed loop -- without a trip through the host build and a new handle.  Everything
// else in the pool (costs, risk measures, S2 projector) does not depend on the dynamics and stays.
//
// One workgroup; the recursion over the stages is sequential, each stage a handful of small dense products (nx, nu <= 64) formed
// cooperatively by the workgroup's threads.  Arithmetic is double whatever `real` is (as in the host build step); the order
// of every sum is the host's (spock_host.cpp: mul / add / spd_inverse), so the two builds agree to rounding.
// X (execution policy): par(n, f) runs f(idx) for idx in [0, n) over the threads, then a barrier; one(f) runs f on one thread,
// then a barrier.  Host emulation (tests/emu): plain loops.
#pragma once
#include "spock_device.hpp"

#ifndef SPK_ILAMBDA
#ifdef SPOCK_HOST_EMU
#define SPK_ILAMBDA
#else
#define SPK_ILAMBDA __attribute__((always_inline))
#endif
#endif

namespace spock {

SPK_DEV long build_ws_doubles_dev(int nx, int nu, int d) {
    return 7L * nx * nx + 6L * nu * nx + 4L * nu * nu + (long)d * nx * nx;
}

// C (m x n) = op(A) op(B): column-major, ta / tb = transposed operand, k = inner dimension; s accumulates over p ascending (host mul)
template <class X>
SPK_DEV void bmm(X &x, double *C, int m, int n, int k, const double *A, bool ta, const double *B, bool tb) {
    const int lda = ta ? k : m, ldb = tb ? n : k;
    x.par(m * n, [&](int idx) SPK_ILAMBDA {
        const int i = idx % m, j = idx / m;
        double s = 0.;
        for (int p = 0; p < k; ++p) s += (ta ? A[p + (long)i * lda] : A[i + (long)p * lda]) * (tb ? B[j + (long)p * ldb] : B[p + (long)j * ldb]);
        C[idx] = s;
    });
}

template <class X>
SPK_DEV void build_riccati_body(X &x, const BuildArgs &a) {
    const Dims &dm = a.dm;
    const int N = dm.N, d = dm.d, nx = dm.nx, nu = dm.nu;
    const int xx = nx * nx, ux = nu * nx, uu = nu * nu;
    double *w = a.ws;
    double *Pn = w; w += xx;          // P_{t+1}
    double *Pc = w; w += xx;          // P_t
    double *T1 = w; w += xx;          // products (nx x nx)
    double *T2 = w; w += xx;
    double *sum_p = w; w += xx;
    double *T3 = w; w += xx;
    double *T4 = w; w += xx;
    double *BtP = w; w += ux;         // nu x nx
    double *sum_k = w; w += ux;
    double *K = w; w += ux;
    double *G = w; w += ux;           // nx x nu
    double *GR = w; w += ux;
    double *T5 = w; w += ux;
    double *sum_r = w; w += uu;
    double *rt = w; w += uu;
    double *Lc = w; w += uu;
    double *Rinv = w; w += uu;
    double *abk = w; w += (long)d * xx;
    real *M = a.mats;

    x.one([&]() SPK_ILAMBDA { *a.status = 0; });
    // A_w, B_w into the pool (the closed-loop step and the forward sweep read them from there); P_{N-1} = I
    x.par(d * xx, [&](int i) SPK_ILAMBDA { M[dm.mA + i] = (real)a.A[i]; });
    x.par(d * ux, [&](int i) SPK_ILAMBDA { M[dm.mB + i] = (real)a.B[i]; });
    x.par(xx, [&](int i) SPK_ILAMBDA { Pn[i] = (i % nx == i / nx) ? 1. : 0.; M[dm.mP + (long)(N - 1) * xx + i] = (real)Pn[i]; });

    for (int t = N - 2; t >= 0; --t) {
        x.par(uu, [&](int i) SPK_ILAMBDA { sum_r[i] = 0.; });
        x.par(ux, [&](int i) SPK_ILAMBDA { sum_k[i] = 0.; });
        for (int c = 0; c < d; ++c) {
            const int wv = (2 + c) % d;
            const double *Aw = a.A + (long)wv * xx, *Bw = a.B + (long)wv * ux;
            bmm(x, BtP, nu, nx, nx, Bw, true, Pn, false);                  // B_w' P_{t+1}
            bmm(x, rt, nu, nu, nx, BtP, false, Bw, false);
            x.par(uu, [&](int i) SPK_ILAMBDA { sum_r[i] += rt[i]; });
            bmm(x, T5, nu, nx, nx, BtP, false, Aw, false);
            x.par(ux, [&](int i) SPK_ILAMBDA { sum_k[i] += T5[i]; });
        }
        // Rtilde = I + sym(sum_r); its inverse through the Cholesky factor (spd_inverse)
        x.par(uu, [&](int idx) SPK_ILAMBDA { const int i = idx % nu, j = idx / nu; rt[idx] = (i == j ? 1. : 0.) + 0.5 * (sum_r[i + j * nu] + sum_r[j + i * nu]); });
        x.one([&]() SPK_ILAMBDA {
            for (int j = 0; j < nu; ++j) {
                double s = rt[j + j * nu];
                for (int k = 0; k < j; ++k) s -= Lc[j + k * nu] * Lc[j + k * nu];
                if (!(s > 0.)) { *a.status = 1; s = 1.; }
                const double dj = sqrt(s);
                Lc[j + j * nu] = dj;
                for (int i = j + 1; i < nu; ++i) {
                    double tt = rt[i + j * nu];
                    for (int k = 0; k < j; ++k) tt -= Lc[i + k * nu] * Lc[j + k * nu];
                    Lc[i + j * nu] = tt / dj;
                }
            }
        });
        x.par(nu, [&](int c) SPK_ILAMBDA {       // column c of the inverse: L y = e_c, L' x = y (in place in Rinv's column)
            double *xc = Rinv + (long)c * nu;
            for (int i = 0; i < nu; ++i) {
                double s = (i == c) ? 1. : 0.;
                for (int k = 0; k < i; ++k) s -= Lc[i + k * nu] * xc[k];
                xc[i] = s / Lc[i + i * nu];
            }
            for (int i = nu - 1; i >= 0; --i) {
                double s = xc[i];
                for (int k = i + 1; k < nu; ++k) s -= Lc[k + i * nu] * xc[k];
                xc[i] = s / Lc[i + i * nu];
            }
        });
        x.par(uu, [&](int idx) SPK_ILAMBDA { const int i = idx % nu, j = idx / nu; rt[idx] = (i == j) ? Rinv[idx] : 0.5 * ((i < j ? Rinv[i + j * nu] : Rinv[j + i * nu]) + (i < j ? Rinv[j + i * nu] : Rinv[i + j * nu])); });
        x.par(uu, [&](int i) SPK_ILAMBDA { Rinv[i] = rt[i]; M[dm.mRinv + (long)t * uu + i] = (real)rt[i]; });
        // K = Rinv (-sum_k)
        x.par(ux, [&](int i) SPK_ILAMBDA { T5[i] = -sum_k[i]; });
        bmm(x, K, nu, nx, nu, Rinv, false, T5, false);
        x.par(ux, [&](int i) SPK_ILAMBDA { M[dm.mK + (long)t * ux + i] = (real)K[i]; M[dm.mSweep + (long)t * dm.sw_stride + dm.sw_oK + i] = (real)K[i]; });
        // ABK_c = A_w + B_w K;  P_t = I + K'K + sum_c ABK_c' P_{t+1} ABK_c
        x.par(xx, [&](int i) SPK_ILAMBDA { sum_p[i] = 0.; });
        for (int c = 0; c < d; ++c) {
            const int wv = (2 + c) % d;
            const double *Aw = a.A + (long)wv * xx, *Bw = a.B + (long)wv * ux;
            double *ab = abk + (long)c * xx;
            bmm(x, T1, nx, nx, nu, Bw, false, K, false);
            x.par(xx, [&](int i) SPK_ILAMBDA {
                ab[i] = Aw[i] + T1[i];
                M[dm.mABK + ((long)(t + 1) * d + c) * xx + i] = (real)ab[i];
                M[dm.mSweep + (long)t * dm.sw_stride + dm.sw_oABK + (long)c * xx + i] = (real)ab[i];
            });
            bmm(x, T2, nx, nx, nx, ab, true, Pn, false);                   // ABK_c' P_{t+1}
            bmm(x, T3, nx, nx, nx, T2, false, ab, false);
            x.par(xx, [&](int i) SPK_ILAMBDA { sum_p[i] += T3[i]; });
        }
        bmm(x, T4, nx, nx, nu, K, true, K, false);                          // K'K
        x.par(xx, [&](int i) SPK_ILAMBDA {
            Pc[i] = (((i % nx == i / nx) ? 1. : 0.) + T4[i]) + sum_p[i];
            M[dm.mP + (long)t * xx + i] = (real)Pc[i];
        });
        // G = K' + sum_c (ABK_c' P_{t+1}) B_w;  U = G Rinv - K';  W_c = ABK_c' - G Rinv B_w';  V_c = Rinv B_w'
        x.par(ux, [&](int idx) SPK_ILAMBDA { const int i = idx % nx, j = idx / nx; G[idx] = K[j + i * nu]; });
        for (int c = 0; c < d; ++c) {
            const int wv = (2 + c) % d;
            const double *Bw = a.B + (long)wv * ux;
            bmm(x, T2, nx, nx, nx, abk + (long)c * xx, true, Pn, false);
            bmm(x, T5, nx, nu, nx, T2, false, Bw, false);
            x.par(ux, [&](int i) SPK_ILAMBDA { G[i] += T5[i]; });
        }
        bmm(x, GR, nx, nu, nu, G, false, Rinv, false);
        x.par(ux, [&](int idx) SPK_ILAMBDA { const int i = idx % nx, j = idx / nx; M[dm.mU + (long)t * ux + idx] = (real)(GR[idx] + -K[j + i * nu]); });
        for (int c = 0; c < d; ++c) {
            const int wv = (2 + c) % d;
            const double *Bw = a.B + (long)wv * ux;
            bmm(x, T1, nx, nx, nu, GR, false, Bw, true);                    // G Rinv B_w'
            x.par(xx, [&](int idx) SPK_ILAMBDA {
                const int i = idx % nx, j = idx / nx;
                const double wv2 = abk[(long)c * xx + j + (long)i * nx] + -T1[idx];
                M[dm.mW + ((long)t * d + c) * xx + idx] = (real)wv2;
                M[dm.mSweep + (long)t * dm.sw_stride + dm.sw_oW + (long)c * xx + idx] = (real)wv2;
            });
            bmm(x, T5, nu, nx, nu, Rinv, false, Bw, true);                  // Rinv B_w'
            x.par(ux, [&](int i) SPK_ILAMBDA {
                M[dm.mV + ((long)t * d + c) * ux + i] = (real)T5[i];
                M[dm.mSweep + (long)t * dm.sw_stride + dm.sw_oV + (long)c * ux + i] = (real)T5[i];
            });
        }
        x.par(xx, [&](int i) SPK_ILAMBDA { Pn[i] = Pc[i]; });
    }
}

}  // namespace spock
