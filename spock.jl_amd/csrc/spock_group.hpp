// spock_group.hpp -- what ONE workgroup does: a "group" is a complete d-ary piece of the scenario tree
// (the whole tree, one subtree below the split stage, or the crown above it) processed with one thread
// per non-leaf node.  The orchestration is written once against an execution policy X:
//   X::each(f)      run f(tid, state) for every thread of the workgroup
//   X::barrier()    workgroup barrier (also orders the group's global-memory accesses)
//   X::lds()        the q / x exchange buffer of the S1 sweeps
//   X::flag()       one int shared by the workgroup
//   X::reduce3(..)  NaN-propagating max over the workgroup of three per-thread values -> thread 0
// On the GPU (spock_kernels.hip) each() runs its body once for threadIdx.x and the state lives in
// registers; the host emulation (tests/emu) loops over the threads.
//
// Kernel sequence of one CP iteration (driver: spock_abi.cpp, reference loop: cp.jl:205-228):
//   no split (t0 == 0):  FULL
//   split:               SUB_A (all subtrees)  ->  CROWN (one workgroup)  ->  SUB_B (all subtrees)
// SUB_A : phase A on the subtree + backward sweep up to the subtree root (q_root, d_i to global memory)
// CROWN : termination test of the PREVIOUS iteration, then phase A, backward and forward sweep and
//         phase B on the crown (stages < t0); stores x at the subtree roots
// SUB_B : forward sweep from the subtree root + phase B on the subtree; residual maxima per workgroup
#pragma once
#include "spock_device.hpp"

namespace spock {

template <class P, int MODE, class X>
SPK_DEV void group_body(X &x, const KArgs &a, int inst, int g) {
    const Dims &dm = a.dm;
    const P p(dm);
    const unsigned flags = a.flags;
    Ctl &ctl = a.ctl[inst];
    const int n_groups = dm.t0 > 0 ? dm.nsub + 1 : 1;
    double *partials = a.partials + (long)inst * n_groups * kPartialStride;

    // ---- control: has this instance finished?  (FULL / CROWN also run the pending termination test)
    if (!(flags & F_NO_CTL)) {
        if (MODE == MODE_FULL || MODE == MODE_CROWN || MODE == MODE_FINALIZE) {
            x.each([&](int tid, TState<P> &) {
                if (tid == 0) {
                    double *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
                    x.flag() = ctl_prologue(ctl, partials, n_groups, hist, a.hist_cap, a.parity, MODE != MODE_FINALIZE) ? 1 : 0;
                }
            });
            x.barrier();
            if (x.flag()) return;
        } else {
            if (ctl.done) return;
        }
    }
    if (MODE == MODE_FINALIZE) return;

    const double gamma = ctl.gamma, sigma = ctl.sigma;
    const double *M = a.mats;
    const double *z = a.z_in + (long)inst * dm.nz;
    const double *v = a.v_in + (long)inst * dm.nv;
    double *zo = a.z_out + (long)inst * dm.nz;
    double *vo = a.v_out + (long)inst * dm.nv;
    double *ex = x.lds();

    // ---- geometry of this group
    const bool crown = (MODE == MODE_CROWN);
    const bool sub = (MODE == MODE_SUB_A || MODE == MODE_SUB_B);
    const int root_stage = sub ? dm.t0 : 0;
    const int root_pos = sub ? g : 0;
    const int stages = sub ? dm.sub_stages : (crown ? dm.t0 + 1 : dm.N);
    const int root_node = dm.first[root_stage] + root_pos;
    x.each([&](int tid, TState<P> &s) { init_geometry(p, dm, tid, root_stage, root_pos, stages, crown, s); });

    // ---- phase A (node-local)
    if (MODE != MODE_SUB_B && !(flags & F_NO_ALOCAL)) {
        x.each([&](int, TState<P> &s) { if (s.active) a_local(p, dm, M, z, v, zo, s, gamma, flags); });
    }
    if (MODE == MODE_FULL || MODE == MODE_CROWN) x.barrier();   // zbar.s written by the parents is read in phase B

    // ---- S1 sweeps
    if (!(flags & F_NO_SWEEP)) {
        if (MODE != MODE_SUB_B) {
            const double *qg = crown ? a.qroot + (long)inst * dm.nsub * dm.nx : nullptr;
            double *q_out = (MODE == MODE_SUB_A) ? a.qroot + ((long)inst * dm.nsub + g) * dm.nx : nullptr;
            for (int l = stages - 2; l >= 0; --l) {
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) bwd_node(p, dm, M, root_stage + l, tid, ex, qg, q_out, s); });
                x.barrier();
            }
            if (MODE == MODE_SUB_A) {
                double *db = a.dbuf + (long)inst * dm.nn * dm.nu;
                x.each([&](int, TState<P> &s) {
                    if (s.active) for (int k = 0; k < p.nu(); ++k) db[s.node * p.nu() + k] = s.dd[k];
                });
            }
        }
        if (MODE != MODE_SUB_A) {
            if (MODE == MODE_SUB_B) {
                const double *db = a.dbuf + (long)inst * dm.nn * dm.nu;
                x.each([&](int, TState<P> &s) {
                    if (s.active) for (int k = 0; k < p.nu(); ++k) s.dd[k] = db[s.node * p.nu() + k];
                });
            }
            const double *xroot = (MODE == MODE_SUB_B) ? zo + dm.zx + (long)root_node * dm.nx : a.x0 + (long)inst * dm.nx;
            for (int l = 0; l <= stages - 2; ++l) {
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) fwd_node(p, dm, M, root_stage + l, tid, ex, xroot, zo, s); });
                x.barrier();
            }
        }
    }

    // ---- phase B (node-local) + residual maxima
    if (MODE != MODE_SUB_A && !(flags & F_NO_BLOCAL)) {
        ProxScal ps;
        ps.sigma = sigma;
        ps.inv_sigma_neg = -1. / sigma;
        ps.inv_gamma_neg = -1. / gamma;
        ps.want_xi = (ctl.log_mode == 2 && ctl.alg == 1);
        x.each([&](int tid, TState<P> &s) {
            double mx[3] = {0., 0., 0.};
            if (s.active) b_local(p, dm, M, z, v, zo, vo, s, ps, flags, mx);
            x.stash3(tid, mx);
        });
        const int slot = (MODE == MODE_SUB_B) ? g : (crown ? dm.nsub : 0);
        x.reduce3(partials + (long)slot * kPartialStride);
    }
}

}  // namespace spock
