// spock_group.hpp -- what ONE workgroup does: a "group" is a complete d-ary piece of the scenario tree
// (the whole tree, one subtree below the split stage, or the crown above it) processed with one thread
// per non-leaf node.  The orchestration is written once against an execution policy X:
//   X::each(f)      run f(tid, state) for every thread of the workgroup
//   X::barrier()    workgroup barrier (also orders the group's global-memory accesses)
//   X::lds()        the q / x exchange buffer of the S1 sweeps;  X::lds_mats() the staged sweep matrices
//   X::flag()       one int shared by the workgroup
//   X::stash3 / reduce3 / red3(k)   NaN-propagating max over the workgroup of three per-thread values;
//                   after reduce3() (which ends in a barrier) every thread can read the results
// On the GPU (spock_kernels.hip) each() runs its body once for threadIdx.x and the state lives in
// registers; the host emulation (tests/emu) loops over the threads.
//
// Kernel sequence of one CP iteration (driver: spock_abi.cpp, reference loop: cp.jl:205-228):
//   no split (t0 == 0):  FULL
//   split:               SUB_A (nsub + 1 workgroups)  ->  CROWN (one workgroup)  ->  SUB_B (nsub + 1 workgroups)
// SUB_A : workgroup g < nsub: phase A on subtree g + backward sweep up to its root (q_root, d_i to global
//         memory); workgroup nsub: phase A on the crown (stages < t0), c_i / e_i to global memory
// CROWN : termination test of the PREVIOUS iteration, then only the latency-critical part: backward and
//         forward sweep through the crown; stores x at the subtree roots
// SUB_B : workgroup g < nsub: forward sweep from the subtree root + phase B on the subtree;
//         workgroup nsub: phase B on the crown; residual maxima per workgroup
#pragma once
#include "spock_device.hpp"
#include "spock_team_local.hpp"
#include "spock_sp.hpp"
#ifdef SPOCK_HOST_EMU
#include <cstdlib>
#endif

namespace spock {

#ifndef SPK_LAMBDA_INLINE
#ifdef SPOCK_HOST_EMU
#define SPK_LAMBDA_INLINE
#else
#define SPK_LAMBDA_INLINE __attribute__((always_inline))
#endif
#endif

SPK_DEV int group_threads_dev(int d, int stages) {   // non-leaf nodes of a complete d-ary tree with `stages` stages
    int cnt = 0, lvl = 1;
    for (int t = 0; t < stages - 1; ++t) { cnt += lvl; lvl *= d; }
    return cnt;
}

// dst[k] = src[k], k < n, strided over the workgroup with four loads in flight per thread (a plain loop waits for every load before
// its store: 18 dependent round trips when a 1024-thread crown copies its 18 k sweep inputs)
template <class D, class S>
SPK_DEV void wg_copy(D dst, S src, int n, int tid, int nthr) {
    int k = tid;
    for (; k + 3 * nthr < n; k += 4 * nthr) {
        const real a0 = src[k], a1 = src[k + nthr], a2 = src[k + 2 * nthr], a3 = src[k + 3 * nthr];
        dst[k] = a0; dst[k + nthr] = a1; dst[k + 2 * nthr] = a2; dst[k + 3 * nthr] = a3;
    }
    for (; k < n; k += nthr) dst[k] = src[k];
}

template <class P, int MODE_, class X>
SPK_DEV void group_body(X &x, const KArgs &a, int inst, int g) {
    constexpr int MODE = (MODE_ == MODE_SUB_B_SP || MODE_ == MODE_SUB_B_LAT) ? MODE_SUB_B : (MODE_ == MODE_FULL_SP) ? MODE_FULL : MODE_;
    constexpr bool SP_SUMS = (MODE_ == MODE_SUB_B_SP || MODE_ == MODE_FULL_SP);   // phase B also accumulates the SP driver's inner products
    const Dims &dm = a.dm;
    const P p(dm);
    const unsigned flags = a.flags;
    Ctl &ctl = a.ctl[inst];
    // flat node-local kernels (spock_flat_local.hpp; lane-team shapes): this kernel then runs the S1 sweeps only -- c_i, e_i and the
    // leaves' q come from zbar's x / u entries, where MODE_FLAT_A parked them -- and the flat kernels' workgroups own the partials
    const bool flat = P::TEAMLOCAL && a.flat_local != 0;
    const int n_groups = a.flat_local ? a.flat_chunks : (dm.t0 > 0 ? dm.nsub + 1 : 1);   // (FINALIZE may run under a policy without lane teams)
    double *partials = a.partials + (long)inst * n_groups * kPartialStride;

    int stamp_idx = 0;
    auto stamp = [&]() {
        // stamps go to LDS (a global store in front of a barrier would make the barrier wait for its acknowledgement
        // and distort the segment being measured); dump_stamps() copies them out at the end
        if ((flags & F_DBG_STAMPS) && a.dbg && inst == 0 && g == 0 && stamp_idx < 64)
            x.each([&](int tid, TState<P> &) { if (tid == 0) x.clocks(x.stamp_lds() + 2 * stamp_idx); });
        ++stamp_idx;
    };
    auto dump_stamps = [&]() {
        if ((flags & F_DBG_STAMPS) && a.dbg && inst == 0 && g == 0) {
            unsigned long long *d = a.dbg + (long)(MODE == MODE_SUB_A ? 0 : (MODE == MODE_CROWN ? 1 : 2)) * 128;
            const int n = stamp_idx < 64 ? stamp_idx : 64;
            x.each([&](int tid, TState<P> &) { if (tid == 0) for (int k = 0; k < 2 * n; ++k) d[k] = x.stamp_lds()[k]; });
        }
    };
    stamp();   // 0: entry

    // ---- control: has this instance finished?  (FULL / CROWN also run the pending termination test)
    if (!(flags & F_NO_CTL)) {
        if ((MODE == MODE_FULL || MODE == MODE_CROWN || MODE == MODE_FINALIZE) && !(flags & F_SP)) {
            // one batch of loads: the control block's fields and this thread's share of the residual partials
            const CtlSnap snap = ctl_snapshot(ctl);
            x.each([&](int tid, TState<P> &) {
                double mx[3] = {0., 0., 0.};
                for (int gg = tid; gg < n_groups; gg += x.nthreads()) {
                    for (int k = 0; k < 3; ++k) mx[k] = spk_max_nan(mx[k], partials[gg * kPartialStride + k]);
                }
                x.stash3(tid, mx);
            });
            // Every thread reads the control block itself, and thread 0 is about to UPDATE it (pending, counters): all
            // snapshots must have been taken before that, or a late wave sees the new `pending` and takes the other branch
            // -- from then on the workgroup's barriers pair up out of phase.  (reduce3 starts with a barrier.)
            if (snap.pending && !snap.done) x.reduce3();   // workgroup-parallel max over the per-workgroup residual maxima
            else x.barrier();
            x.each([&](int tid, TState<P> &) {
                if (tid == 0) {
                    double *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
                    x.flag() = ctl_prologue(ctl, snap, x.red3(0), x.red3(1), x.red3(2), hist, a.hist_cap, a.parity, MODE != MODE_FINALIZE) ? 1 : 0;
                }
            });
            x.barrier();
            if (x.flag()) return;
        } else {
            if (ctl.done) return;
        }
    }
    if (MODE == MODE_FINALIZE) return;
    stamp();   // 1: control done


    const real *mats = a.mats + (long)inst * a.mats_stride;   // (per-instance pools: spock_update_dynamics_batch)
    const MatPtr M = (MatPtr)mats;
    const real *z = a.z_in + (long)inst * a.stride_z;
    const real *v = a.v_in + (long)inst * a.stride_v;
    real *zo = a.z_out + (long)inst * a.stride_z;
    real *vo = a.v_out + (long)inst * a.stride_v;
    real *ex = x.lds();

    // ---- geometry of this group
    const bool sub = (MODE == MODE_SUB_A || MODE == MODE_SUB_B || MODE == MODE_SUB_BA);
    const bool crown_blk = sub && (g == dm.nsub);          // the extra workgroup of SUB_A / SUB_B
    const bool crown = (MODE == MODE_CROWN) || crown_blk;
    const int root_stage = (sub && !crown_blk) ? dm.t0 : 0;
    const int root_pos = (sub && !crown_blk) ? g : 0;
    const int stages = crown ? dm.t0 + 1 : (sub ? dm.sub_stages : dm.N);
    const int root_node = dm.first[root_stage] + root_pos;
    const int nxu = p.nx() + p.nu();
    real *cb = a.cbuf ? a.cbuf + (long)inst * group_threads_dev(p.d(), dm.t0 + 1) * nxu : nullptr;
    x.each([&](int tid, TState<P> &s) { init_geometry(p, dm, tid, root_stage, root_pos, stages, crown, s); });

    // ---- stage the sweep matrices of this group's stages (and B) in LDS: one coalesced copy instead of a
    //      dependent matrix fetch in front of every tree level (measured 0.65 / 1.25 us per level without it)
    const bool sweeping = !(flags & F_NO_SWEEP) && !crown_blk;
    const bool lds_mats = a.lds_mats && sweeping && !(P::REGROWS && a.team_reg);
    real *lm = x.lds_mats();
    const int n_sw = (stages - 1) * dm.sw_stride, n_b = p.d() * p.nx() * p.nu();
    if (lds_mats) {
        const real *src = mats + dm.mSweep + (long)root_stage * dm.sw_stride;
        const real *srcB = mats + dm.mB;
        x.each([&](int tid, TState<P> &) {
            for (int k = tid; k < n_sw; k += x.nthreads()) lm[k] = src[k];
            for (int k = tid; k < n_b; k += x.nthreads()) lm[n_sw + k] = srcB[k];
        });
        x.barrier();
    }

    // ---- team sweeps: carve the LDS vectors (spock_device.hpp: TeamLds)
    const int team = sweeping ? a.team : 0;
#ifndef SPOCK_HOST_EMU
    // sweep levels as matrix products on the matrix cores (spock_device.hpp: bwd_level_mfma) when the blocks fill 16 x 16 tiles
    const bool use_mfma = P::MFMA_SWEEP && team > 0 && mfma_sweep_applies(p.nx(), p.nu(), p.d());
#endif
    const int nl_grp = group_threads_dev(p.d(), stages);          // non-leaf nodes of this group
    int ngl_grp = 1;                                               // leaves of this group
    for (int k = 0; k < stages - 1; ++k) ngl_grp *= p.d();
    TeamLds tl;
    tl.XQ = ex; tl.ED = tl.XQ + nl_grp * p.nx(); tl.UL = tl.ED + nl_grp * p.nu(); tl.XL = tl.UL + nl_grp * p.nu();
    real *lds_after_vectors = tl.XL + ngl_grp * p.nx();
    // flat node-local kernels + group-leaf vector in global memory: that vector IS a region of global memory already -- zbar's own
    // x entries of the group leaves (subtrees: q_leaf in, x_leaf out), the subtree roots' q in qroot (crown, on the way up)
    const bool xl_direct = flat && a.xl_global && !crown_blk;
    const long gl_first_x = dm.first[root_stage + stages - 1] + (long)root_pos * ngl_grp;
    if (a.xl_global) {   // the group-leaf vector in global memory (one slice per workgroup), everything else as before
        lds_after_vectors = tl.XL;
        tl.XL = a.xlbuf + ((long)inst * (dm.nsub + 1) + g) * a.xl_stride;
        if (xl_direct) tl.XL = (MODE == MODE_CROWN) ? a.qroot + (long)inst * dm.nsub * dm.nx : zo + dm.zx + gl_first_x * p.nx();
    }

    // register-resident variant of the team sweep: rows of this thread's node in registers, loaded before phase A so that
    // their latency hides behind it
    const bool team_reg = P::REGROWS && team > 0 && a.team_reg && (team * nl_grp <= x.nthreads());
    if (team_reg) x.each([&](int tid, TState<P> &s) { team_reg_init(p, dm, M, root_stage, stages, team, tid, s, MODE == MODE_SUB_BA ? 2 : (MODE == MODE_CROWN ? 1 : 3)); });

    // Every first touch of data another kernel wrote is a cold miss (~1.5-2 us: the writer sat on another XCD).  Issue the
    // matrix-row loads above and the sweep inputs staged here in registers back to back, before waiting for any of them.
    // (Hoisting them above the control section as well was tried: no measurable gain.  The wrong iterates seen with that
    // variant were not its fault: they came from the control-block snapshot race fixed above, which the changed timing exposed.)
    //   crown : c_i / e_i of the crown nodes (cbuf) and q of the subtree roots (qroot)
    //   SUB_B : d_i of the subtree's nodes (dbuf)
    const int st_n1 = (MODE == MODE_CROWN) ? nl_grp * nxu : (((MODE == MODE_SUB_B || MODE == MODE_SUB_BA) && !crown_blk) ? nl_grp * p.nu() : 0);
    const int st_n2 = (MODE == MODE_CROWN) ? ngl_grp * p.nx() : 0;
    const bool staged = team && (st_n1 + st_n2 > 0) && (st_n1 + st_n2 <= kStageRegs * x.nthreads());
    if (staged) {
        const real *qg = a.qroot + (long)inst * dm.nsub * dm.nx;
        const real *db = a.dbuf + (long)inst * dm.nn * dm.nu;
        x.each([&](int tid, TState<P> &s) {
            for (int j = 0; j < kStageRegs; ++j) {
                const int k = tid + j * x.nthreads();
                real val = 0.;
                if (MODE == MODE_CROWN) {
                    if (k < st_n1) val = flat ? (k < nl_grp * p.nx() ? zo[dm.zx + k] : zo[dm.zu + (k - nl_grp * p.nx())]) : cb[k];   // (crown nodes = global nodes 0 ..)
                    else if (k < st_n1 + st_n2) val = qg[k - st_n1];
                } else if (k < st_n1) {   // element k of ED: local node k / nu sits in the level chunk that contains it
                    const int li = k / p.nu(), comp = k - li * p.nu();
                    int lfirst = 0, cnt = 1, l = 0;
                    while (li >= lfirst + cnt) { lfirst += cnt; cnt *= p.d(); ++l; }
                    const long gnode = dm.first[root_stage + l] + (long)root_pos * cnt + (li - lfirst);
                    val = db[gnode * p.nu() + comp];
                }
                s.stg[j] = val;
            }
        });
    }

    const real gamma = ctl.gamma, sigma = ctl.sigma;
    const int nx = p.nx(), nu = p.nu(), d = p.d(), nthr = x.nthreads();
    auto level_geo = [&](int l, int &lfirst, int &cnt, long &gfirst) SPK_LAMBDA_INLINE {
        cnt = 1; lfirst = 0;
        for (int k = 0; k < l; ++k) { lfirst += cnt; cnt *= d; }
        gfirst = dm.first[root_stage + l] + (long)root_pos * cnt;
    };
    // lane-team node-local phases (large nx / nu, diagonal costs): T lanes per node, team scratch behind the sweep vectors
    const bool team_local = P::TEAMLOCAL;
    const int TT = a.team > 0 ? a.team : 8, nteams = nthr / TT;
    real *tscr = lds_after_vectors;
    const int tscr_stride = team_scratch_doubles(nu, p.ny() + 2 * d, d, TT);

    // ---- phase A (node-local) on the iterate (zi, vi); its part of zbar goes to zout, c_i / e_i (and the leaves' xbar) to
    //      the thread state or the lane teams' LDS vectors, for the crown's extra workgroup to cbuf
    auto phase_a = [&](const real *zi, const real *vi, real *zout, const real *v2x) SPK_LAMBDA_INLINE {
        if (flat) {   // c_i, e_i, q_leaf of this subtree: zbar's x / u entries -> the lane teams' LDS vectors (level chunks are contiguous)
            if (crown_blk || !sweeping) return;
            x.each([&](int tid, TState<P> &) {
                for (int l = 0; l <= stages - 2; ++l) {
                    int lfirst, cnt; long gfirst;
                    level_geo(l, lfirst, cnt, gfirst);
                    wg_copy(tl.XQ + lfirst * nx, zout + dm.zx + gfirst * nx, cnt * nx, tid, nthr);
                    wg_copy(tl.ED + lfirst * nu, zout + dm.zu + gfirst * nu, cnt * nu, tid, nthr);
                }
                if (!xl_direct) wg_copy(tl.XL, zout + dm.zx + gl_first_x * nx, ngl_grp * nx, tid, nthr);
            });
            return;   // (backward() starts with a barrier)
        }
        if (flags & F_NO_ALOCAL) return;
        if (team_local) {
            for (int base = 0; base < nl_grp; base += nteams) {
                x.each([&](int tid, TState<P> &s) {
                    const int tm = tid / TT, r = tid - tm * TT;
                    if (tm >= nteams) return;
                    const TeamNode tn = team_node(dm, d, base + tm, nl_grp, root_stage, root_pos, stages, crown);
                    ta_p1(p, dm, M, zi, vi, zout, tn, r, TT, tscr + tm * tscr_stride, crown_blk ? nullptr : tl.XL, gamma, flags, s.tlr, v2x);
                });
                x.barrier();
                if (!(flags & F_RAW_LT))
                    x.each([&](int tid, TState<P> &s) {
                        const int tm = tid / TT, r = tid - tm * TT;
                        if (tm >= nteams) return;
                        const TeamNode tn = team_node(dm, d, base + tm, nl_grp, root_stage, root_pos, stages, crown);
                        real *cq = crown_blk ? cb + tn.li * nx : tl.XQ + tn.li * nx;
                        real *ce = crown_blk ? cb + nl_grp * nx + tn.li * nu : tl.ED + tn.li * nu;
                        ta_p2(p, dm, M, zout, tn, r, TT, tscr + tm * tscr_stride, cq, ce, s.tlr);
                    });
                x.barrier();
            }
            return;
        }
        x.each([&](int, TState<P> &s) { if (s.active) a_local(p, dm, M, zi, vi, zout, s, gamma, flags, v2x); });
        if (team && !crown_blk) {   // hand c_i, e_i and the leaves' q = -xbar over to the lane teams
            x.each([&](int tid, TState<P> &s) {
                if (!s.active) return;
                for (int k = 0; k < nx; ++k) tl.XQ[tid * nx + k] = s.xb[k];
                for (int k = 0; k < nu; ++k) tl.ED[tid * nu + k] = s.ub[k];
                if (s.kids_leaf)
                    for (int c = 0; c < d; ++c)
                        for (int k = 0; k < nx; ++k) tl.XL[(s.pos * d + c) * nx + k] = -s.xl[c][k];
            });
        }
        if (crown_blk && !(flags & F_NO_SWEEP))
            x.each([&](int tid, TState<P> &s) {   // layout [c of every crown node | e of every crown node]
                if (!s.active) return;
                for (int k = 0; k < nx; ++k) cb[tid * nx + k] = s.xb[k];
                for (int k = 0; k < nu; ++k) cb[nl_grp * nx + tid * nu + k] = s.ub[k];
            });
    };

    // ---- S1 backward sweep over the group's levels; the subtree kernels leave q_root and every d_i in global memory
    const bool stores_q = (MODE == MODE_SUB_A || MODE == MODE_SUB_BA);
    auto backward = [&]() SPK_LAMBDA_INLINE {
        if (!sweeping || (flags & F_DBG_NO_BWD)) return;
        if (team) {   // lane-team version (all sweep vectors in LDS)
            if (MODE == MODE_CROWN) {   // c_i, e_i of the crown and q of the subtree roots -> LDS
                const real *qg = a.qroot + (long)inst * dm.nsub * dm.nx;
                x.each([&](int tid, TState<P> &s) {
                    if (staged) {   // XQ and ED are adjacent in LDS, exactly like c | e in cbuf
                        for (int j = 0; j < kStageRegs; ++j) {
                            const int k = tid + j * nthr;
                            if (k < st_n1) tl.XQ[k] = s.stg[j];
                            else if (k < st_n1 + st_n2) tl.XL[k - st_n1] = s.stg[j];
                        }
                    } else {
                        wg_copy(tl.XQ, flat ? (const real *)(zo + dm.zx) : (const real *)cb, nl_grp * nx, tid, nthr);
                        wg_copy(tl.ED, flat ? (const real *)(zo + dm.zu) : (const real *)(cb + nl_grp * nx), nl_grp * nu, tid, nthr);
                        if (!xl_direct) wg_copy(tl.XL, qg, ngl_grp * nx, tid, nthr);
                    }
                });
            }
            x.barrier();
            stamp();   // team vectors in LDS
            for (int l = stages - 2; l >= 0; --l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                if (team_reg) {
                    // crown: the forward sweep's rows are fetched behind the last backward level, so that the two row sets
                    // are not live together (with both the (5, 5) crown kernel spills registers)
                    if (MODE == MODE_CROWN && l == 0)
                        x.each([&](int tid, TState<P> &s) { team_reg_init(p, dm, M, root_stage, stages, team, tid, s, 2); });
                    x.each([&](int, TState<P> &s) { if (s.tn_li >= 0 && s.tn_lvl == l) bwd_team_reg(p, stages, tl, s); });
                } else if (lds_mats) {
                    const real *SW = lm + l * dm.sw_stride;
#ifndef SPOCK_HOST_EMU
                    // (nx = nu = 20: the stage's matrices are staged in LDS; the fragments come from there)
                    const bool mf20 = use_mfma && nx == 20;
                    if constexpr (P::MFMA_SWEEP) { if (mf20) {
                        if (d == 2) x.each([&](int tid, TState<P> &) { bwd_level_mfma<20, 20, 2>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                        else x.each([&](int tid, TState<P> &) { bwd_level_mfma<20, 20, 3>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                    } }
                    if (!mf20)
#endif
                    x.each([&](int tid, TState<P> &) { bwd_level_team(p, dm, SW, lfirst, cnt, l == stages - 2, team, tid, nthr, tl); });
                } else {
                    const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
#ifndef SPOCK_HOST_EMU
                    if constexpr (P::MFMA_SWEEP) {
                        if (use_mfma) {
                            if (nx == 64) x.each([&](int tid, TState<P> &) { bwd_level_mfma<64, 16, 2>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                            else if (d == 2) x.each([&](int tid, TState<P> &) { bwd_level_mfma<20, 20, 2>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                            else x.each([&](int tid, TState<P> &) { bwd_level_mfma<20, 20, 3>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                        }
                    }
                    if (!use_mfma)
#endif
                    x.each([&](int tid, TState<P> &) { bwd_level_team(p, dm, SW, lfirst, cnt, l == stages - 2, team, tid, nthr, tl); });
                }
                x.barrier();
                stamp();   // one per backward level
            }
            if (stores_q) {   // q at the subtree root and every d_i -> HBM (level chunks are contiguous)
                real *q_out = a.qroot + ((long)inst * dm.nsub + g) * dm.nx;
                real *db = a.dbuf + (long)inst * dm.nn * dm.nu;
                x.each([&](int tid, TState<P> &) {
                    if (tid < nx) q_out[tid] = tl.XQ[tid];
                    for (int l = 0; l <= stages - 2; ++l) {
                        int lfirst, cnt; long gfirst;
                        level_geo(l, lfirst, cnt, gfirst);
                        wg_copy(db + gfirst * nu, tl.ED + lfirst * nu, cnt * nu, tid, nthr);
                    }
                });
            }
            return;
        }
        // one-thread-per-node version (fallback when the team vectors do not fit in LDS, and the batched regime)
        if (MODE == MODE_CROWN) {
            x.each([&](int tid, TState<P> &s) {
                if (!s.active) return;
                for (int k = 0; k < nx; ++k) s.xb[k] = cb[tid * nx + k];
                for (int k = 0; k < nu; ++k) s.ub[k] = cb[nl_grp * nx + tid * nu + k];
            });
        }
        const real *qg = (MODE == MODE_CROWN) ? a.qroot + (long)inst * dm.nsub * dm.nx : nullptr;
        real *q_out = stores_q ? a.qroot + ((long)inst * dm.nsub + g) * dm.nx : nullptr;
        for (int l = stages - 2; l >= 0; --l) {
            if (lds_mats) {
                const real *SW = lm + l * dm.sw_stride;
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) bwd_node(p, dm, SW, tid, ex, qg, q_out, s); });
            } else {
                const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) bwd_node(p, dm, SW, tid, ex, qg, q_out, s); });
            }
            x.barrier();
            stamp();   // one per backward level
        }
        if (stores_q) {
            real *db = a.dbuf + (long)inst * dm.nn * dm.nu;
            x.each([&](int, TState<P> &s) {
                if (s.active) for (int k = 0; k < nu; ++k) db[s.node * nu + k] = s.dd[k];
            });
        }
    };

    // ---- S1 forward roll-out from the group root; x_i, u_i (and the group leaves' x) go to zbar and the thread state
    const bool loads_d = (MODE == MODE_SUB_B || MODE == MODE_SUB_BA);
    auto forward = [&]() SPK_LAMBDA_INLINE {
        if (!sweeping || (flags & F_DBG_NO_FWD)) return;
        const real *xroot = loads_d ? zo + dm.zx + (long)root_node * dm.nx : a.x0 + (long)inst * dm.nx;
        const real *db = a.dbuf + (long)inst * dm.nn * dm.nu;
        if (team) {
            const real *lmB = lm + n_sw;
            const MatPtr gB = M + dm.mB;
            if (xl_direct && MODE == MODE_CROWN) tl.XL = zo + dm.zx + gl_first_x * nx;   // on the way down: x of the subtree roots, in place
            x.each([&](int tid, TState<P> &s) {
                if (tid < nx) tl.XQ[tid] = xroot[tid];
                if (loads_d && staged) {
                    for (int j = 0; j < kStageRegs; ++j) { const int k = tid + j * nthr; if (k < st_n1) tl.ED[k] = s.stg[j]; }
                } else if (loads_d) {
                    for (int l = 0; l <= stages - 2; ++l) {
                        int lfirst, cnt; long gfirst;
                        level_geo(l, lfirst, cnt, gfirst);
                        wg_copy(tl.ED + lfirst * nu, db + gfirst * nu, cnt * nu, tid, nthr);
                    }
                }
            });
            x.barrier();
            stamp();   // forward inputs in LDS
            for (int l = 0; l <= stages - 2; ++l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                if (team_reg) {
                    x.each([&](int, TState<P> &s) { if (s.tn_li >= 0 && s.tn_lvl == l) fwd_team_reg(p, stages, tl, s); });
                } else if (lds_mats) {
                    const real *SW = lm + l * dm.sw_stride;
#ifndef SPOCK_HOST_EMU
                    const bool mf20 = use_mfma && nx == 20;
                    if constexpr (P::MFMA_SWEEP) { if (mf20) {
                        if (d == 2) x.each([&](int tid, TState<P> &) { fwd_level_mfma<20, 20, 2>(p, dm, SW, lmB, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                        else x.each([&](int tid, TState<P> &) { fwd_level_mfma<20, 20, 3>(p, dm, SW, lmB, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                    } }
                    if (!mf20)
#endif
                    x.each([&](int tid, TState<P> &) { fwd_level_team(p, dm, SW, lmB, lfirst, cnt, l == stages - 2, team, tid, nthr, tl); });
                } else {
                    const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
#ifndef SPOCK_HOST_EMU
                    if constexpr (P::MFMA_SWEEP) {
                        if (use_mfma) {
                            if (nx == 64) x.each([&](int tid, TState<P> &) { fwd_level_mfma<64, 16, 2>(p, dm, SW, gB, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                            else if (d == 2) x.each([&](int tid, TState<P> &) { fwd_level_mfma<20, 20, 2>(p, dm, SW, gB, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                            else x.each([&](int tid, TState<P> &) { fwd_level_mfma<20, 20, 3>(p, dm, SW, gB, lfirst, cnt, l == stages - 2, tid, nthr, tl); });
                        }
                    }
                    if (!use_mfma)
#endif
                    x.each([&](int tid, TState<P> &) { fwd_level_team(p, dm, SW, gB, lfirst, cnt, l == stages - 2, team, tid, nthr, tl); });
                }
                x.barrier();
                stamp();   // one per forward level
            }
            // x_i, u_i and the group leaves' x: coalesced level chunks LDS -> zbar; back into the thread state for phase B
            x.each([&](int tid, TState<P> &s) {
                for (int l = 0; l <= stages - 2; ++l) {
                    int lfirst, cnt; long gfirst;
                    level_geo(l, lfirst, cnt, gfirst);
                    wg_copy(zo + dm.zx + gfirst * nx, tl.XQ + lfirst * nx, cnt * nx, tid, nthr);
                    wg_copy(zo + dm.zu + gfirst * nu, tl.UL + lfirst * nu, cnt * nu, tid, nthr);
                }
                if (!xl_direct) wg_copy(zo + dm.zx + gl_first_x * nx, tl.XL, ngl_grp * nx, tid, nthr);
                if (s.active && !team_local) {
                    for (int k = 0; k < nx; ++k) s.xb[k] = tl.XQ[tid * nx + k];
                    for (int k = 0; k < nu; ++k) s.ub[k] = tl.UL[tid * nu + k];
                    if (s.kids_leaf)
                        for (int c = 0; c < d; ++c)
                            for (int k = 0; k < nx; ++k) s.xl[c][k] = tl.XL[(s.pos * d + c) * nx + k];
                }
            });
            return;
        }
        if (loads_d) {
            x.each([&](int, TState<P> &s) {
                if (s.active) for (int k = 0; k < nu; ++k) s.dd[k] = db[s.node * nu + k];
            });
        }
        for (int l = 0; l <= stages - 2; ++l) {
            if (lds_mats) {
                const real *SW = lm + l * dm.sw_stride, *Bm = lm + n_sw;
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) fwd_node(p, dm, SW, Bm, tid, ex, xroot, s); });
            } else {
                const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride, Bm = M + dm.mB;
                x.each([&](int tid, TState<P> &s) { if (s.active && s.lvl == l) fwd_node(p, dm, SW, Bm, tid, ex, xroot, s); });
            }
            x.barrier();
            stamp();   // one per forward level
        }
        x.each([&](int, TState<P> &s) { if (s.active) fwd_store(p, dm, zo, s); });
        stamp();
    };

    // ---- phase B (node-local) + residual maxima
    auto make_ps = [&]() {
        ProxScal ps;
        ps.sigma = sigma;
        ps.inv_sigma = 1. / sigma;
        ps.inv_sigma_neg = -1. / sigma;
        ps.inv_gamma_neg = -1. / gamma;
        ps.gamma = gamma;
        ps.want_xi = (ctl.log_mode == 2 && ctl.alg == 1);
        return ps;
    };
    auto phase_b = [&]() SPK_LAMBDA_INLINE {
        if ((flags & F_NO_BLOCAL) || flat) return;
        const ProxScal ps = make_ps();
        const unsigned bflags = crown_blk ? (flags | F_NO_SWEEP) : flags;   // the crown's x, u come from zbar
        if (team_local) {
            const bool from_zb = (bflags & F_NO_SWEEP) != 0;
            x.each([&](int, TState<P> &s) { s.mx3[0] = s.mx3[1] = s.mx3[2] = 0.; s.sp3[0] = s.sp3[1] = s.sp3[2] = 0.; });
            const real *dz_ = (SP_SUMS && a.dir) ? a.dir + (long)inst * a.stride_z : nullptr;
            for (int base = 0; base < nl_grp; base += nteams) {
                x.each([&](int tid, TState<P> &s) {
                    const int tm = tid / TT, r = tid - tm * TT;
                    if (tm >= nteams) return;
                    const TeamNode tn = team_node(dm, d, base + tm, nl_grp, root_stage, root_pos, stages, crown);
                    const real *xs = from_zb ? zo + dm.zx + (long)tn.node * nx : tl.XQ + tn.li * nx;
                    const real *us = from_zb ? zo + dm.zu + (long)tn.node * nu : tl.UL + tn.li * nu;
                    const real *xl = from_zb ? zo + dm.zx + (long)tn.c0 * nx : tl.XL + (tn.pos * d) * nx;
                    tb_p1(p, dm, M, z, v, zo, tn, r, TT, tscr + tm * tscr_stride, xs, us, xl, nx, ps, bflags, s.tlr);
                });
                x.barrier();
                x.each([&](int tid, TState<P> &s) {
                    const int tm = tid / TT, r = tid - tm * TT;
                    if (tm >= nteams) return;
                    const TeamNode tn = team_node(dm, d, base + tm, nl_grp, root_stage, root_pos, stages, crown);
                    if constexpr (SP_SUMS) {
                        if (a.sp_mode == 2) tb_p2<P, 2>(p, dm, M, z, v, zo, vo, tn, r, TT, tscr + tm * tscr_stride, ps, bflags, s.tlr, s.mx3, s.sp3, dz_, dz_ + dm.nz);
                        else tb_p2<P, 1>(p, dm, M, z, v, zo, vo, tn, r, TT, tscr + tm * tscr_stride, ps, bflags, s.tlr, s.mx3, s.sp3);
                    } else tb_p2(p, dm, M, z, v, zo, vo, tn, r, TT, tscr + tm * tscr_stride, ps, bflags, s.tlr, s.mx3);
                });
                x.barrier();
            }
            x.each([&](int tid, TState<P> &s) { x.stash3(tid, s.mx3); if (SP_SUMS) x.stash_sum3(tid, s.sp3); });
        } else if constexpr (!SP_SUMS) {
            // Latency regime (register-resident lane-team sweep: the workgroup has T threads per node, phase B keeps only
            // the first wave busy): the leaf-children blocks are done by thread 64 + li on the second wave while thread li
            // does the rest of node li (b_local PART).  The leaves' x comes from the sweep's LDS vector.
            // (A kernel of its own, MODE_SUB_B_LAT: the two extra copies of phase B change the register allocation of
            // whatever kernel contains them -- spills in the fused SUB_BA kernel, 8 % in the batched SUB_B.)
            constexpr bool CAN_SHADOW = (MODE_ == MODE_SUB_B_LAT) && P::REGROWS;
            const bool leaf_shadow = CAN_SHADOW && team_reg && !crown_blk && !(bflags & F_NO_SWEEP) && nl_grp <= 64 && nthr >= 128 &&
                                     (root_stage + stages == dm.N);
            x.each([&](int tid, TState<P> &s) {
                real mx[3] = {0., 0., 0.};
                if constexpr (CAN_SHADOW) {
                    if (s.active) {
                        if (leaf_shadow) b_local<P, 0, 1>(p, dm, M, z, v, zo, vo, s, ps, bflags, mx);
                        else b_local(p, dm, M, z, v, zo, vo, s, ps, bflags, mx);
                    } else if (leaf_shadow && tid >= 64 && tid < 64 + nl_grp) {
                        TState<P> sh;
                        init_geometry(p, dm, tid - 64, root_stage, root_pos, stages, crown, sh);
                        if (sh.active && sh.kids_leaf) {
                            for (int c = 0; c < d; ++c)
                                for (int k = 0; k < nx; ++k) sh.xl[c][k] = tl.XL[(sh.pos * d + c) * nx + k];
                            b_local<P, 0, 2>(p, dm, M, z, v, zo, vo, sh, ps, bflags, mx);
                        }
                    }
                } else if (s.active) b_local(p, dm, M, z, v, zo, vo, s, ps, bflags, mx);
                x.stash3(tid, mx);
            });
        } else {   // SP driver: the inner products with the residual of this evaluation ride along (b_local, SPM)
            const real *dz_ = a.dir ? a.dir + (long)inst * a.stride_z : nullptr;
            // Latency regime (register-resident sweep: T threads per node, so phase B keeps one wave busy and leaves the
            // others idle): the residuals of the direction d (below) are formed AT THE SAME TIME by the second wave --
            // thread 64 + li does node li -- instead of after phase B by the first.
            const int shadow0 = 64;
            const bool lop_shadow = a.sp_mode == 2 && nl_grp <= 64 && nthr >= 128;
            x.each([&](int tid, TState<P> &s) {
                real mx[3] = {0., 0., 0.}, sp[3] = {0., 0., 0.};
                s.mx3[0] = s.mx3[1] = s.mx3[2] = 0.;
                if (s.active) {
                    if (a.sp_mode == 2) b_local<P, 2>(p, dm, M, z, v, zo, vo, s, ps, bflags, mx, sp, dz_, dz_ + dm.nz);
                    else b_local<P, 1>(p, dm, M, z, v, zo, vo, s, ps, bflags, mx, sp);
                } else if (lop_shadow && tid >= shadow0 && tid < shadow0 + nl_grp) {
                    TState<P> sh;
                    init_geometry(p, dm, tid - shadow0, root_stage, root_pos, stages, crown, sh);
                    double acc[4] = {0., 0., 0., 0.};
                    if (sh.active) lop_local(p, dm, M, nullptr, dz_, nullptr, sh, gamma, sigma, acc);
                    s.mx3[0] = acc[2]; s.mx3[1] = acc[3];
                }
                x.stash3(tid, mx);
                x.stash_sum3(tid, sp);
            });
        }
        const int slot = (MODE == MODE_FULL) ? 0 : g;
        x.reduce3();
        x.each([&](int tid, TState<P> &) { if (tid < 3) partials[(long)slot * kPartialStride + tid] = x.red3(tid); });
        if (SP_SUMS) {
            x.reduce_sum3();
            x.each([&](int tid, TState<P> &) { if (tid < 3) partials[(long)slot * kPartialStride + 4 + tid] = x.sum3(tid); });
            if (a.sp_mode == 2) {   // the termination residuals of the direction d itself (K1 step: dp = d), spock_sp.hpp: sp_ctl_cd
                const real *dd_ = a.dir + (long)inst * a.stride_z;
                const bool lop_shadow = !team_local && nl_grp <= 64 && nthr >= 128;   // done next to phase B (above)
                x.each([&](int tid, TState<P> &s) {
                    double acc[4] = {0., 0., 0., 0.};
                    if (lop_shadow) { acc[2] = s.mx3[0]; acc[3] = s.mx3[1]; }
                    else if (team_local) {   // T lanes per node, no exchange between lanes: rounds over the group's nodes
                        const int tm = tid / TT, r = tid - tm * TT;
                        for (int li = tm; li < nl_grp && tm < nteams; li += nteams) {
                            const TeamNode tn = team_node(dm, d, li, nl_grp, root_stage, root_pos, stages, crown);
                            tlop(p, dm, M, nullptr, dd_, nullptr, tn, r, gamma, sigma, acc);
                        }
                    } else if (s.active) lop_local(p, dm, M, nullptr, dd_, nullptr, s, gamma, sigma, acc);
                    const double mx[3] = {acc[2], acc[3], 0.};
                    x.stash3(tid, mx);
                });
                x.reduce3();
                x.each([&](int tid, TState<P> &) {
                    if (tid == 0) { partials[(long)slot * kPartialStride + 3] = x.red3(0); partials[(long)slot * kPartialStride + 7] = x.red3(1); }
                });
            }
        }
        stamp();   // phase B done
    };

    // ---- the kernel's sequence
    if (flat && crown_blk) return;   // the crown's node-local phases run in the flat kernels
    if (MODE == MODE_FULL || MODE == MODE_SUB_A) {
        phase_a(z, v, zo, nullptr);
        if (crown_blk) return;
        if (MODE == MODE_FULL) x.barrier();   // zbar.s written by the parents is read by the children in phase B
        stamp();   // phase A done
    }
    if (MODE == MODE_FULL || MODE == MODE_SUB_A || MODE == MODE_CROWN) backward();
    if (MODE != MODE_SUB_A) forward();
    if (MODE == MODE_FULL || MODE == MODE_SUB_B || MODE == MODE_SUB_BA) phase_b();
    if (MODE == MODE_SUB_BA) {
        // Fused with the first half of the NEXT iteration, whose input is this iteration's output (z <- zbar at lambda = 1,
        // cp.jl:34-52): phase A on (zbar, vbar) and the backward sweep.  Its part of the next zbar goes to a third z buffer --
        // this launch's other workgroups may still be reading z.  Everything phase A reads was written by this thread, by
        // this workgroup (hence the barrier) or by an earlier launch, with one exception: the crown's workgroup needs
        // vbar.v2 of the subtree roots, which their own workgroups produce in this same launch; it recomputes them.
        x.barrier();
        real *zn = a.z_next + (long)inst * a.stride_z;
        real *v2x = crown_blk ? a.v2x + (long)inst * dm.nsub : nullptr;
        const ProxScal ps = make_ps();
        if (crown_blk)
            x.each([&](int, TState<P> &s) {
                if (s.active && s.kids_gleaf)
                    for (int c = 0; c < d; ++c) v2x[s.pos * d + c] = risk_v2_only(p, dm, M, z, v, zo, s.c0 + c, ps);
            });
        if (crown_blk && team_local) x.barrier();   // other lanes of the node's team read them
#ifdef SPOCK_HOST_EMU
        if (std::getenv("SPOCK_EMU_NO_V2FIX")) v2x = nullptr;   // negative test: the emulation must expose the hazard the fix removes
#endif
        if (team_reg && !crown_blk) x.each([&](int tid, TState<P> &s) { team_reg_init(p, dm, M, root_stage, stages, team, tid, s, 1); });
        phase_a(zo, vo, zn, v2x);
        if (crown_blk) return;
        stamp();
        backward();
    }
    dump_stamps();
}

}  // namespace spock
