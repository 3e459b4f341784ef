// tests/emu/emu.cpp -- HOST EMULATION HARNESS (test infrastructure, never shipped, never loaded by the
// product package).  It compiles the very same per-thread / per-workgroup sources as the gfx950 kernels
// (spock_device.hpp, spock_group.hpp) with SPOCK_HOST_EMU and runs the "threads" of a workgroup one after
// the other between barriers, and it implements spock_rt.hpp on host memory, so that the whole driver in
// spock_abi.cpp can be debugged (and run under sanitizers) on a machine without a GPU.  Linking
// spock_abi.cpp + spock_host.cpp + this file gives tests/emu/libspock_emu.so.
//
// SPOCK_EMU_REVERSE=1 runs the threads of every phase in reverse order: a result that changes with the
// order betrays a missing barrier.  SPOCK_EMU_FORCE_DYN=1 forces the run-time-dimension variant.
#define SPOCK_HOST_EMU 1
#include <array>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../spock.jl_amd/csrc/spock_group.hpp"
#include "../../spock.jl_amd/csrc/spock_flat_local.hpp"
#include "../../spock.jl_amd/csrc/spock_igroup.hpp"
#include "../../spock.jl_amd/csrc/spock_ilevel.hpp"
#include "../../spock.jl_amd/csrc/spock_rgroup.hpp"
#include "../../spock.jl_amd/csrc/spock_build.hpp"
#include "../../spock.jl_amd/csrc/spock_sp.hpp"
#include "../../spock.jl_amd/csrc/spock_team_local.hpp"
#include "../../spock.jl_amd/csrc/spock_rt.hpp"

namespace spock {

static bool env_on(const char *k) { const char *e = std::getenv(k); return e && e[0] == '1'; }

template <class P>
struct EmuExec {
    std::vector<TState<P>> st;
    std::vector<real> ex, lmv;
    std::vector<std::array<real, 3>> m;
    int fl = 0, nthreads_v = 0;
    bool reverse = false;
    EmuExec(int thr, int ex_doubles, size_t nmats) : st(thr), ex((size_t)ex_doubles + 8, 0.), lmv(nmats + 8, 0.), m(thr), nthreads_v(thr), reverse(env_on("SPOCK_EMU_REVERSE")) {}
    template <class F>
    void each(F &&f) {
        if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t]);
        else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t]);
    }
    void barrier() {}
    void team_sync() {}
    // role waves: every thread up to the barrier, then every thread behind it
    template <class F>
    void phased(F &&f) {
        for (int ph = 0; ph < 2; ++ph) {
            if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t], ph == 0, ph == 1);
            else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t], ph == 0, ph == 1);
        }
    }
    void sync() {}
    real *lds() { return ex.data(); }
    real *lds_mats() { return lmv.data(); }
    int &flag() { return fl; }
    void stash3(int tid, const real *mx) { m[tid] = {mx[0], mx[1], mx[2]}; }
    real r3[3] = {0., 0., 0.};
    int nthreads() const { return nthreads_v; }
    unsigned long long stamps_v[128] = {0};
    unsigned long long *stamp_lds() { return stamps_v; }
    void clocks(unsigned long long *d) { d[0] = 0; d[1] = 0; }
    void reduce3() {
        for (int k = 0; k < 3; ++k) r3[k] = 0.;
        for (int t = 0; t < nthreads_v; ++t) for (int k = 0; k < 3; ++k) r3[k] = spk_max_nan(r3[k], m[t][k]);
    }
    real red3(int k) const { return r3[k]; }
    std::vector<std::array<real, 3>> sm3;
    real s3[3] = {0., 0., 0.};
    void stash_sum3(int tid, const real *v) { if (sm3.size() != m.size()) sm3.assign(m.size(), {0., 0., 0.}); sm3[tid] = {v[0], v[1], v[2]}; }
    void reduce_sum3() {
        for (int k = 0; k < 3; ++k) s3[k] = 0.;
        for (size_t t = 0; t < sm3.size(); ++t) for (int k = 0; k < 3; ++k) s3[k] += sm3[t][k];
    }
    real sum3(int k) const { return s3[k]; }
};

static inline int round64(int t) { return t < 64 ? 64 : ((t + 63) / 64) * 64; }

template <class P, int MODE>
static void run_blocks(const KArgs &a, int gx, int thr) {
    // SPOCK_EMU_REVERSE_BLOCKS=1 runs the workgroups of a launch in the opposite order (the crown's extra workgroup first):
    // a result must not depend on the order in which the workgroups of one launch run
    const bool rev = env_on("SPOCK_EMU_REVERSE_BLOCKS");
    for (int inst = 0; inst < a.batch; ++inst)
        for (int gi = 0; gi < gx; ++gi) {
            const int g = rev ? gx - 1 - gi : gi;
            EmuExec<P> x(thr, a.ex_doubles, (size_t)a.dm.N * a.dm.sw_stride + (size_t)a.dm.d * a.dm.nx * a.dm.nu);
            group_body<P, MODE>(x, a, inst, g);
        }
}

template <class P>
static int launch_variant(int mode, const KArgs &a_in) {
    KArgs a = a_in;
    const LaunchGeo g = launch_geometry(mode, a.dm, a.batch);
    a.lds_mats = env_on("SPOCK_EMU_NO_LDS_MATS") ? 0 : g.lds_mats;
    a.team = g.team; a.team_reg = g.team_reg; a.ex_doubles = g.ex_doubles;
    a.xl_global = (g.team && g.xl_global && a.xlbuf) ? 1 : 0; a.xl_stride = g.xl_stride;
    if (g.team && g.xl_global && !a.xlbuf) return 9;
    const int thr = g.threads, gx = g.grid_x;
    if (thr > (P::TEAMLOCAL ? 1024 : 512)) return 9;
    switch (mode) {
        case MODE_FULL: run_blocks<P, MODE_FULL>(a, gx, thr); break;
        case MODE_SUB_A: run_blocks<P, MODE_SUB_A>(a, gx, thr); break;
        case MODE_CROWN: run_blocks<P, MODE_CROWN>(a, gx, thr); break;
        case MODE_SUB_B: run_blocks<P, MODE_SUB_B>(a, gx, thr); break;
        case MODE_FINALIZE: run_blocks<P, MODE_FINALIZE>(a, gx, thr); break;
        case MODE_SUB_BA: run_blocks<P, MODE_SUB_BA>(a, gx, thr); break;
        case MODE_SUB_B_SP: run_blocks<P, MODE_SUB_B_SP>(a, gx, thr); break;
        case MODE_SUB_B_LAT: run_blocks<P, P::REGROWS ? MODE_SUB_B_LAT : MODE_SUB_B>(a, gx, thr); break;
        case MODE_FULL_SP: run_blocks<P, MODE_FULL_SP>(a, gx, thr); break;
        default: return 1;
    }
    return 0;
}


bool sweep_kernels_apply(const Dims &) { return false; }   // the matrix-core sweep kernels are gfx950 code: the emulation runs group_body's sweeps

// flat node-local kernels (spock_flat_local.hpp)
template <class P, int MODE>
static void run_flat_blocks(const KArgs &a, int gx, int thr) {
    const bool rev = env_on("SPOCK_EMU_REVERSE_BLOCKS");
    for (int inst = 0; inst < a.batch; ++inst)
        for (int gi = 0; gi < gx; ++gi) {
            const int g = rev ? gx - 1 - gi : gi;
            EmuExec<P> x(thr, a.ex_doubles, 8);
            flat_body<P, MODE>(x, a, inst, g);
        }
}
template <class P>
static int launch_flat(int mode, const KArgs &a_in) {
    KArgs a = a_in;
    const FlatGeo g = flat_geometry(a.dm);
    a.team = g.team; a.ex_doubles = g.ex_doubles; a.flat_chunks = g.chunks;
    switch (mode) {
        case MODE_FLAT_A: run_flat_blocks<P, MODE_FLAT_A>(a, g.chunks, g.threads); break;
        case MODE_FLAT_B: run_flat_blocks<P, MODE_FLAT_B>(a, g.chunks, g.threads); break;
        case MODE_FLAT_B_SP: run_flat_blocks<P, MODE_FLAT_B_SP>(a, g.chunks, g.threads); break;
        case MODE_FLAT_BA: run_flat_blocks<P, MODE_FLAT_BA>(a, g.chunks, g.threads); break;
        default: return 1;
    }
    return 0;
}

// ---- element-major batched regime, emulated: the "threads" of a workgroup run one after the other between barriers -------
template <class P>
struct EmuIExec {
    std::vector<IState<P>> st;
    std::vector<real> ex, rd;
    int lf[64] = {0};
    int nthreads_v;
    bool reverse;
    EmuIExec(int thr, int ex_doubles) : st(thr), ex((size_t)ex_doubles + 8, 0.), rd((size_t)(thr / 64) * 3 * 64 + 8, 0.), nthreads_v(thr), reverse(env_on("SPOCK_EMU_REVERSE")) {}
    template <class F>
    void each(F &&f) {
        if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t]);
        else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t]);
    }
    void barrier() {}
    // role waves: every thread up to the barrier, then every thread behind it
    template <class F>
    void phased(F &&f) {
        for (int ph = 0; ph < 2; ++ph) {
            if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t], ph == 0, ph == 1);
            else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t], ph == 0, ph == 1);
        }
    }
    void sync() {}
    real *lds() { return ex.data(); }
    double *red() { return rd.data(); }
    int *live_flags() { return lf; }
    int nthreads() const { return nthreads_v; }
};

template <class P, int MODE, bool WX = false>
static void run_iblocks(const KArgs &a, int gx, int thr) {
    const bool rev = env_on("SPOCK_EMU_REVERSE_BLOCKS");
    const int t_lo = a.n_tiles > 0 ? a.tile0 : 0, t_hi = a.n_tiles > 0 ? a.tile0 + a.n_tiles : (int)(a.bp / 64);
    for (int tile = t_lo; tile < t_hi; ++tile)
        for (int gi = 0; gi < gx; ++gi) {
            const int g = rev ? gx - 1 - gi : gi;
            EmuIExec<P> x(thr, a.ex_doubles);
            igroup_body<P, MODE, WX>(x, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next, tile, g);
        }
}

template <class P, int MODE, bool WX = false>
static void run_lblocks(const KArgs &a, int gx, int thr, const real *zin, const real *vin, real *zout, real *vout, real *znext) {
    const bool rev = env_on("SPOCK_EMU_REVERSE_BLOCKS");
    const int t_lo = a.n_tiles > 0 ? a.tile0 : 0, t_hi = a.n_tiles > 0 ? a.tile0 + a.n_tiles : (int)(a.bp / 64);
    for (int tile = t_lo; tile < t_hi; ++tile)
        for (int gi = 0; gi < gx; ++gi) {
            const int g = rev ? gx - 1 - gi : gi;
            EmuIExec<P> x(thr, a.ex_doubles);
            ilevel_body<P, MODE, WX>(x, a, zin, vin, zout, vout, znext, tile, g);
        }
}

template <class P>
static int launch_ivariant(int mode, const KArgs &a_in) {
    KArgs a = a_in;
    const IlvGeo g = ilv_geometry(mode, a.dm, a.grp_level, a.want_xi);
    a.ex_doubles = g.ex_doubles; a.cchunk = g.cchunk; a.n_cchunks = g.n_cchunks;
    a.ls_cnt = g.ls_cnt; a.ls_first = g.ls_first; a.ls_first_child = g.ls_first_child; a.ls_slot = g.ls_slot; a.ls_chunk_k = g.ls_chunk_k;
    switch (mode) {
        case MODE_SUB_A: run_iblocks<P, MODE_SUB_A>(a, g.grid_x, g.threads); break;
        case MODE_CROWN: if (a.want_xi) run_iblocks<P, MODE_CROWN, true>(a, g.grid_x, g.threads); else run_iblocks<P, MODE_CROWN>(a, g.grid_x, g.threads); break;
        case MODE_I_UP: run_iblocks<P, MODE_I_UP>(a, g.grid_x, g.threads); break;
        case MODE_ICROWN_A: run_iblocks<P, MODE_ICROWN_A>(a, g.grid_x, g.threads); break;
        case MODE_SUB_BA: if (a.want_xi) run_iblocks<P, MODE_SUB_BA, true>(a, g.grid_x, g.threads); else run_iblocks<P, MODE_SUB_BA>(a, g.grid_x, g.threads); break;
        case MODE_I_DOWN: if (a.want_xi) run_iblocks<P, MODE_I_DOWN, true>(a, g.grid_x, g.threads); else run_iblocks<P, MODE_I_DOWN>(a, g.grid_x, g.threads); break;
        case MODE_FINALIZE: run_iblocks<P, MODE_FINALIZE>(a, g.grid_x, g.threads); break;
        case MODE_LB: if (a.want_xi) run_lblocks<P, MODE_LB, true>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, nullptr); else run_lblocks<P, MODE_LB>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, nullptr); break;
        case MODE_LBA: if (a.want_xi) run_lblocks<P, MODE_LBA, true>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next); else run_lblocks<P, MODE_LBA>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next); break;
        case MODE_LBA2: if (a.want_xi) run_lblocks<P, MODE_LBA2, true>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next); else run_lblocks<P, MODE_LBA2>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next); break;
        case MODE_LA: run_lblocks<P, MODE_LA>(a, g.grid_x, g.threads, nullptr, nullptr, a.z_out, a.v_out, a.z_next); break;
        case MODE_LA0: run_lblocks<P, MODE_LA0>(a, g.grid_x, g.threads, a.z_in, a.v_in, a.z_out, nullptr, nullptr); break;
        default: return 1;
    }
    return 0;
}

struct EmuBuildExec {
    template <class F> void par(int n, F &&f) { for (int i = 0; i < n; ++i) f(i); }
    template <class F> void one(F &&f) { f(); }
};
int launch_build_riccati(const BuildArgs &a_in, void *) {
    const int n = a_in.n_inst > 0 ? a_in.n_inst : 1;
    const Dims &dm = a_in.dm;
    for (int b = 0; b < n; ++b) {     // one "workgroup" per instance (spock_bkernels.hip)
        BuildArgs a = a_in;
        a.mats = a_in.mats + (long)b * a_in.mats_stride;
        a.A = a_in.A + (long)b * dm.d * dm.nx * dm.nx; a.B = a_in.B + (long)b * dm.d * dm.nx * dm.nu;
        a.ws = a_in.ws + (long)b * build_ws_doubles(dm); a.status = a_in.status + b;
        EmuBuildExec x;
        build_riccati_body(x, a);
    }
    return 0;
}

int launch_transpose(real *dst, const real *src, long rows, long cols, long ld_src, long ld_dst, void *) {
    for (long r = 0; r < rows; ++r) for (long c = 0; c < cols; ++c) dst[c * ld_dst + r] = src[r * ld_src + c];
    return 0;
}

int launch_ilv_collect(real *const *z3, real *const *v2, const Ctl *ctl, long nz, long nv, long bp, int batch, void *) {
    for (int b = 0; b < batch; ++b) {
        const int rp = ctl[b].result_parity;
        if (rp % 3) for (long e = 0; e < nz; ++e) z3[0][e * bp + b] = z3[rp % 3][e * bp + b];
        if (rp % 2) for (long e = 0; e < nv; ++e) v2[0][e * bp + b] = v2[1][e * bp + b];
    }
    return 0;
}

#define SPOCK_STATIC_SHAPES(X) X(5, 2, 2, 5) X(5, 5, 2, 5) X(2, 1, 2, 5) X(3, 2, 3, 7)

bool have_static_variant(int nx, int nu, int d, int ny) {
#define X(a, b, c, e) if (nx == a && nu == b && d == c && ny == e) return true;
    SPOCK_STATIC_SHAPES(X)
#undef X
    return false;
}

// ---- role-lane subtree kernel (spock_rgroup.hpp), emulated: threads one after the other between barriers, role step in two phases
template <class P>
struct EmuRExec {
    std::vector<RState<P>> st;
    std::vector<real> ex;
    std::vector<std::array<real, 3>> m;
    int nthreads_v;
    bool reverse;
    real r3[3] = {0., 0., 0.};
    EmuRExec(int thr, int ex_doubles) : st(thr), ex((size_t)ex_doubles + 8, 0.), m(thr), nthreads_v(thr), reverse(env_on("SPOCK_EMU_REVERSE")) {}
    template <class F>
    void each(F &&f) {
        if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t]);
        else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t]);
    }
    void barrier() {}
    template <class F>
    void phased(F &&f) {
        for (int ph = 0; ph < 2; ++ph) {
            if (!reverse) for (int t = 0; t < nthreads_v; ++t) f(t, st[t], ph == 0, ph == 1);
            else for (int t = nthreads_v - 1; t >= 0; --t) f(t, st[t], ph == 0, ph == 1);
        }
    }
    void sync() {}
    real *lds() { return ex.data(); }
    int nthreads() const { return nthreads_v; }
    void stash3(int tid, const real *mx) { m[tid] = {mx[0], mx[1], mx[2]}; }
    void clocks(unsigned long long *d) { d[0] = 0; d[1] = 0; }
    void reduce3() {
        for (int k = 0; k < 3; ++k) r3[k] = 0.;
        for (int t = 0; t < nthreads_v; ++t) for (int k = 0; k < 3; ++k) r3[k] = spk_max_nan(r3[k], m[t][k]);
    }
    real red3(int k) const { return r3[k]; }
    void max3_init() { for (int k = 0; k < 3; ++k) r3[k] = 0.; }
    void max3_add(int, const real *mx) { for (int k = 0; k < 3; ++k) r3[k] = spk_max_nan(r3[k], mx[k]); }
    real max3_get(int k) const { return r3[k]; }
};
template <class P>
static int launch_rvariant(const KArgs &a_in) {
    KArgs a = a_in;
    const Dims &dm = a.dm;
    a.ex_doubles = rgroup_lds_doubles(dm.nx, dm.nu, dm.d, a.want_xi != 0);
    const bool rev = env_on("SPOCK_EMU_REVERSE_BLOCKS");
    const int gx = dm.nsub + 1, thr = kRgroupThreads;
    for (int inst = 0; inst < a.batch; ++inst)
        for (int gi = 0; gi < gx; ++gi) {
            const int g = rev ? gx - 1 - gi : gi;
            EmuRExec<P> x(thr, a.ex_doubles);
            if (a.want_xi) rgroup_body<P, true>(x, a, inst, g); else rgroup_body<P, false>(x, a, inst, g);
        }
    return 0;
}
int launch_rgroup(const KArgs &a, void *) {
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
#define X(a_, b_, c_, e_)                                                                  \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                            \
        return cm == 2 ? launch_rvariant<StaticDims<a_, b_, c_, e_, 2>>(a) : cm == 1 ? launch_rvariant<StaticDims<a_, b_, c_, e_, 1>>(a) : 9;
    SPOCK_STATIC_SHAPES(X)
#undef X
    return 9;
}

int launch_group(int mode, const KArgs &a, void *) {
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
    if (mode == MODE_SUB_BA_R) return launch_rgroup(a, nullptr);
    if (a.ilv) {
#define X(a_, b_, c_, e_)                                                                  \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                            \
        return cm == 2 ? launch_ivariant<StaticDims<a_, b_, c_, e_, 2>>(mode, a) : cm == 1 ? launch_ivariant<StaticDims<a_, b_, c_, e_, 1>>(mode, a) : 9;
        SPOCK_STATIC_SHAPES(X)
#undef X
        return 9;
    }
    if (mode >= MODE_FLAT_A && mode <= MODE_FLAT_BA) {
        if (cm != 2 || !team_local_applies(dm)) return 9;
        return dm.d == 2 ? launch_flat<FlatDims<2>>(mode, a) : dm.d == 3 ? launch_flat<FlatDims<3>>(mode, a) : launch_flat<FlatDims<4>>(mode, a);
    }
    if (!env_on("SPOCK_EMU_FORCE_DYN")) {
#define X(a_, b_, c_, e_)                                                                  \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                            \
        return cm == 2 ? launch_variant<StaticDims<a_, b_, c_, e_, 2>>(mode, a) : cm == 1 ? launch_variant<StaticDims<a_, b_, c_, e_, 1>>(mode, a)    \
                             : launch_variant<StaticDims<a_, b_, c_, e_, 0>>(mode, a);
        SPOCK_STATIC_SHAPES(X)
#undef X
    }
    if (cm == 2 && launch_geometry(mode == MODE_FINALIZE ? MODE_CROWN : mode, dm, a.batch).team_local) return launch_variant<DynDims<2, true>>(mode, a);
    return cm == 2 ? launch_variant<DynDims<2>>(mode, a) : cm == 1 ? launch_variant<DynDims<1>>(mode, a) : launch_variant<DynDims<0>>(mode, a);
}

// ---- SP kernels, emulated ---------------------------------------------------------------------------
int sp_flat_blocks(long nr) {
    long b = (nr + 1023) / 1024;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
int sp_lop_blocks(const Dims &dm) { return dm.t0 > 0 ? dm.nsub + 1 : 1; }

template <class P>
static int emu_lop(const SpArgs &a) {
    const Dims &dm = a.dm;
    const P p(dm);
    const int sub_thr = group_threads(dm.d, dm.sub_stages), crown_thr = group_threads(dm.d, dm.t0 + 1);
    const int nthr = round64(dm.t0 > 0 ? (sub_thr > crown_thr ? sub_thr : crown_thr) : group_threads(dm.d, dm.N));
    for (int inst = 0; inst < a.batch; ++inst) {
        if (a.ctl[inst].done) continue;
        for (int g = 0; g < sp_lop_blocks(dm); ++g) {
            const bool split = dm.t0 > 0, crown = split && g == dm.nsub;
            const long o = (long)inst * a.nr;
            const real *A = nullptr, *B = nullptr, *B2 = nullptr;
            if (a.op == SP_LOP_R) { A = a.R + o; B = a.R + o; }
            else if (a.op == SP_LOP_RT) { A = a.RW + o; B = a.RW + o; B2 = a.D + o; }
            else { B = a.DP + o; }
            real tot[4] = {0., 0., 0., 0.};
            if (P::TEAMLOCAL) {
                int T = 8; while (T < dm.nx || T < dm.nu) T *= 2;
                const int stages = crown ? dm.t0 + 1 : (split ? dm.sub_stages : dm.N), nl_grp = group_threads(dm.d, stages);
                for (int li = 0; li < nl_grp; ++li)
                    for (int r = 0; r < T; ++r) {
                        const TeamNode tn = team_node(dm, p.d(), li, nl_grp, (split && !crown) ? dm.t0 : 0, (split && !crown) ? g : 0, stages, crown);
                        real acc[4] = {0., 0., 0., 0.};
                        tlop(p, dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), A, B, B2, tn, r, a.ctl[inst].gamma, a.ctl[inst].sigma, acc);
                        tot[0] += acc[0]; tot[1] += acc[1];
                        tot[2] = spk_max_nan(tot[2], acc[2]); tot[3] = spk_max_nan(tot[3], acc[3]);
                    }
            } else
            for (int tid = 0; tid < nthr; ++tid) {
                TState<P> s;
                init_geometry(p, dm, tid, (split && !crown) ? dm.t0 : 0, (split && !crown) ? g : 0,
                              crown ? dm.t0 + 1 : (split ? dm.sub_stages : dm.N), crown, s);
                real acc[4] = {0., 0., 0., 0.};
                if (s.active) lop_local(p, dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), A, B, B2, s, a.ctl[inst].gamma, a.ctl[inst].sigma, acc);
                tot[0] += acc[0]; tot[1] += acc[1];
                tot[2] = spk_max_nan(tot[2], acc[2]); tot[3] = spk_max_nan(tot[3], acc[3]);
            }
            real *dst = a.partL + ((long)inst * a.n_partL + g) * kSpSums;
            for (int k = 0; k < 4; ++k) dst[k] = tot[k];
        }
    }
    return 0;
}

int launch_sp(const SpArgs &a, void *) {
    const Dims &dm = a.dm;
    if ((a.op >= SP_FLAT1 && a.op <= SP_FLAT5) || a.op == SP_FLATW) {
        for (int inst = 0; inst < a.batch; ++inst) {
            const int done = a.ctl[inst].done;
            if (done == 1 || (done && a.op != SP_FLAT5)) continue;   // done == 2: the state update of the final iteration still runs
            const SpCtl c = a.sp[inst];
            const double rho2 = c.crho[(c.head + 1) % 3], rho3 = c.crho[(c.head + 2) % 3];
            for (int blk = 0; blk < a.n_partF; ++blk) {
                real sums[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
                for (long e = blk; e < a.nr; e += a.n_partF) sp_flat_elem(a, c, inst, e, sums, rho2, rho3);
                real *dst = a.partF + ((long)inst * a.n_partF + blk) * kSpSums;
                if (a.op == SP_FLAT1 || a.op == SP_FLAT2) for (int k = 0; k < 9; ++k) dst[k] = sums[k];
            }
        }
        return 0;
    }
    if (a.op == SP_CTL_M) {   // closed loop on the device (spock_kernels.hip: spock_sp_ctl_kernel)
        for (int inst = 0; inst < a.batch; ++inst) {
            Ctl &ctl = a.ctl[inst];
            if (ctl.done != 2) continue;
            const int t = a.mpc_t[inst];
            const real *z = a.Z + (long)inst * a.nr;
            real *x0 = a.mpc_x0 + (long)inst * dm.nx;
            const int w = a.mpc_w[(long)inst * a.mpc_steps + t];
            std::vector<real> xn(dm.nx);
            for (int r = 0; r < dm.nx; ++r) {
                if (t == 0) a.mpc_xt[((long)inst * (a.mpc_steps + 1)) * dm.nx + r] = x0[r];
                xn[r] = mpc_advance_row(dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), z, 1, x0, w, r);
            }
            for (int r = 0; r < dm.nu; ++r) a.mpc_ut[((long)inst * a.mpc_steps + t) * dm.nu + r] = z[dm.zu + r];
            for (int r = 0; r < dm.nx; ++r) { x0[r] = xn[r]; a.mpc_xt[((long)inst * (a.mpc_steps + 1) + t + 1) * dm.nx + r] = xn[r]; }
            sp_mpc_next(ctl, a.sp[inst], a.mpc_t + inst, a.mpc_steps, a.mpc_res + (long)inst * a.mpc_steps * kMpcRec);
        }
        return 0;
    }
    if (a.op >= SP_CTL_A) {
        for (int inst = 0; inst < a.batch; ++inst) {
            Ctl &ctl = a.ctl[inst];
            if (ctl.done) { if (a.op == SP_CTL_A && ctl.done == 2) ctl.done = 1; if (a.op == SP_CTL_E && ctl.done == 3) ctl.done = 0; continue; }
            if (a.op == SP_CTL_E) continue;
            SpCtl &c = a.sp[inst];
            const real *pf = a.partF + (long)inst * a.n_partF * kSpSums, *pl = a.partL + (long)inst * a.n_partL * kSpSums;
            real f[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, l[2] = {0., 0.}, m[2] = {0., 0.};
            for (int r = 0; r < a.n_partF; ++r) for (int k = 0; k < 9; ++k) f[k] += pf[(long)r * kSpSums + k];
            for (int r = 0; r < a.n_partL; ++r) {
                l[0] += pl[(long)r * kSpSums]; l[1] += pl[(long)r * kSpSums + 1];
                m[0] = spk_max_nan(m[0], pl[(long)r * kSpSums + 2]); m[1] = spk_max_nan(m[1], pl[(long)r * kSpSums + 3]);
            }
            real g2[2] = {0., 0.};
            if ((a.op == SP_CTL_A || a.op == SP_CTL_C || a.op == SP_CTL_CD) && a.fused_lop) {
                const real *pg = a.partG + (long)inst * a.n_partG * kPartialStride;
                l[0] = l[1] = 0.; m[0] = m[1] = 0.;
                for (int r = 0; r < a.n_partG; ++r) {
                    l[0] += -ctl.gamma * pg[(long)r * kPartialStride + 4] - ctl.sigma * pg[(long)r * kPartialStride + 5];
                    l[1] += pg[(long)r * kPartialStride + 6];
                    m[0] = spk_max_nan(m[0], pg[(long)r * kPartialStride + 0]); m[1] = spk_max_nan(m[1], pg[(long)r * kPartialStride + 1]);
                    g2[0] = spk_max_nan(g2[0], pg[(long)r * kPartialStride + 3]); g2[1] = spk_max_nan(g2[1], pg[(long)r * kPartialStride + 7]);
                }
            }
            real *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
            if (a.op == SP_CTL_CD && a.variant == 1) sp_ctl_trial(c, ctl, l[0], l[1], m[0], m[1], g2[0], g2[1], a.tau, a.trial, a.last_trial, hist, a.hist_cap);
            else if (a.op == SP_CTL_CD) sp_ctl_cd(c, ctl, l[0], l[1], m[0], m[1], g2[0], g2[1], hist, a.hist_cap);
            else if (a.op == SP_CTL_A && a.variant == 1) sp_ctl_a_ls(c, f, l[0], m[0], m[1]);
            else if (a.op == SP_CTL_A) sp_ctl_a(c, f, l[0], m[0], m[1]);
            else if (a.op == SP_CTL_B) sp_ctl_b(c, f);
            else if (a.op == SP_CTL_C) sp_ctl_c(c, ctl, l[0], l[1]);
            else sp_ctl_d(c, ctl, m[0], m[1], a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr, a.hist_cap);
        }
        return 0;
    }
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
    if (!env_on("SPOCK_EMU_FORCE_DYN")) {
#define X(a_, b_, c_, e_)                                                  \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)            \
        return cm == 2 ? emu_lop<StaticDims<a_, b_, c_, e_, 2>>(a) : cm == 1 ? emu_lop<StaticDims<a_, b_, c_, e_, 1>>(a) : emu_lop<StaticDims<a_, b_, c_, e_, 0>>(a);
        SPOCK_STATIC_SHAPES(X)
#undef X
    }
    if (cm == 2 && team_local_applies(dm)) return emu_lop<DynDims<2, true>>(a);
    return cm == 2 ? emu_lop<DynDims<2>>(a) : cm == 1 ? emu_lop<DynDims<1>>(a) : emu_lop<DynDims<0>>(a);
}

int launch_mpc_advance(const MpcArgs &a, void *) {
    const Dims &dm = a.dm;
    for (int inst = 0; inst < a.batch; ++inst) {
        const real *z = a.z + (long)inst * a.stride_z;
        real *x0 = a.x0 + (long)inst * dm.nx;
        const int w = a.w_seq[(long)inst * a.steps + a.t];
        std::vector<real> xn(dm.nx);
        for (int r = 0; r < dm.nx; ++r) {
            if (a.t == 0) a.x_traj[((long)inst * (a.steps + 1)) * dm.nx + r] = x0[r];
            xn[r] = mpc_advance_row(dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), z, a.stride_e, x0, w, r);
        }
        for (int r = 0; r < dm.nu; ++r) a.u_traj[((long)inst * a.steps + a.t) * dm.nu + r] = z[(long)(dm.zu + r) * a.stride_e];
        for (int r = 0; r < dm.nx; ++r) { x0[r] = xn[r]; a.x_traj[((long)inst * (a.steps + 1) + a.t + 1) * dm.nx + r] = xn[r]; }
    }
    return 0;
}

// ---- spock_rt.hpp on host memory ---------------------------------------------------------------------
namespace rt {
int set_device(int) { return 0; }
int get_device(int *d) { *d = 0; return 0; }
int stream_create(void **s) { *s = (void *)0x1; return 0; }
int stream_destroy(void *) { return 0; }
int dmalloc(void **p, size_t bytes) { *p = std::malloc(bytes ? bytes : 8); return *p ? 0 : 2; }
int dfree(void *p) { std::free(p); return 0; }
int dzero(void *p, size_t bytes, void *) { std::memset(p, 0, bytes); return 0; }
int h2d(void *dst, const void *src, size_t bytes, void *) { std::memcpy(dst, src, bytes); return 0; }
int d2h(void *dst, const void *src, size_t bytes, void *) { std::memcpy(dst, src, bytes); return 0; }
int d2d(void *dst, const void *src, size_t bytes, void *) { std::memmove(dst, src, bytes); return 0; }
int sync(void *) { return 0; }
struct Ev { std::chrono::steady_clock::time_point t; };
int event_create(void **e) { *e = new Ev(); return 0; }
int event_destroy(void *e) { delete (Ev *)e; return 0; }
int stream_wait_event(void *, void *) { return 0; }
int event_record(void *e, void *) { ((Ev *)e)->t = std::chrono::steady_clock::now(); return 0; }
int event_elapsed(float *ms, void *e0, void *e1) {
    *ms = std::chrono::duration<float, std::milli>(((Ev *)e1)->t - ((Ev *)e0)->t).count();
    return 0;
}
bool graphs_supported() { return false; }
bool timing_supported() { return false; }
int capture_begin(void *) { return 1; }
int capture_end(void *, void **) { return 1; }
int graph_launch(void *, void *) { return 1; }
int graph_destroy(void *) { return 0; }
const char *err(int code) { return code == 0 ? "ok" : (code == 9 ? "group larger than one workgroup" : "emulation runtime error"); }
}  // namespace rt
}  // namespace spock
