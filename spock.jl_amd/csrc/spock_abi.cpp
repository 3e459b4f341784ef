// spock_abi.cpp -- the C ABI of include/spock_amd.h: handle management and the device-resident
// Chambolle-Pock driver (run_cp!, /root/reference/src/model_algorithms/cp.jl:188-232).
//
// Driver design: the iteration loop never returns to the host for a decision.  Each instance has a
// control block in device memory (Ctl); the first kernel of iteration k+1 runs the termination test of
// iteration k (should_terminate!, cp.jl:54-123) and, once an instance is done, every later kernel for
// it exits immediately, so the state freezes at exactly the iterate the reference would stop at.  The
// host replays a captured hipGraph of `graph_iters` iterations and polls the control blocks in between.
// "z <- zbar" at lambda = 1 is a pointer swap (cp.jl:34-52): v ping-pongs between two buffers, z rotates through
// three (iteration k reads slot k % 3, writes zbar to (k+1) % 3, and the fused kernel's early phase A of iteration
// k + 1 writes to (k+2) % 3), so the buffer roles repeat with period 6 = `parity` of a launch.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spock_amd.h"
#include "spock_internal.hpp"
#include "spock_rt.hpp"

using namespace spock;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

struct spock_handle_s {
    HostProblem hp;
    BuildProducts bp;
    int alg = SPOCK_ALG_CP, device = 0, batch = 1, max_iter = 5000, graph_iters = 32;
    double L_norm = 10.;
    void *stream = nullptr, *ev0 = nullptr, *ev1 = nullptr, *graph = nullptr;
    // element-major regime: the tiles are split into groups that run their kernel chains on streams of their own -- the crown
    // chain of one group (latency-bound, a handful of workgroups) then overlaps the bottom-layer kernel of the others
    int n_tg = 1;
    std::vector<void *> tg_stream, tg_event;
    void *ev_fork = nullptr;
    void *graph_log = nullptr;   // element-major CP: LOG-mode solves run another kernel variant (KArgs::want_xi), hence another graph
    int want_xi = 0;
    // device memory
    real *d_mats = nullptr, *d_z[3] = {nullptr, nullptr, nullptr}, *d_v[2] = {nullptr, nullptr}, *d_v2x = nullptr;
    bool lat_regime = false;   // few workgroups in flight (register-resident sweep): SUB_B runs as MODE_SUB_B_LAT
    bool rlane = false;        // ... and the fused iteration's SUB_BA as the role-lane kernel MODE_SUB_BA_R (spock_rgroup.hpp)
    bool dyn_on_device = false;   // spock_update_dynamics has run: P, K, Rtilde^-1, ABK of spock_get_build_product come from the device pool
    int sp_variant = 0, sp_max_bt = 8;   // SP: 0 = the reference's run_sp! as it stands, 1 = corrected Anderson + working line search
    bool sp_fused = false;   // SP: the inner products with the residuals come out of phase B instead of out of LOP kernels
    bool fused = false;   // CP iteration = CROWN + SUB_BA (after one SUB_A) instead of SUB_A + CROWN + SUB_B
    // lane-team shapes (large nx / nu): the node-local phases as flat kernels over chunks of nodes (spock_flat_local.hpp); the subtree /
    // crown kernels then run the sweeps only.  CP: FLAT_A once, then SUB_A + CROWN + SUB_B + FLAT_BA per iteration (fused) or
    // FLAT_A + SUB_A + CROWN + SUB_B + FLAT_B
    bool flat_local = false;
    int flat_chunks = 0;
    long mats_stride = 0;   // > 0: every instance has its own matrix pool of this many elements (spock_options_t.per_instance_dynamics)
    real *d_dbuf = nullptr, *d_qroot = nullptr, *d_cbuf = nullptr, *d_x0 = nullptr, *d_xlbuf = nullptr;
    double *d_partials = nullptr, *d_hist = nullptr;   // residual maxima / sums and the residual log stay double
    // element-major batched regime (spock_igroup.hpp): the iterate as [element][bp]; d_z / d_v then serve as instance-major
    // staging buffers of the API (get / set state)
    bool ilv = false;
    int layout_reason = 0;   // spock_sizes_t.layout_reason
    long bpad = 0;   // batch rounded up to a multiple of 64
    real *i_z[3] = {nullptr, nullptr, nullptr}, *i_v[2] = {nullptr, nullptr}, *i_cq = nullptr;
    bool stage_valid = false;   // d_z[0] / d_v[0] hold the instance-major copy of the current state
    Ctl *d_ctl = nullptr;
    unsigned long long *d_dbg = nullptr;
    int hist_cap = 0;
    // single-instance buffers of the operator-level entry points
    real *o_z[2] = {nullptr, nullptr}, *o_v[2] = {nullptr, nullptr}, *o_dbuf = nullptr, *o_qroot = nullptr,
         *o_cbuf = nullptr, *o_x0 = nullptr;
    double *o_partials = nullptr;
    Ctl *o_ctl = nullptr;
    // SP (SuperMann) state: ten [batch][nz+nv] vectors, the circular Anderson buffers, per-instance scalars
    real *s_vec[10] = {nullptr}, *s_MP = nullptr, *s_Q = nullptr;
    double *s_partF = nullptr, *s_partL = nullptr;
    SpCtl *s_ctl = nullptr;
    // closed loop chained on the device (SP_CTL_M): buffers kept in the handle between calls, a graph of their own
    void *graph_mpc = nullptr;
    int *m_w = nullptr, *m_t = nullptr, m_cap_steps = 0, m_graph_steps = 0;
    real *m_xt = nullptr, *m_ut = nullptr;
    double *m_res = nullptr;
    std::vector<SpCtl> h_sp;
    std::vector<Ctl> h_ctl;
    double last_ms = 0.;
    int last_iters = 0;
    std::vector<void *> allocs;
};

namespace {

#define RTCHECK(expr)                                                                          \
    do {                                                                                       \
        int _e = (expr);                                                                       \
        if (_e != 0) return fail(SPOCK_ERR_HIP, std::string(#expr) + ": " + rt::err(_e));      \
    } while (0)

// Every ABI entry point runs with the handle's device current and puts the caller's current device back on return
// (a process may hold handles on several GPUs, or use another device through torch).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (rt::get_device(&prev) != 0) prev = -1;
        if (prev != dev) changed = (rt::set_device(dev) == 0);
    }
    ~DeviceGuard() { if (changed && prev >= 0) rt::set_device(prev); }
};

template <class T>
int dev_alloc(spock_handle_s *h, T **p, size_t count) {
    void *q = nullptr;
    int e = rt::dmalloc(&q, count * sizeof(T));
    if (e != 0) return e;
    h->allocs.push_back(q);
    *p = (T *)q;
    return rt::dzero(q, count * sizeof(T), h->stream);
}

// The C ABI is double; the device path computes and stores in `real` (double in libspock_amd.so, float in libspock_amd_f32.so).
// up / down move `n` elements across the boundary, converting on the host when the two differ.
int up(spock_handle_s *h, real *dst, const double *src, size_t n) {
    if (kRealIsDouble) return rt::h2d(dst, src, n * sizeof(real), h->stream);
    std::vector<real> tmp(n);
    for (size_t k = 0; k < n; ++k) tmp[k] = (real)src[k];
    int e = rt::h2d(dst, tmp.data(), n * sizeof(real), h->stream);
    if (!e) e = rt::sync(h->stream);   // tmp must outlive the copy
    return e;
}
int down(spock_handle_s *h, double *dst, const real *src, size_t n) {
    if (kRealIsDouble) return rt::d2h(dst, src, n * sizeof(real), h->stream);
    std::vector<real> tmp(n);
    int e = rt::d2h(tmp.data(), src, n * sizeof(real), h->stream);
    if (!e) e = rt::sync(h->stream);
    for (size_t k = 0; k < n && !e; ++k) dst[k] = (double)tmp[k];
    return e;
}

KArgs make_args(spock_handle_s *h, bool op, int parity, unsigned flags) {
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.dm = h->bp.dm;
    a.mats = h->d_mats;
    a.mats_stride = op ? 0 : h->mats_stride;   // (the operator entry points work on instance 0's pool)
    if (!op) {
        a.z_in = h->d_z[parity % 3]; a.v_in = h->d_v[parity % 2];
        a.z_out = h->d_z[(parity + 1) % 3]; a.v_out = h->d_v[(parity + 1) % 2];
        a.z_next = h->d_z[(parity + 2) % 3]; a.v2x = h->d_v2x;
        a.dbuf = h->d_dbuf; a.qroot = h->d_qroot; a.cbuf = h->d_cbuf; a.partials = h->d_partials; a.ctl = h->d_ctl; a.x0 = h->d_x0;
        a.hist = h->d_hist; a.hist_cap = h->hist_cap; a.batch = h->batch; a.dbg = h->d_dbg;
    } else {
        a.z_in = h->o_z[0]; a.v_in = h->o_v[0]; a.z_out = h->o_z[1]; a.v_out = h->o_v[1];
        a.dbuf = h->o_dbuf; a.qroot = h->o_qroot; a.cbuf = h->o_cbuf; a.partials = h->o_partials; a.ctl = h->o_ctl; a.x0 = h->o_x0;
        a.hist = nullptr; a.hist_cap = 0; a.batch = 1;
    }
    a.stride_z = h->bp.dm.nz; a.stride_v = h->bp.dm.nv;
    if (!op && h->ilv) {
        a.z_in = h->i_z[parity % 3]; a.v_in = h->i_v[parity % 2];
        a.z_out = h->i_z[(parity + 1) % 3]; a.v_out = h->i_v[(parity + 1) % 2];
        a.z_next = h->i_z[(parity + 2) % 3];
        a.ilv = 1; a.bp = h->bpad; a.cq = h->i_cq; a.want_xi = h->want_xi;
    }
    if (!op && h->rlane) a.want_xi = h->want_xi;
    a.xlbuf = h->d_xlbuf;
    a.flat_local = (!op && h->flat_local) ? 1 : 0; a.flat_chunks = h->flat_chunks;
    a.parity = parity;
    a.flags = flags;
    return a;
}

// the kernels of one evaluation of the CP operator (+ residuals) on input buffer `parity`
int launch_iteration(spock_handle_s *h, bool op, int parity, unsigned flags, bool phase_a = true, bool phase_b = true) {
    const KArgs a = make_args(h, op, parity, flags);
    if (h->bp.dm.t0 == 0) return launch_group(MODE_FULL, a, h->stream);
    int e = 0;
    if (phase_a) e = launch_group(MODE_SUB_A, a, h->stream);
    if (e == 0) e = launch_group(MODE_CROWN, a, h->stream);
    if (e == 0 && phase_b) e = launch_group(MODE_SUB_B, a, h->stream);
    return e;
}

// CP driver: the kernels of iteration k (k taken modulo 6).  Fused sequence: SUB_A of iteration 0 is launched once per
// solve (cp_prologue); every iteration is then CROWN + SUB_BA, the latter already running phase A and the backward
// sweep of iteration k + 1.
int launch_cp_iteration(spock_handle_s *h, int k, unsigned flags = 0, void *stream = nullptr, int tile0 = 0, int n_tiles = 0) {
    KArgs a = make_args(h, false, k % 6, flags);
    if (!stream) stream = h->stream;
    if (h->bp.dm.t0 == 0) return launch_group(MODE_FULL, a, stream);
    if (h->ilv) {   // element-major regime: hierarchy of groups (spock_igroup.hpp): UP the mid layers, TOP, DOWN the mid layers, bottom
        const Dims &dm = h->bp.dm;
        a.tile0 = tile0; a.n_tiles = n_tiles;
        int e = 0;
        for (int lv = dm.n_lv - 2; lv >= 1 && e == 0; --lv) { a.grp_level = lv; e = launch_group(MODE_I_UP, a, stream); }
        a.grp_level = 0;
        if (e == 0) e = launch_group(MODE_CROWN, a, stream);
        for (int lv = 1; lv <= dm.n_lv - 2 && e == 0; ++lv) { a.grp_level = lv; e = launch_group(MODE_I_DOWN, a, stream); }
        a.grp_level = 0;
        if (!dm.ilv_stream) { if (e == 0) e = launch_group(MODE_SUB_BA, a, stream); return e; }
        // level-streaming kernels (spock_ilevel.hpp) for the stages t0 .. N-2: phase B top-down, the bottom stage fused with the
        // next phase A and the backward step, then the next phase A + backward step bottom-up
        // (ilv_fuse2: the stages N-3 and N-2 as one launch, LBA2 -- LB and LA then stop one stage higher)
        const int t_lb = dm.ilv_fuse2 ? dm.N - 4 : dm.N - 3;
        for (int t = dm.t0; t <= t_lb && e == 0; ++t) { a.grp_level = t; e = launch_group(MODE_LB, a, stream); }
        a.grp_level = dm.ilv_fuse2 ? dm.N - 3 : dm.N - 2;
        if (e == 0) e = launch_group(dm.ilv_fuse2 ? MODE_LBA2 : MODE_LBA, a, stream);
        for (int t = t_lb; t >= dm.t0 && e == 0; --t) { a.grp_level = t; e = launch_group(MODE_LA, a, stream); }
        return e;
    }
    int e = 0;
    if (h->flat_local) {   // sweeps in the subtree / crown kernels, node-local phases in the flat kernels
        if (!h->fused) e = launch_group(MODE_FLAT_A, a, stream);
        if (e == 0) e = launch_group(MODE_SUB_A, a, stream);
        if (e == 0) e = launch_group(MODE_CROWN, a, stream);
        if (e == 0) e = launch_group(MODE_SUB_B, a, stream);
        if (e == 0) e = launch_group(h->fused ? MODE_FLAT_BA : MODE_FLAT_B, a, stream);
        return e;
    }
    if (!h->fused) e = launch_group(MODE_SUB_A, a, stream);
    if (e == 0) e = launch_group(MODE_CROWN, a, stream);
    if (e == 0) e = launch_group(h->fused ? (h->rlane ? MODE_SUB_BA_R : MODE_SUB_BA) : (h->lat_regime ? MODE_SUB_B_LAT : MODE_SUB_B), a, stream);
    return e;
}
// G iterations of every instance.  Element-major regime with several tile groups: fork from the handle's stream, one chain of G
// iterations per group on its own stream, join (the same calls build parallel branches when the handle's stream is capturing).
int launch_cp_iterations(spock_handle_s *h, int G) {
    if (!h->ilv || h->n_tg <= 1) {
        int e = 0;
        for (int it = 0; it < G && e == 0; ++it) e = launch_cp_iteration(h, it);
        return e;
    }
    const int tiles = (int)(h->bpad / 64);
    int e = rt::event_record(h->ev_fork, h->stream);
    for (int gi = 0; gi < h->n_tg && e == 0; ++gi) {
        const int t_lo = (int)((long)tiles * gi / h->n_tg), t_hi = (int)((long)tiles * (gi + 1) / h->n_tg);
        if (t_hi <= t_lo) continue;
        void *st = h->tg_stream[gi];
        e = rt::stream_wait_event(st, h->ev_fork);
        for (int it = 0; it < G && e == 0; ++it) e = launch_cp_iteration(h, it, 0, st, t_lo, t_hi - t_lo);
        if (e == 0) e = rt::event_record(h->tg_event[gi], st);
        if (e == 0) e = rt::stream_wait_event(h->stream, h->tg_event[gi]);
    }
    return e;
}
int cp_prologue(spock_handle_s *h, unsigned flags = 0) {
    if (!h->fused) return 0;
    int e = 0;
    if (h->ilv && h->bp.dm.ilv_stream) {
        KArgs a = make_args(h, false, 0, flags);
        for (int t = h->bp.dm.N - 2; t >= h->bp.dm.t0 && e == 0; --t) { a.grp_level = t; e = launch_group(MODE_LA0, a, h->stream); }
    } else e = launch_group(h->flat_local ? MODE_FLAT_A : MODE_SUB_A, make_args(h, false, 0, flags), h->stream);
    if (e == 0 && h->ilv) e = launch_group(MODE_ICROWN_A, make_args(h, false, 0, flags), h->stream);
    return e;
}

int ensure_op_buffers(spock_handle_s *h) {
    if (h->o_ctl) return 0;
    const Dims &dm = h->bp.dm;
    int e = 0;
    for (int k = 0; k < 2 && !e; ++k) { e = dev_alloc(h, &h->o_z[k], (size_t)dm.nz); if (!e) e = dev_alloc(h, &h->o_v[k], (size_t)dm.nv); }
    if (!e) e = dev_alloc(h, &h->o_dbuf, (size_t)dm.nn * dm.nu);
    if (!e) e = dev_alloc(h, &h->o_qroot, (size_t)dm.nsub * dm.nx);
    if (!e) e = dev_alloc(h, &h->o_cbuf, (size_t)(group_threads(dm.d, dm.t0 + 1) + 1) * (dm.nx + dm.nu));
    if (!e) e = dev_alloc(h, &h->o_partials, (size_t)(dm.nsub + 1) * kPartialStride);
    if (!e) e = dev_alloc(h, &h->o_x0, (size_t)dm.nx);
    if (!e) e = dev_alloc(h, &h->o_ctl, 1);
    return e;
}

Ctl fresh_ctl(const spock_handle_s *h, double gamma, double sigma, double tol, int verbose) {
    Ctl c;
    std::memset(&c, 0, sizeof(c));
    c.max_iter = h->max_iter;
    c.log_mode = verbose;
    c.alg = h->alg;
    c.res0[0] = -INFINITY; c.res0[1] = -INFINITY;   // cpock.jl:186
    c.xi1 = c.xi2 = c.xi = NAN;
    c.gamma = gamma; c.sigma = sigma; c.tol = tol;
    return c;
}

// Fused or three-kernel sequence?  The fused kernel saves a launch and the re-read of zbar / vbar, but it is the most
// register-hungry kernel of the library: for the larger shapes (nx = nu = 5 and up) the compiler spills in it and the plain
// sequence is faster.  The choice is a deterministic rule on the shape (measured on MI355X: (nx, nu) = (5, 2) fused wins at
// batch 1 and 1024, (5, 5) / (20, 20) / (64, 16) the plain sequence wins), so that two handles of the same problem always
// run the same kernels and give bitwise identical iterates; spock_sizes_t.fused_iteration reports it.
// SPOCK_AMD_FUSE=1 / 0 forces the choice, SPOCK_AMD_FUSE=time decides by timing six iterations of both (experiments).
int time_cp_sequences(spock_handle_s *h);
int choose_cp_sequence(spock_handle_s *h) {
    const Dims &dm = h->bp.dm;
    if (h->ilv) { h->fused = true; return 0; }   // the element-major regime has the fused sequence only
    if (h->flat_local) {   // flat node-local kernels: phase B of iteration k and phase A of k + 1 in one launch (FLAT_BA)
        h->fused = !(std::getenv("SPOCK_AMD_FLAT_FUSE") && std::getenv("SPOCK_AMD_FLAT_FUSE")[0] == '0');
        return 0;
    }
    h->fused = dm.t0 > 0 && (dm.nx + dm.nu <= 8);
    // (the role-lane kernel MODE_SUB_BA_R, which replaces SUB_BA when few workgroups are in flight, does not have SUB_BA's register
    // problem: with it the fused sequence also serves the larger compile-time shapes, e.g. nx = nu = 5)
    if (!h->fused && h->mats_stride == 0 && rlane_applies(dm, h->batch, true)) h->fused = true;
    if (std::getenv("SPOCK_AMD_NO_FUSE")) { h->fused = false; return 0; }
    if (const char *e = std::getenv("SPOCK_AMD_FUSE")) {
        if (e[0] == 't') return time_cp_sequences(h);
        h->fused = dm.t0 > 0 && (e[0] == '1');
    }
    return 0;
}
int time_cp_sequences(spock_handle_s *h) {
    if (!rt::timing_supported()) return 0;
    const size_t B = (size_t)h->batch;
    const Dims &dm = h->bp.dm;
    Ctl c = fresh_ctl(h, 0.3, 0.3, 0., SPOCK_SILENT);
    c.max_iter = 1 << 30;
    std::vector<Ctl> cs(B, c);
    double best[2] = {1e30, 1e30};
    int e = 0;
    for (int trial = 0; trial < 3 && !e; ++trial)
        for (int f = 0; f < 2 && !e; ++f) {
            h->fused = (f == 1);
            if ((e = rt::h2d(h->d_ctl, cs.data(), B * sizeof(Ctl), h->stream))) break;
            if ((e = rt::event_record(h->ev0, h->stream))) break;
            e = cp_prologue(h);
            for (int it = 0; it < 6 && !e; ++it) e = launch_cp_iteration(h, it);
            if (!e) e = rt::event_record(h->ev1, h->stream);
            if (!e) e = rt::sync(h->stream);
            float ms = 0.f;
            if (!e) e = rt::event_elapsed(&ms, h->ev0, h->ev1);
            if (!e && trial > 0 && ms < best[f]) best[f] = ms;   // trial 0 warms up (code upload, caches)
        }
    if (e) return e;
    h->fused = best[1] <= best[0];
    for (int k = 0; k < 3 && !e; ++k) e = rt::dzero(h->d_z[k], B * dm.nz * sizeof(real), h->stream);
    for (int k = 0; k < 2 && !e; ++k) e = rt::dzero(h->d_v[k], B * dm.nv * sizeof(real), h->stream);
    for (auto &x : cs) x.done = 1;
    if (!e) e = rt::h2d(h->d_ctl, cs.data(), B * sizeof(Ctl), h->stream);
    if (!e) e = rt::sync(h->stream);
    return e;
}

// element-major regime: the API's instance-major view of the state lives in the staging buffers d_z[0] / d_v[0]
int ilv_state_to_stage(spock_handle_s *h) {
    if (h->stage_valid) return 0;
    const Dims &dm = h->bp.dm;
    int e = launch_transpose(h->d_z[0], h->i_z[0], dm.nz, h->batch, h->bpad, dm.nz, h->stream);
    if (!e) e = launch_transpose(h->d_v[0], h->i_v[0], dm.nv, h->batch, h->bpad, dm.nv, h->stream);
    if (!e) h->stage_valid = true;
    return e;
}
int ilv_stage_to_state(spock_handle_s *h) {
    const Dims &dm = h->bp.dm;
    int e = launch_transpose(h->i_z[0], h->d_z[0], h->batch, dm.nz, dm.nz, h->bpad, h->stream);
    if (!e) e = launch_transpose(h->i_v[0], h->d_v[0], h->batch, dm.nv, dm.nv, h->bpad, h->stream);
    return e;
}

// run one operator-style launch sequence: inputs (may be null = zeros) -> op buffers -> outputs
int run_op(spock_handle_s *h, const double *z_in, const double *v_in, const double *zbar_in, const double *x0,
           double gamma, double sigma, unsigned flags, bool phase_a, bool phase_b, double *z_out, double *v_out) {
    const Dims &dm = h->bp.dm;
    DeviceGuard guard(h->device);
    RTCHECK(ensure_op_buffers(h));
    const size_t bz = sizeof(real) * dm.nz, bv = sizeof(real) * dm.nv;
    if (z_in) RTCHECK(up(h, h->o_z[0], z_in, dm.nz)); else RTCHECK(rt::dzero(h->o_z[0], bz, h->stream));
    if (v_in) RTCHECK(up(h, h->o_v[0], v_in, dm.nv)); else RTCHECK(rt::dzero(h->o_v[0], bv, h->stream));
    if (zbar_in) RTCHECK(up(h, h->o_z[1], zbar_in, dm.nz)); else RTCHECK(rt::dzero(h->o_z[1], bz, h->stream));
    RTCHECK(rt::dzero(h->o_v[1], bv, h->stream));
    if (x0) RTCHECK(up(h, h->o_x0, x0, dm.nx)); else RTCHECK(rt::dzero(h->o_x0, sizeof(real) * dm.nx, h->stream));
    Ctl c = fresh_ctl(h, gamma, sigma, 0., SPOCK_SILENT);
    RTCHECK(rt::h2d(h->o_ctl, &c, sizeof(Ctl), h->stream));
    RTCHECK(launch_iteration(h, true, 0, flags | F_NO_CTL, phase_a, phase_b));
    if (z_out) RTCHECK(down(h, z_out, h->o_z[1], dm.nz));
    if (v_out) RTCHECK(down(h, v_out, h->o_v[1], dm.nv));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}

// ---------------------------------------------------------------------------------------------------
// SP driver (run_sp!, src/model_algorithms/sp.jl:358-469): see spock_sp.hpp for the kernel sequence
enum { SV_Z = 0, SV_ZB, SV_ZOLD, SV_R, SV_ROLD, SV_DP, SV_W, SV_WB, SV_RW, SV_D };

SpArgs make_sp_args(spock_handle_s *h, int op) {
    SpArgs a;
    std::memset(&a, 0, sizeof(a));
    const Dims &dm = h->bp.dm;
    a.dm = dm; a.mats = h->d_mats; a.mats_stride = h->mats_stride; a.ctl = h->d_ctl; a.sp = h->s_ctl;
    a.Z = h->s_vec[SV_Z]; a.ZB = h->s_vec[SV_ZB]; a.ZOLD = h->s_vec[SV_ZOLD]; a.R = h->s_vec[SV_R]; a.ROLD = h->s_vec[SV_R];   // r and r_old share a buffer (spock_sp.hpp: FLAT1)
    a.DP = h->s_vec[SV_DP]; a.W = h->s_vec[SV_W]; a.WB = h->s_vec[SV_WB]; a.RW = h->s_vec[SV_RW]; a.D = h->s_vec[SV_D];
    a.MP = h->s_MP; a.Q = h->s_Q; a.partF = h->s_partF; a.partL = h->s_partL;
    a.hist = h->d_hist; a.hist_cap = h->hist_cap;
    a.nr = (long)dm.nz + dm.nv;
    a.n_partF = sp_flat_blocks(a.nr); a.n_partL = sp_lop_blocks(dm);
    a.batch = h->batch; a.op = op;
    a.partG = h->d_partials; a.n_partG = h->flat_local ? h->flat_chunks : (dm.t0 > 0 ? dm.nsub + 1 : 1); a.fused_lop = h->sp_fused ? 1 : 0;
    a.variant = h->sp_variant; a.tau = 1.; a.trial = 0; a.last_trial = 1;
    a.mpc_w = h->m_w; a.mpc_x0 = h->d_x0; a.mpc_xt = h->m_xt; a.mpc_ut = h->m_ut; a.mpc_res = h->m_res; a.mpc_t = h->m_t; a.mpc_steps = h->m_graph_steps;
    return a;
}

// T(in) -> out on the concatenated z|v vectors with the CP kernels (no CP termination logic: F_SP)
int launch_sp_T(spock_handle_s *h, real *in, real *out, int sp_mode = 0, const real *dir = nullptr) {
    const Dims &dm = h->bp.dm;
    KArgs a = make_args(h, false, 0, F_SP);
    a.sp_mode = h->sp_fused ? sp_mode : 0; a.dir = dir;
    a.z_in = in; a.v_in = in + dm.nz; a.z_out = out; a.v_out = out + dm.nz;
    a.stride_z = a.stride_v = (long)dm.nz + dm.nv;
    a.hist = nullptr; a.hist_cap = 0;
    if (dm.t0 == 0) return launch_group(a.sp_mode ? MODE_FULL_SP : MODE_FULL, a, h->stream);
    int e = 0;
    if (h->flat_local) {
        e = launch_group(MODE_FLAT_A, a, h->stream);
        if (!e) e = launch_group(MODE_SUB_A, a, h->stream);
        if (!e) e = launch_group(MODE_CROWN, a, h->stream);
        if (!e) e = launch_group(MODE_SUB_B, a, h->stream);
        if (!e) e = launch_group(a.sp_mode ? MODE_FLAT_B_SP : MODE_FLAT_B, a, h->stream);
        return e;
    }
    e = launch_group(MODE_SUB_A, a, h->stream);
    if (!e) e = launch_group(MODE_CROWN, a, h->stream);
    if (!e) e = launch_group(a.sp_mode ? MODE_SUB_B_SP : MODE_SUB_B, a, h->stream);
    return e;
}

int launch_sp_iteration(spock_handle_s *h) {
    int e = launch_sp_T(h, h->s_vec[SV_Z], h->s_vec[SV_ZB], 1);
    if (h->sp_variant == 1) {
        // corrected variant (spock_sp.hpp: sp_ctl_a_ls, sp_ctl_trial): T(Z), FLAT1, CTL_A (normal equations), FLAT3, then per
        // line-search candidate j = 0 .. max_backtrack: [W = Z + beta^j D], T(W), decision; FLAT5, CTL_E.  The kernels of the
        // candidates behind the accepted one exit at once (done == 3).
        const int seq1[] = {SP_FLAT1, SP_CTL_A, SP_FLAT3};
        for (int op : seq1) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
        double tau = 1.;
        for (int j = 0; j <= h->sp_max_bt && !e; ++j, tau *= 0.5) {
            SpArgs a = make_sp_args(h, SP_FLATW);
            a.tau = tau; a.trial = j; a.last_trial = (j == h->sp_max_bt) ? 1 : 0;
            if (j > 0) e = launch_sp(a, h->stream);
            if (!e) e = launch_sp_T(h, h->s_vec[SV_W], h->s_vec[SV_WB], 2, h->s_vec[SV_D]);
            a.op = SP_CTL_CD;
            if (!e) e = launch_sp(a, h->stream);
        }
        const int seq2[] = {SP_FLAT5, SP_CTL_E};
        for (int op : seq2) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
        return e;
    }
    if (h->sp_fused) {   // every inner product and residual norm comes out of phase B of the two T evaluations: 13 launches
        const int seq1[] = {SP_FLAT1, SP_CTL_A, SP_FLAT2, SP_CTL_B, SP_FLAT3};
        for (int op : seq1) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
        if (!e) e = launch_sp_T(h, h->s_vec[SV_W], h->s_vec[SV_WB], 2, h->s_vec[SV_D]);
        const int seq2[] = {SP_CTL_CD, SP_FLAT5};
        for (int op : seq2) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
        return e;
    }
    const int seq1[] = {SP_FLAT1, SP_LOP_R, SP_CTL_A, SP_FLAT2, SP_CTL_B, SP_FLAT3};
    for (int op : seq1) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
    if (!e) e = launch_sp_T(h, h->s_vec[SV_W], h->s_vec[SV_WB]);
    const int seq2[] = {SP_FLAT4, SP_LOP_RT, SP_CTL_C, SP_FLAT5, SP_LOP_DP, SP_CTL_D};
    for (int op : seq2) if (!e) e = launch_sp(make_sp_args(h, op), h->stream);
    return e;
}

// initial values of build_spock (src/models/spock.jl:83-201): z = v = 0, z_old = v_old = rz_old = rv_old = 1,
// MP = Q = 0, aa_R = I, everything else 0
int sp_init_state(spock_handle_s *h) {
    const Dims &dm = h->bp.dm;
    const size_t B = (size_t)h->batch, nr = (size_t)dm.nz + dm.nv;
    for (int k = 0; k < 10; ++k) { if (!h->s_vec[k]) continue; int e = rt::dzero(h->s_vec[k], B * nr * sizeof(real), h->stream); if (e) return e; }
    std::vector<double> ones(B * nr, 1.0);
    int e = up(h, h->s_vec[SV_R], ones.data(), B * nr);   // rz_old = rv_old = 1 (models/spock.jl:159-201) in the buffer r shares with r_old; z_old itself is not stored (spock_sp.hpp: FLAT1, FLAT5)
    if (!e) e = rt::sync(h->stream);
    if (!e) e = rt::dzero(h->s_MP, B * 3 * nr * sizeof(real), h->stream);
    if (!e) e = rt::dzero(h->s_Q, B * 3 * nr * sizeof(real), h->stream);
    SpCtl c;
    std::memset(&c, 0, sizeof(c));
    c.R[0] = c.R[4] = c.R[8] = 1.;
    c.crho[0] = c.crho[1] = c.crho[2] = 1.;
    c.r_safe = INFINITY;
    h->h_sp.assign(B, c);
    if (!e) e = rt::h2d(h->s_ctl, h->h_sp.data(), B * sizeof(SpCtl), h->stream);
    if (!e) e = rt::sync(h->stream);
    return e;
}

int sp_solve(spock_handle_s *h, const double *x0, double tol, int verbose, const double *z0, const double *v0,
             double gamma, double sigma, spock_result_t *results) {
    const Dims &dm = h->bp.dm;
    const size_t B = (size_t)h->batch, nr = (size_t)dm.nz + dm.nv;
    if (x0) RTCHECK(up(h, h->d_x0, x0, B * dm.nx));   // NULL: x0 is already on the device (MPC loop)
    if (z0) {
        for (size_t b = 0; b < B; ++b) {
            RTCHECK(up(h, h->s_vec[SV_Z] + b * nr, z0 + b * dm.nz, dm.nz));
            RTCHECK(up(h, h->s_vec[SV_Z] + b * nr + dm.nz, v0 + b * dm.nv, dm.nv));
        }
    }
    h->h_ctl.assign(B, fresh_ctl(h, gamma, sigma, tol, verbose));
    RTCHECK(rt::h2d(h->d_ctl, h->h_ctl.data(), B * sizeof(Ctl), h->stream));
    // per-solve locals of run_sp! (sp.jl:383-389): Anderson counter and r_safe restart, R / head / buffers persist
    RTCHECK(rt::d2h(h->h_sp.data(), h->s_ctl, B * sizeof(SpCtl), h->stream));
    RTCHECK(rt::sync(h->stream));
    for (auto &c : h->h_sp) { c.k = 0; c.r_safe = INFINITY; c.rnorm = INFINITY; c.rtilde = INFINITY; c.accept = 0; c.backtrack = 0; }
    RTCHECK(rt::h2d(h->s_ctl, h->h_sp.data(), B * sizeof(SpCtl), h->stream));

    const int G = h->graph_iters > 0 ? h->graph_iters : -h->graph_iters;
    const bool use_graph = h->graph_iters > 0 && rt::graphs_supported();
    if (use_graph && !h->graph) {
        RTCHECK(rt::capture_begin(h->stream));
        int e = 0;
        for (int it = 0; it < G && e == 0; ++it) e = launch_sp_iteration(h);
        void *exec = nullptr;
        const int e2 = rt::capture_end(h->stream, &exec);
        if (e) return fail(SPOCK_ERR_HIP, std::string("kernel launch during capture: ") + rt::err(e));
        if (e2) return fail(SPOCK_ERR_HIP, std::string("hipStreamEndCapture/hipGraphInstantiate: ") + rt::err(e2));
        h->graph = exec;
    }
    RTCHECK(rt::event_record(h->ev0, h->stream));
    const int max_reps = (h->max_iter + 1 + G - 1) / G + 2;
    bool all_done = false;
    for (int rep = 0; rep < max_reps && !all_done; ++rep) {
        if (use_graph) RTCHECK(rt::graph_launch(h->graph, h->stream));
        else for (int it = 0; it < G; ++it) RTCHECK(launch_sp_iteration(h));
        RTCHECK(rt::d2h(h->h_ctl.data(), h->d_ctl, B * sizeof(Ctl), h->stream));
        RTCHECK(rt::sync(h->stream));
        all_done = true;
        for (const Ctl &c : h->h_ctl) all_done = all_done && c.done;
    }
    RTCHECK(rt::event_record(h->ev1, h->stream));
    RTCHECK(rt::sync(h->stream));
    float ms = 0.f;
    RTCHECK(rt::event_elapsed(&ms, h->ev0, h->ev1));
    if (!all_done) return fail(SPOCK_ERR_HIP, "driver did not terminate (internal error)");
    int evals = 0;
    for (size_t b = 0; b < B; ++b) {
        const Ctl &c = h->h_ctl[b];
        if (c.evals > evals) evals = c.evals;
        if (results) {
            spock_result_t &r = results[b];
            r.status = c.status; r.iterations = c.evals; r.iter_ref = c.iter; r.reserved = 0;
            r.xi1 = c.xi1; r.xi2 = c.xi2; r.xi = c.xi; r.solve_ms = ms;
        }
    }
    h->last_ms = ms; h->last_iters = evals;
    return SPOCK_OK;
}

// Closed-loop simulations with the steps chained ON THE DEVICE (SP, the reference's algorithm as it stands): every instance runs its own
// sequence solve -> apply u -> advance x0 -> next warm-started solve; the control kernel SP_CTL_M at the end of an iteration does for a
// finished instance what mpc_simulation.jl:47-54,164-177 does between two solve_model! calls.  No host decision per step and no instance
// waits for the slowest one of its batch: the host replays a graph of G iterations and polls the control blocks until every instance
// has done its last step.  (The host loop -- one solve call per step, all instances in lockstep -- remains for CP and SPOCK_AMD_MPC_HOST=1.)
int sp_mpc_chain(spock_handle_s *h, const double *x0, int steps, const int32_t *w_seq, double tol, int verbose,
                 double *x_traj, double *u_traj, spock_result_t *results) {
    const Dims &dm = h->bp.dm;
    const size_t B = (size_t)h->batch;
    double gamma = 0.99 / std::sqrt(h->L_norm), sigma = gamma;                     // cp.jl:198-201
    if (steps > h->m_cap_steps) {   // (grown buffers stay with the handle; the old ones are released with it)
        int e = dev_alloc(h, &h->m_w, B * steps);
        if (!e) e = dev_alloc(h, &h->m_t, B);
        if (!e) e = dev_alloc(h, &h->m_xt, B * (size_t)(steps + 1) * dm.nx);
        if (!e) e = dev_alloc(h, &h->m_ut, B * (size_t)steps * dm.nu);
        if (!e) e = dev_alloc(h, &h->m_res, B * (size_t)steps * kMpcRec);
        if (e) return fail(SPOCK_ERR_HIP, std::string("hipMalloc(trajectories): ") + rt::err(e));
        h->m_cap_steps = steps;
        if (h->graph_mpc) { rt::graph_destroy(h->graph_mpc); h->graph_mpc = nullptr; }
    }
    if (h->graph_mpc && h->m_graph_steps != steps) { rt::graph_destroy(h->graph_mpc); h->graph_mpc = nullptr; }
    h->m_graph_steps = steps;
    RTCHECK(rt::h2d(h->m_w, w_seq, B * steps * sizeof(int), h->stream));
    RTCHECK(rt::dzero(h->m_t, B * sizeof(int), h->stream));
    RTCHECK(up(h, h->d_x0, x0, B * dm.nx));
    h->h_ctl.assign(B, fresh_ctl(h, gamma, sigma, tol, verbose));
    RTCHECK(rt::h2d(h->d_ctl, h->h_ctl.data(), B * sizeof(Ctl), h->stream));
    RTCHECK(rt::d2h(h->h_sp.data(), h->s_ctl, B * sizeof(SpCtl), h->stream));
    RTCHECK(rt::sync(h->stream));
    for (auto &c : h->h_sp) { c.k = 0; c.r_safe = INFINITY; c.rnorm = INFINITY; c.rtilde = INFINITY; c.accept = 0; c.backtrack = 0; }
    RTCHECK(rt::h2d(h->s_ctl, h->h_sp.data(), B * sizeof(SpCtl), h->stream));

    int G = 16;
    if (const char *e = std::getenv("SPOCK_AMD_MPC_GRAPH_ITERS")) { G = std::atoi(e); if (G < 1) G = 1; }
    auto launch_G = [&]() {
        int e = 0;
        for (int it = 0; it < G && e == 0; ++it) {
            e = launch_sp_iteration(h);
            if (!e) e = launch_sp(make_sp_args(h, SP_CTL_M), h->stream);
        }
        return e;
    };
    const bool use_graph = h->graph_iters > 0 && rt::graphs_supported();
    if (use_graph && !h->graph_mpc) {
        RTCHECK(rt::capture_begin(h->stream));
        const int e = launch_G();
        void *exec = nullptr;
        const int e2 = rt::capture_end(h->stream, &exec);
        if (e) return fail(SPOCK_ERR_HIP, std::string("kernel launch during capture: ") + rt::err(e));
        if (e2) return fail(SPOCK_ERR_HIP, std::string("hipStreamEndCapture/hipGraphInstantiate: ") + rt::err(e2));
        h->graph_mpc = exec;
    }
    RTCHECK(rt::event_record(h->ev0, h->stream));
    const long max_reps = ((long)steps * (h->max_iter + 2) + G - 1) / G + 2;
    bool all_done = false;
    for (long rep = 0; rep < max_reps && !all_done; ++rep) {
        if (use_graph) RTCHECK(rt::graph_launch(h->graph_mpc, h->stream));
        else RTCHECK(launch_G());
        RTCHECK(rt::d2h(h->h_ctl.data(), h->d_ctl, B * sizeof(Ctl), h->stream));
        RTCHECK(rt::sync(h->stream));
        all_done = true;
        for (const Ctl &c : h->h_ctl) all_done = all_done && (c.done == 1);
    }
    RTCHECK(rt::event_record(h->ev1, h->stream));
    RTCHECK(rt::sync(h->stream));
    float ms = 0.f;
    RTCHECK(rt::event_elapsed(&ms, h->ev0, h->ev1));
    if (!all_done) return fail(SPOCK_ERR_HIP, "closed loop did not terminate (internal error)");
    std::vector<double> rec(B * (size_t)steps * kMpcRec);
    RTCHECK(rt::d2h(rec.data(), h->m_res, rec.size() * sizeof(double), h->stream));
    if (x_traj) RTCHECK(down(h, x_traj, h->m_xt, B * (size_t)(steps + 1) * dm.nx));
    if (u_traj) RTCHECK(down(h, u_traj, h->m_ut, B * (size_t)steps * dm.nu));
    RTCHECK(rt::sync(h->stream));
    int evals = 0;
    for (size_t k = 0; k < B * (size_t)steps; ++k) {
        const double *r = rec.data() + k * kMpcRec;
        evals += (int)r[1];
        if (results) {
            spock_result_t &o = results[k];
            o.status = (int)r[0]; o.iterations = (int)r[1]; o.iter_ref = (int)r[2]; o.reserved = 0;
            o.xi1 = r[3]; o.xi2 = r[4]; o.xi = r[5];
            o.solve_ms = ms / steps;   // the steps of different instances overlap: the whole call's device time, per step
        }
    }
    h->last_ms = ms; h->last_iters = evals;
    return SPOCK_OK;
}

}  // namespace

extern "C" {

const char *spock_last_error(void) { return g_err.c_str(); }
int spock_version(void) { return SPOCK_AMD_VERSION; }
int spock_real_size(void) { return (int)sizeof(real); }
int spock_abi_struct_sizes(int32_t *out, int32_t cap) {
    if (!out || cap < 1) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    const int32_t sz[4] = {(int32_t)sizeof(spock_problem_t), (int32_t)sizeof(spock_options_t), (int32_t)sizeof(spock_result_t), (int32_t)sizeof(spock_sizes_t)};
    for (int k = 0; k < 4 && k < cap; ++k) out[k] = sz[k];
    return SPOCK_OK;
}

int spock_create(const spock_problem_t *p, const spock_options_t *o, spock_handle_t *out) {
    if (!p || !out) return fail(SPOCK_ERR_BAD_ARG, "problem and out must not be NULL");
    *out = nullptr;
    if (!p->A || !p->B || !p->Q || !p->R || !p->QN || !p->E || !p->F || !p->b || (p->n_cones > 0 && (!p->cone_kind || !p->cone_dim)))
        return fail(SPOCK_ERR_BAD_ARG, "NULL array in spock_problem_t");
    if (p->d <= 1) return fail(SPOCK_ERR_BAD_ARG, "Branching factor d must be larger than 1");
    if (p->N < 2 || p->N > kMaxStages || p->nx < 1 || p->nu < 1 || p->ny < 1) return fail(SPOCK_ERR_BAD_ARG, "invalid N / nx / nu / ny");
    const int alg = o && o->algorithm ? o->algorithm : SPOCK_ALG_CP;
    if (alg != SPOCK_ALG_CP && alg != SPOCK_ALG_SP) return fail(SPOCK_ERR_UNSUPPORTED, "This combination of solver options is not supported.");  // model.jl:295

    spock_handle_s *h = new spock_handle_s();
    HostProblem &hp = h->hp;
    hp.N = p->N; hp.d = p->d; hp.nx = p->nx; hp.nu = p->nu; hp.ny = p->ny;
    const int N = hp.N, d = hp.d, nx = hp.nx, nu = hp.nu, ny = hp.ny;
    long n = 0; { long c = 1; for (int t = 0; t < N; ++t) { n += c; c *= d; } }
    long nl = 1; for (int t = 0; t < N - 1; ++t) nl *= d;
    // the solver assumes the uniform breadth-first layout (children contiguous, implicit_l.jl:712-714)
    if (p->anc || p->w) {
        for (long j = 0; j < n; ++j) {
            const int ea = j == 0 ? -1 : (int)((j - 1) / d), ew = j == 0 ? -1 : (int)((j + 1) % d);
            if ((p->anc && p->anc[j] != ea) || (p->w && p->w[j] != ew)) {
                delete h;
                return fail(SPOCK_ERR_UNSUPPORTED, "scenario tree is not the uniform breadth-first tree of generate_scenario_tree_uniform_branching_factor_v2");
            }
        }
    }
    hp.nQ = p->nQ; hp.nR = p->nR; hp.nQN = p->nQN;
    if (!(hp.nQ == 1 || hp.nQ == n - 1) || !(hp.nR == 1 || hp.nR == n - 1) || !(hp.nQN == 1 || hp.nQN == nl)) {
        delete h;
        return fail(SPOCK_ERR_BAD_ARG, "nQ, nR, nQN must be 1 or n-1 / n-1 / n_leaf");
    }
    hp.A.assign(p->A, p->A + (size_t)d * nx * nx);
    hp.B.assign(p->B, p->B + (size_t)d * nx * nu);
    hp.Q.assign(p->Q, p->Q + (size_t)hp.nQ * nx * nx);
    hp.R.assign(p->R, p->R + (size_t)hp.nR * nu * nu);
    hp.QN.assign(p->QN, p->QN + (size_t)hp.nQN * nx * nx);
    {
        const long nn_ = (nl - 1) / (d - 1);
        hp.n_rms = p->n_rms > 1 ? p->n_rms : 1;
        if (hp.n_rms > 1 && hp.n_rms != nn_) { delete h; return fail(SPOCK_ERR_BAD_ARG, "n_rms must be 0 / 1 or the number of non-leaf nodes"); }
    }
    hp.E.assign(p->E, p->E + (size_t)hp.n_rms * ny * d);
    hp.F.assign(p->F, p->F + (size_t)hp.n_rms * ny * d);
    hp.b.assign(p->b, p->b + (size_t)hp.n_rms * ny);
    hp.cone_kind.assign(p->cone_kind, p->cone_kind + p->n_cones);
    hp.cone_dim.assign(p->cone_dim, p->cone_dim + p->n_cones);
    hp.x_min = p->x_min; hp.x_max = p->x_max; hp.u_min = p->u_min; hp.u_max = p->u_max;

    h->alg = alg;
    h->device = o ? o->device : 0;
    h->batch = o && o->batch > 0 ? o->batch : 1;
    h->L_norm = (o && std::isfinite(o->L_norm) && o->L_norm > 0.) ? o->L_norm : (alg == SPOCK_ALG_CP ? 10. : 3.28);  // cpock.jl:63, models/spock.jl:64
    h->max_iter = o && o->max_iter > 0 ? o->max_iter : (alg == SPOCK_ALG_CP ? 5000 : 1000);                          // cp.jl:190, sp.jl:360
    h->sp_variant = (o && alg == SPOCK_ALG_SP) ? o->sp_variant : 0;
    if (h->sp_variant < 0 || h->sp_variant > 1) { delete h; return fail(SPOCK_ERR_BAD_ARG, "sp_variant must be 0 (as the reference) or 1 (corrected)"); }
    h->sp_max_bt = (o && o->sp_max_backtrack > 0) ? o->sp_max_backtrack : 8;                                        // sp.jl:371
    if (h->sp_max_bt > 16) { delete h; return fail(SPOCK_ERR_BAD_ARG, "sp_max_backtrack must be at most 16"); }
    int gi = o ? o->graph_iters : 0;
    if (const char *e = std::getenv("SPOCK_AMD_GRAPH_ITERS")) gi = std::atoi(e);
    if (gi == 0) gi = (alg == SPOCK_ALG_SP) ? (h->sp_variant == 1 ? 2 : 8) : 32;
    {   // negative = eager launches of |gi| iterations per poll.  Either way every repetition starts at buffer slot 0, so
        // |gi| is a multiple of 6 for CP (z rotates through three buffers, v through two) and even for SP
        const int sgn = gi < 0 ? -1 : 1;
        int mag = gi < 0 ? -gi : gi;
        if (alg == SPOCK_ALG_CP) mag = ((mag + 5) / 6) * 6;
        else if (mag & 1) mag += 1;
        gi = sgn * mag;
    }
    h->graph_iters = gi;
    int split = o ? o->split_stage : 0;
    if (const char *e = std::getenv("SPOCK_AMD_SPLIT_STAGE")) split = std::atoi(e);

    // Layout of the batch: element-major (one lane per instance, spock_igroup.hpp / spock_ilevel.hpp) needs tiles of 64 instances and
    // enough of them to fill the chip.  Measured on MI355X (config 2, iterations/s instance-major -> element-major): 64 instances
    // 346 k -> 181 k, 128: 401 k -> 307 k, 256: 464 k -> 475 k, 384: 472 k -> 541 k, 512: 495 k -> 573 k, 1024: 519 k -> 722 k,
    // 2048: 527 k -> 800 k.  Chosen from 320 instances (five tiles) on, for CP and the compile-time shapes; batch_layout /
    // SPOCK_AMD_ILV override.
    int layout = o ? o->batch_layout : 0;
    if (const char *e = std::getenv("SPOCK_AMD_ILV")) layout = (e[0] == '1') ? 2 : 1;
    if (layout < 0 || layout > 2) { delete h; return fail(SPOCK_ERR_BAD_ARG, "batch_layout must be 0 (choose), 1 (instance-major) or 2 (element-major)"); }
    const bool per_inst = o && o->per_instance_dynamics != 0;
    if (per_inst && layout == 2) { delete h; return fail(SPOCK_ERR_UNSUPPORTED, "per-instance dynamics need the instance-major layout (in the element-major kernels a matrix element is one scalar for 64 instances)"); }
    const bool ilv_possible = !per_inst && alg == SPOCK_ALG_CP && N >= 3 && have_static_variant(nx, nu, d, ny) && hp.nQ == 1 && hp.nR == 1 && hp.nQN == 1 && hp.n_rms == 1;
    if (layout == 2 && !ilv_possible) { delete h; return fail(SPOCK_ERR_UNSUPPORTED, "the element-major batch layout exists for CP, the compile-time shapes and shared cost matrices"); }
    bool want_ilv = ilv_possible && (layout == 2 || (layout == 0 && h->batch >= 320));
    if (!ilv_possible && layout != 1 && h->batch >= 320) h->layout_reason = 1;

    bool unsupported = false;
    std::string msg;
    if (want_ilv) {
        const int isplit = split > 0 ? split : choose_split_stage_ilv(N, d, nx);
        // (buffer addressing: 32-bit byte offsets inside an array)
        long nvv = 0; { long nn_ = (nl - 1) / (d - 1); nvv = nn_ * ny + nn_ + (long)(nx + nu + 2) * (n - 1) + nn_ * (nx + nu) + nl * (2L * nx + 2); }
        const bool small = nvv * (((long)(h->batch + 63) / 64) * 64) * 8 < (1L << 32) - 4096;
        // (the bottom-group kernels keep a subtree's sweep slots in LDS; the level-streaming kernels have no such limit)
        const bool stream_lv = !(std::getenv("SPOCK_AMD_ILV_STREAM") && std::getenv("SPOCK_AMD_ILV_STREAM")[0] == '0');
        const bool fits = small && isplit >= 1 && isplit <= N - 2 && (stream_lv || (long)group_threads(d, N - isplit) * nx * 64 * 8 <= 56 * 1024);
        msg = fits ? host_build(hp, isplit, h->bp, unsupported, h->batch) : std::string("subtree does not fit the element-major kernels' LDS slots");
        if (!msg.empty() && layout == 2) { delete h; return fail(SPOCK_ERR_UNSUPPORTED, msg); }
        if (!msg.empty()) { want_ilv = false; h->layout_reason = small ? 3 : 2; }   // chosen automatically: fall back to the instance-major kernels, and say so
    }
    if (!want_ilv) msg = host_build(hp, split, h->bp, unsupported, h->batch);
    if (!msg.empty()) { delete h; return fail(unsupported ? SPOCK_ERR_UNSUPPORTED : SPOCK_ERR_BUILD, msg); }
    h->ilv = want_ilv;
    h->bpad = want_ilv ? ((long)(h->batch + 63) / 64) * 64 : 0;
    if (o && o->L_norm < 0.) h->L_norm = h->bp.L_norm_closed * (1. + 1e-9);   // negative: closed form instead of the hard-coded constant
    const Dims &dm = h->bp.dm;

    auto bail = [&](int e, const char *what) {
        std::string m = std::string(what) + ": " + rt::err(e);
        spock_destroy(h);
        return fail(SPOCK_ERR_HIP, m);
    };
    DeviceGuard guard(h->device);
    int e = 0;
    { int cur = -1; if ((e = rt::get_device(&cur)) || cur != h->device) return bail(e ? e : 101 /* hipErrorInvalidDevice */, "hipSetDevice"); }
    if ((e = rt::stream_create(&h->stream))) return bail(e, "hipStreamCreate");
    if ((e = rt::event_create(&h->ev0)) || (e = rt::event_create(&h->ev1))) return bail(e, "hipEventCreate");
    const size_t B = (size_t)h->batch;
    // one matrix pool for the batch, or -- per_instance_dynamics -- one per instance (all equal until spock_update_dynamics_batch)
    const size_t pool_n = h->bp.mats.size();
    h->mats_stride = (o && o->per_instance_dynamics != 0) ? (long)pool_n : 0;
    if ((e = dev_alloc(h, &h->d_mats, pool_n * (h->mats_stride ? B : 1)))) return bail(e, "hipMalloc(matrix pool)");
    for (size_t b = 0; b < (h->mats_stride ? B : 1); ++b)
        if ((e = up(h, h->d_mats + b * pool_n, h->bp.mats.data(), pool_n))) return bail(e, "upload");
    const size_t W = h->ilv ? (size_t)h->bpad : B;   // columns of the per-instance work arrays
    if (alg == SPOCK_ALG_CP && h->ilv) {
        if ((e = dev_alloc(h, &h->d_z[0], B * dm.nz)) || (e = dev_alloc(h, &h->d_v[0], B * dm.nv))) return bail(e, "hipMalloc(staging)");
        for (int k = 0; k < 3; ++k) if ((e = dev_alloc(h, &h->i_z[k], W * dm.nz))) return bail(e, "hipMalloc(state)");
        for (int k = 0; k < 2; ++k) if ((e = dev_alloc(h, &h->i_v[k], W * dm.nv))) return bail(e, "hipMalloc(state)");
        if ((e = dev_alloc(h, &h->i_cq, W * (size_t)dm.nn * dm.nx))) return bail(e, "hipMalloc(state)");   // q of every non-leaf node
        h->stage_valid = true;
        {   // tile groups = parallel branches of the captured graph, each with its own chain of kernels over a share of the tiles.
            // Bottom-group kernels: at least four tiles per group, at most four groups (one group's latency-bound crown chain
            // overlaps the others' bottom kernel).  Level-streaming kernels: ONE group -- their launches want all the tiles, and
            // measured on MI355X four branches help only under some HIP runtimes (ROCm 7.2: 2.52 -> 2.30 ms per iteration on one
            // box, no change on another; the 7.0 runtime a PyTorch process brings along: 2.68 -> 3.1 ms)
            const int tiles = (int)(h->bpad / 64);
            h->n_tg = h->bp.dm.ilv_stream ? 1 : (tiles >= 16 ? 4 : (tiles >= 8 ? 2 : 1));
            if (const char *e2 = std::getenv("SPOCK_AMD_ILV_TGROUPS")) { h->n_tg = std::atoi(e2); if (h->n_tg < 1) h->n_tg = 1; if (h->n_tg > tiles) h->n_tg = tiles; if (h->n_tg > 8) h->n_tg = 8; }
            if (h->n_tg > 1) {
                if ((e = rt::event_create(&h->ev_fork))) return bail(e, "hipEventCreate");
                h->tg_stream.assign(h->n_tg, nullptr); h->tg_event.assign(h->n_tg, nullptr);
                for (int gi = 0; gi < h->n_tg; ++gi)
                    if ((e = rt::stream_create(&h->tg_stream[gi])) || (e = rt::event_create(&h->tg_event[gi]))) return bail(e, "hipStreamCreate");
            }
        }
    } else if (alg == SPOCK_ALG_CP) {
        for (int k = 0; k < 3; ++k) if ((e = dev_alloc(h, &h->d_z[k], B * dm.nz))) return bail(e, "hipMalloc(state)");
        for (int k = 0; k < 2; ++k) if ((e = dev_alloc(h, &h->d_v[k], B * dm.nv))) return bail(e, "hipMalloc(state)");
        if ((e = dev_alloc(h, &h->d_v2x, B * (size_t)dm.nsub))) return bail(e, "hipMalloc(state)");
            h->lat_regime = dm.t0 > 0 && launch_geometry(MODE_SUB_B, dm, h->batch).team_reg != 0;
    } else {
        const size_t nr = (size_t)dm.nz + dm.nv;
        for (int k = 0; k < 10; ++k) {
            if (k == SV_ZOLD || k == SV_ROLD || (k == SV_RW && !std::getenv("SPOCK_AMD_SP_LOP"))) continue;   // not stored (FLAT1, FLAT5) / only the separate-LOP sequence forms rt = w - wbar
            if ((e = dev_alloc(h, &h->s_vec[k], B * nr))) return bail(e, "hipMalloc(SP state)");
        }
        if ((e = dev_alloc(h, &h->s_MP, B * 3 * nr)) || (e = dev_alloc(h, &h->s_Q, B * 3 * nr)) ||
            (e = dev_alloc(h, &h->s_partF, B * (size_t)sp_flat_blocks((long)nr) * kSpSums)) ||
            (e = dev_alloc(h, &h->s_partL, B * (size_t)sp_lop_blocks(dm) * kSpSums)) || (e = dev_alloc(h, &h->s_ctl, B)))
            return bail(e, "hipMalloc(SP state)");
    }
    h->flat_local = !h->ilv && flat_local_applies(dm, h->batch);
    h->flat_chunks = h->flat_local ? flat_geometry(dm).chunks : 0;
    h->hist_cap = h->max_iter + 2;
    const size_t n_groups_part = (size_t)((h->flat_chunks > dm.nsub + 1) ? h->flat_chunks : dm.nsub + 1);
    const size_t n_part = h->ilv ? (size_t)(h->bpad / 64) * (size_t)(dm.n_bslots + dm.n_crown_groups) * 3 * 64
                                 : B * n_groups_part * kPartialStride;
    if ((e = dev_alloc(h, &h->d_dbuf, W * dm.nn * dm.nu)) || (e = dev_alloc(h, &h->d_qroot, W * dm.nsub * dm.nx)) ||
        (e = dev_alloc(h, &h->d_cbuf, W * (size_t)(group_threads(dm.d, dm.t0 + 1) + 1) * (dm.nx + dm.nu))) ||
        (e = dev_alloc(h, &h->d_partials, n_part)) || (e = dev_alloc(h, &h->d_x0, B * dm.nx)) ||
        (e = dev_alloc(h, &h->d_hist, B * (size_t)h->hist_cap * kHistStride)) || (e = dev_alloc(h, &h->d_ctl, B)) ||
        (e = dev_alloc(h, &h->d_dbg, (size_t)3 * 64 * 2)))
        return bail(e, "hipMalloc(workspace)");
    {   // group-leaf sweep vector in global memory, when the lane-team vectors do not all fit in LDS
        const LaunchGeo g1 = launch_geometry(dm.t0 > 0 ? MODE_SUB_A : MODE_FULL, dm, h->batch), g2 = launch_geometry(MODE_CROWN, dm, h->batch);
        if ((g1.team && g1.xl_global) || (dm.t0 > 0 && g2.team && g2.xl_global)) {
            const size_t stride = (size_t)(g1.xl_stride > g2.xl_stride ? g1.xl_stride : g2.xl_stride);
            if ((e = dev_alloc(h, &h->d_xlbuf, B * (size_t)(dm.nsub + 1) * stride))) return bail(e, "hipMalloc(workspace)");
        }
    }
    h->h_ctl.assign(B, fresh_ctl(h, 0., 0., 0., SPOCK_SILENT));
    for (auto &c : h->h_ctl) c.done = 1;
    if ((e = rt::h2d(h->d_ctl, h->h_ctl.data(), B * sizeof(Ctl), h->stream))) return bail(e, "upload");
    if (alg == SPOCK_ALG_CP && (e = choose_cp_sequence(h))) return bail(e, "timing the CP kernel sequences");
    h->rlane = alg == SPOCK_ALG_CP && !h->ilv && !h->flat_local && h->mats_stride == 0 && rlane_applies(dm, h->batch, h->fused);
    if (alg == SPOCK_ALG_SP) h->sp_fused = !std::getenv("SPOCK_AMD_SP_LOP");
    if (alg == SPOCK_ALG_SP && (e = sp_init_state(h))) return bail(e, "SP state initialisation");
    if ((e = rt::sync(h->stream))) return bail(e, "hipStreamSynchronize");
    *out = h;
    return SPOCK_OK;
}

int spock_destroy(spock_handle_t h) {
    if (!h) return SPOCK_OK;
    DeviceGuard guard(h->device);
    if (h->stream) rt::sync(h->stream);
    if (h->graph) rt::graph_destroy(h->graph);
    if (h->graph_log) rt::graph_destroy(h->graph_log);
    if (h->graph_mpc) rt::graph_destroy(h->graph_mpc);
    for (void *st : h->tg_stream) if (st) { rt::sync(st); rt::stream_destroy(st); }
    for (void *ev : h->tg_event) if (ev) rt::event_destroy(ev);
    if (h->ev_fork) rt::event_destroy(h->ev_fork);
    for (void *p : h->allocs) rt::dfree(p);
    if (h->ev0) rt::event_destroy(h->ev0);
    if (h->ev1) rt::event_destroy(h->ev1);
    if (h->stream) rt::stream_destroy(h->stream);
    delete h;
    return SPOCK_OK;
}

int spock_sizes(spock_handle_t h, spock_sizes_t *s) {
    if (!h || !s) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    const Dims &dm = h->bp.dm;
    s->n = dm.n; s->n_nonleaf = dm.nn; s->n_leaf = dm.nl; s->nz = dm.nz; s->nv = dm.nv; s->batch = h->batch;
    s->z_x = dm.zx; s->z_u = dm.zu; s->z_s = dm.zs; s->z_tau = dm.zt; s->z_y = dm.zy;
    s->v1 = 0; s->v2 = dm.o2; s->v3 = dm.o3; s->v4 = dm.o4; s->v5 = dm.o5; s->v6 = dm.o6; s->v7 = dm.o7;
    s->v11 = dm.o11; s->v12 = dm.o12; s->v13 = dm.o13; s->v14 = dm.o14;
    s->split_stage = dm.t0; s->n_subtrees = dm.nsub; s->graph_iters = h->graph_iters;
    s->fused_iteration = h->fused ? 1 : 0;
    s->interleaved = h->ilv ? 1 : 0;
    s->layout_reason = h->layout_reason;
    return SPOCK_OK;
}

static int solve_impl(spock_handle_t h, const double *x0, double tol, int32_t verbose, const double *z0, const double *v0,
                      double gamma, double sigma, spock_result_t *results);

int spock_solve(spock_handle_t h, const double *x0, double tol, int32_t verbose, const double *z0, const double *v0,
                double gamma, double sigma, spock_result_t *results) {
    if (!h || !x0) return fail(SPOCK_ERR_BAD_ARG, "handle and x0 must not be NULL");
    DeviceGuard guard(h->device);
    return solve_impl(h, x0, tol, verbose, z0, v0, gamma, sigma, results);
}

// Closed-loop MPC simulation of examples/server_heat/mpc_simulation.jl:47-54,164-177 (Spock branch): `steps` warm-started
// solves; after each one the first-stage input u = z[u_inds[1:nu]] is applied and the state advanced on the DEVICE,
// x0 <- A_w x0 + B_w u with the event w = w_seq[b][t] supplied by the caller (the reference draws rand(1:d)).
int spock_mpc_simulate(spock_handle_t h, const double *x0, int32_t steps, const int32_t *w_seq, double tol, int32_t verbose,
                       double *x_traj, double *u_traj, spock_result_t *results) {
    if (!h || !x0 || !w_seq || steps < 1) return fail(SPOCK_ERR_BAD_ARG, "handle, x0, w_seq must not be NULL and steps >= 1");
    DeviceGuard guard(h->device);
    const Dims &dm = h->bp.dm;
    const size_t B = (size_t)h->batch;
    for (size_t k = 0; k < B * (size_t)steps; ++k)
        if (w_seq[k] < 0 || w_seq[k] >= dm.d) return fail(SPOCK_ERR_BAD_ARG, "w_seq entries must be events in [0, d)");
    if (verbose != SPOCK_SILENT && verbose != SPOCK_LOG) return fail(SPOCK_ERR_BAD_ARG, "verbose must be SPOCK_SILENT or SPOCK_LOG");
    if (!(tol >= 0.)) return fail(SPOCK_ERR_BAD_ARG, "tol must be non-negative");
    {
        const char *e = std::getenv("SPOCK_AMD_MPC_HOST");
        if (h->alg == SPOCK_ALG_SP && h->sp_variant == 0 && h->sp_fused && !(e && e[0] == '1'))
            return sp_mpc_chain(h, x0, steps, w_seq, tol, verbose, x_traj, u_traj, results);
    }
    int *d_w = nullptr;
    real *d_xt = nullptr, *d_ut = nullptr;
    void *pw = nullptr, *pxt = nullptr, *put = nullptr;
    auto cleanup = [&]() { if (pw) rt::dfree(pw); if (pxt) rt::dfree(pxt); if (put) rt::dfree(put); };
    {
        int e = rt::dmalloc(&pw, B * steps * sizeof(int));
        if (!e) e = rt::dmalloc(&pxt, B * (steps + 1) * dm.nx * sizeof(double));
        if (!e) e = rt::dmalloc(&put, B * steps * dm.nu * sizeof(double));
        if (e) { cleanup(); return fail(SPOCK_ERR_HIP, std::string("hipMalloc(trajectories): ") + rt::err(e)); }
    }
    d_w = (int *)pw; d_xt = (real *)pxt; d_ut = (real *)put;
    int rc = SPOCK_OK;
    do {
        int e = rt::h2d(d_w, w_seq, B * steps * sizeof(int), h->stream);
        if (!e) e = up(h, h->d_x0, x0, B * dm.nx);
        if (e) { rc = fail(SPOCK_ERR_HIP, rt::err(e)); break; }
        std::vector<spock_result_t> step_res(B);
        for (int t = 0; t < steps && rc == SPOCK_OK; ++t) {
            rc = solve_impl(h, nullptr, tol, verbose, nullptr, nullptr, NAN, NAN, step_res.data());
            if (rc) break;
            if (results) for (size_t b = 0; b < B; ++b) results[b * steps + t] = step_res[b];
            MpcArgs a;
            std::memset(&a, 0, sizeof(a));
            a.dm = dm; a.mats = h->d_mats; a.mats_stride = h->mats_stride; a.x0 = h->d_x0; a.w_seq = d_w; a.x_traj = d_xt; a.u_traj = d_ut;
            a.t = t; a.steps = steps; a.batch = h->batch;
            a.stride_e = 1;
            if (h->alg == SPOCK_ALG_SP) { a.z = h->s_vec[0]; a.stride_z = (long)dm.nz + dm.nv; }
            else if (h->ilv) { a.z = h->i_z[0]; a.stride_z = 1; a.stride_e = h->bpad; }
            else { a.z = h->d_z[0]; a.stride_z = dm.nz; }
            e = launch_mpc_advance(a, h->stream);
            if (e) rc = fail(SPOCK_ERR_HIP, rt::err(e));
        }
        if (rc) break;
        int e2 = 0;
        if (x_traj) e2 = down(h, x_traj, d_xt, B * (steps + 1) * dm.nx);
        if (!e2 && u_traj) e2 = down(h, u_traj, d_ut, B * steps * dm.nu);
        if (!e2) e2 = rt::sync(h->stream);
        if (e2) rc = fail(SPOCK_ERR_HIP, rt::err(e2));
    } while (0);
    cleanup();
    return rc;
}

static int solve_impl(spock_handle_t h, const double *x0, double tol, int32_t verbose, const double *z0, const double *v0,
                      double gamma, double sigma, spock_result_t *results) {
    if ((z0 == nullptr) != (v0 == nullptr)) return fail(SPOCK_ERR_BAD_ARG, "z0 and v0 must be given together");  // cpock.jl:180-183
    if (verbose != SPOCK_SILENT && verbose != SPOCK_LOG) return fail(SPOCK_ERR_BAD_ARG, "verbose must be SPOCK_SILENT or SPOCK_LOG");
    if (!(tol >= 0.)) return fail(SPOCK_ERR_BAD_ARG, "tol must be non-negative");
    const Dims &dm = h->bp.dm;
    const size_t B = (size_t)h->batch;
    if (!(std::isfinite(gamma) && std::isfinite(sigma))) { sigma = 0.99 / std::sqrt(h->L_norm); gamma = sigma; }  // cp.jl:198-201
    if (!(gamma > 0. && sigma > 0.)) return fail(SPOCK_ERR_BAD_ARG, "gamma and sigma must be positive");

    if (h->alg == SPOCK_ALG_SP) return sp_solve(h, x0, tol, verbose, z0, v0, gamma, sigma, results);
    if (x0) RTCHECK(up(h, h->d_x0, x0, B * dm.nx));   // NULL: x0 is already on the device
    if (z0) {
        RTCHECK(up(h, h->d_z[0], z0, B * dm.nz));
        RTCHECK(up(h, h->d_v[0], v0, B * dm.nv));
        if (h->ilv) RTCHECK(ilv_stage_to_state(h));
    }
    h->stage_valid = false;
    h->h_ctl.assign(B, fresh_ctl(h, gamma, sigma, tol, verbose));
    RTCHECK(rt::h2d(h->d_ctl, h->h_ctl.data(), B * sizeof(Ctl), h->stream));

    const int G = h->graph_iters > 0 ? h->graph_iters : -h->graph_iters;   // negative: eager launches
    const bool use_graph = h->graph_iters > 0 && rt::graphs_supported();
    h->want_xi = ((h->ilv || h->rlane) && verbose == SPOCK_LOG) ? 1 : 0;   // (kernel variants with the combined residual: another graph)
    void *&graph = h->want_xi ? h->graph_log : h->graph;
    if (use_graph && !graph) {
        RTCHECK(rt::capture_begin(h->stream));
        int e = 0;
        e = launch_cp_iterations(h, G);
        void *exec = nullptr;
        const int e2 = rt::capture_end(h->stream, &exec);
        if (e) return fail(SPOCK_ERR_HIP, std::string("kernel launch during capture: ") + rt::err(e));
        if (e2) return fail(SPOCK_ERR_HIP, std::string("hipStreamEndCapture/hipGraphInstantiate: ") + rt::err(e2));
        graph = exec;
    }
    RTCHECK(rt::event_record(h->ev0, h->stream));
    RTCHECK(cp_prologue(h));
    const int max_reps = (h->max_iter + 1 + G - 1) / G + 2;
    bool all_done = false;
    for (int rep = 0; rep < max_reps && !all_done; ++rep) {
        if (use_graph) RTCHECK(rt::graph_launch(graph, h->stream));
        else RTCHECK(launch_cp_iterations(h, G));
        RTCHECK(launch_group(MODE_FINALIZE, make_args(h, false, 0, 0), h->stream));
        RTCHECK(rt::d2h(h->h_ctl.data(), h->d_ctl, B * sizeof(Ctl), h->stream));
        RTCHECK(rt::sync(h->stream));
        all_done = true;
        for (const Ctl &c : h->h_ctl) all_done = all_done && c.done;
    }
    RTCHECK(rt::event_record(h->ev1, h->stream));
    RTCHECK(rt::sync(h->stream));
    float ms = 0.f;
    RTCHECK(rt::event_elapsed(&ms, h->ev0, h->ev1));
    if (!all_done) return fail(SPOCK_ERR_HIP, "driver did not terminate (internal error)");
    // the persistent state of every instance lives in buffer 0 between solves
    int evals = 0;
    bool same_parity = true;   // every instance stopped at the same iteration (always with tol = 0): one copy per array
    for (size_t b = 1; b < B; ++b) same_parity = same_parity && h->h_ctl[b].result_parity == h->h_ctl[0].result_parity;
    if (h->ilv) {
        const int rp = h->h_ctl[0].result_parity;
        if (same_parity) {
            if (rp % 3) RTCHECK(rt::d2d(h->i_z[0], h->i_z[rp % 3], (size_t)h->bpad * dm.nz * sizeof(real), h->stream));
            if (rp % 2) RTCHECK(rt::d2d(h->i_v[0], h->i_v[1], (size_t)h->bpad * dm.nv * sizeof(real), h->stream));
        } else RTCHECK(launch_ilv_collect(h->i_z, h->i_v, h->d_ctl, dm.nz, dm.nv, h->bpad, h->batch, h->stream));
    } else if (same_parity) {
        const int rp = h->h_ctl[0].result_parity;
        if (rp % 3) RTCHECK(rt::d2d(h->d_z[0], h->d_z[rp % 3], B * dm.nz * sizeof(real), h->stream));
        if (rp % 2) RTCHECK(rt::d2d(h->d_v[0], h->d_v[1], B * dm.nv * sizeof(real), h->stream));
    }
    for (size_t b = 0; b < B; ++b) {
        const Ctl &c = h->h_ctl[b];
        if (!h->ilv && !same_parity && c.result_parity % 3) RTCHECK(rt::d2d(h->d_z[0] + b * dm.nz, h->d_z[c.result_parity % 3] + b * dm.nz, sizeof(real) * dm.nz, h->stream));
        if (!h->ilv && !same_parity && c.result_parity % 2) RTCHECK(rt::d2d(h->d_v[0] + b * dm.nv, h->d_v[1] + b * dm.nv, sizeof(real) * dm.nv, h->stream));
        if (c.evals > evals) evals = c.evals;
        if (results) {
            spock_result_t &r = results[b];
            r.status = c.status; r.iterations = c.evals; r.iter_ref = c.iter; r.reserved = 0;
            r.xi1 = c.xi1; r.xi2 = c.xi2; r.xi = c.xi; r.solve_ms = ms;
        }
    }
    RTCHECK(rt::sync(h->stream));
    h->last_ms = ms;
    h->last_iters = evals;
    return SPOCK_OK;
}

int spock_get_z(spock_handle_t h, double *z_out) {
    if (!h || !z_out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (h->alg == SPOCK_ALG_SP) {
        const size_t nr = (size_t)h->bp.dm.nz + h->bp.dm.nv;
        for (int b = 0; b < h->batch; ++b)
            RTCHECK(down(h, z_out + (size_t)b * h->bp.dm.nz, h->s_vec[0] + b * nr, h->bp.dm.nz));
        RTCHECK(rt::sync(h->stream));
        return SPOCK_OK;
    }
    if (h->ilv) RTCHECK(ilv_state_to_stage(h));
    RTCHECK(down(h, z_out, h->d_z[0], (size_t)h->batch * h->bp.dm.nz));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}
int spock_get_v(spock_handle_t h, double *v_out) {
    if (!h || !v_out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (h->alg == SPOCK_ALG_SP) {
        const size_t nr = (size_t)h->bp.dm.nz + h->bp.dm.nv;
        for (int b = 0; b < h->batch; ++b)
            RTCHECK(down(h, v_out + (size_t)b * h->bp.dm.nv, h->s_vec[0] + b * nr + h->bp.dm.nz, h->bp.dm.nv));
        RTCHECK(rt::sync(h->stream));
        return SPOCK_OK;
    }
    if (h->ilv) RTCHECK(ilv_state_to_stage(h));
    RTCHECK(down(h, v_out, h->d_v[0], (size_t)h->batch * h->bp.dm.nv));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}
int spock_set_state(spock_handle_t h, const double *z, const double *v) {
    if (!h || !z || !v) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (h->alg == SPOCK_ALG_SP) {
        const size_t nr = (size_t)h->bp.dm.nz + h->bp.dm.nv;
        for (int b = 0; b < h->batch; ++b) {
            RTCHECK(up(h, h->s_vec[0] + b * nr, z + (size_t)b * h->bp.dm.nz, h->bp.dm.nz));
            RTCHECK(up(h, h->s_vec[0] + b * nr + h->bp.dm.nz, v + (size_t)b * h->bp.dm.nv, h->bp.dm.nv));
        }
        RTCHECK(rt::sync(h->stream));
        return SPOCK_OK;
    }
    RTCHECK(up(h, h->d_z[0], z, (size_t)h->batch * h->bp.dm.nz));
    RTCHECK(up(h, h->d_v[0], v, (size_t)h->batch * h->bp.dm.nv));
    if (h->ilv) { RTCHECK(ilv_stage_to_state(h)); h->stage_valid = true; }
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}
int spock_reset(spock_handle_t h) {
    if (!h) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (h->alg == SPOCK_ALG_SP) { RTCHECK(sp_init_state(h)); return SPOCK_OK; }
    if (h->ilv) {
        for (int k = 0; k < 3; ++k) RTCHECK(rt::dzero(h->i_z[k], (size_t)h->bpad * h->bp.dm.nz * sizeof(real), h->stream));
        for (int k = 0; k < 2; ++k) RTCHECK(rt::dzero(h->i_v[k], (size_t)h->bpad * h->bp.dm.nv * sizeof(real), h->stream));
        RTCHECK(rt::dzero(h->d_z[0], (size_t)h->batch * h->bp.dm.nz * sizeof(real), h->stream));
        RTCHECK(rt::dzero(h->d_v[0], (size_t)h->batch * h->bp.dm.nv * sizeof(real), h->stream));
        h->stage_valid = true;
        RTCHECK(rt::sync(h->stream));
        return SPOCK_OK;
    }
    for (int k = 0; k < 3; ++k) RTCHECK(rt::dzero(h->d_z[k], (size_t)h->batch * h->bp.dm.nz * sizeof(real), h->stream));
    for (int k = 0; k < 2; ++k) RTCHECK(rt::dzero(h->d_v[k], (size_t)h->batch * h->bp.dm.nv * sizeof(real), h->stream));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}

int spock_get_history(spock_handle_t h, int32_t inst, double *xi, double *xi1, double *xi2, int32_t *backtrack, int32_t cap, int32_t *len) {
    if (!h || inst < 0 || inst >= h->batch) return fail(SPOCK_ERR_BAD_ARG, "bad instance");
    DeviceGuard guard(h->device);
    const int cnt = h->h_ctl[inst].evals < h->hist_cap ? h->h_ctl[inst].evals : h->hist_cap;
    if (len) *len = cnt;
    const int k = cnt < cap ? cnt : cap;
    if (k <= 0) return SPOCK_OK;
    std::vector<double> tmp((size_t)k * kHistStride);
    RTCHECK(rt::d2h(tmp.data(), h->d_hist + (size_t)inst * h->hist_cap * kHistStride, tmp.size() * sizeof(double), h->stream));
    RTCHECK(rt::sync(h->stream));
    for (int i = 0; i < k; ++i) {
        if (xi) xi[i] = tmp[(size_t)i * kHistStride + 0];
        if (xi1) xi1[i] = tmp[(size_t)i * kHistStride + 1];
        if (xi2) xi2[i] = tmp[(size_t)i * kHistStride + 2];
        if (backtrack) backtrack[i] = (int32_t)tmp[(size_t)i * kHistStride + 3];
    }
    return SPOCK_OK;
}

// ---- operator-level entry points: the fused kernels driven with inputs that isolate one operator ----
int spock_op_L(spock_handle_t h, const double *z, double *v_out) {
    if (!h || !z || !v_out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    // phase B with w = zbar, no prox: v_out = 1 * L z + 0
    return run_op(h, nullptr, nullptr, z, nullptr, 1., 1., F_NO_ALOCAL | F_NO_SWEEP | F_W_IS_ZBAR | F_NO_PROX, false, true, nullptr, v_out);
}
int spock_op_Lt(spock_handle_t h, const double *v, double *z_out) {
    if (!h || !v || !z_out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    // phase A raw with gamma = -1 and z = 0: z_out = 1 * L'v + 0
    return run_op(h, nullptr, v, nullptr, nullptr, -1., 1., F_RAW_LT | F_NO_SWEEP | F_NO_BLOCAL, true, false, z_out, nullptr);
}
int spock_op_prox_f(spock_handle_t h, const double *x0, double gamma, double *arg) {
    if (!h || !x0 || !arg) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    // phase A + sweeps with v = 0: z - gamma L'0 = z exactly
    return run_op(h, arg, nullptr, nullptr, x0, gamma, 1., F_NO_BLOCAL, true, true, arg, nullptr);
}
int spock_op_prox_h_conj(spock_handle_t h, double sigma, double *arg) {
    if (!h || !arg) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    // phase B with z = zbar = 0: v + sigma L 0 = v exactly
    return run_op(h, nullptr, arg, nullptr, nullptr, 1., sigma, F_NO_ALOCAL | F_NO_SWEEP, false, true, nullptr, arg);
}
int spock_op_T(spock_handle_t h, const double *x0, const double *z, const double *v, double gamma, double sigma, double *zbar_out, double *vbar_out) {
    if (!h || !x0 || !z || !v || !zbar_out || !vbar_out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    return run_op(h, z, v, nullptr, x0, gamma, sigma, 0, true, true, zbar_out, vbar_out);
}
int spock_op_dot(spock_handle_t h, const double *a_z, const double *a_v, const double *b_z, const double *b_v, double gamma, double sigma, double *out) {
    if (!h || !a_z || !a_v || !b_z || !b_v || !out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    // <a, P b>, P = [[I, -gamma L'], [-sigma L, I]] (implicit_l.jl:485-516): L b_z and L' b_v on the device,
    // the two inner products in the reference's sequential order on the host.
    const Dims &dm = h->bp.dm;
    std::vector<double> Lv(dm.nv), Lz(dm.nz);
    int e = spock_op_L(h, b_z, Lv.data());
    if (e) return e;
    e = spock_op_Lt(h, b_v, Lz.data());
    if (e) return e;
    double res = 0.;
    for (int k = 0; k < dm.nz; ++k) res += a_z[k] * (b_z[k] - gamma * Lz[k]);
    for (int k = 0; k < dm.nv; ++k) res += a_v[k] * (-sigma * Lv[k] + b_v[k]);
    *out = res;
    return SPOCK_OK;
}

int spock_get_build_product(spock_handle_t h, int32_t which, double *out, int32_t cap, int32_t *len) {
    if (!h) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    const std::vector<double> *src = nullptr;
    switch (which) {
        case 0: src = &h->bp.P; break;
        case 1: src = &h->bp.K; break;
        case 2: src = &h->bp.Rinv; break;
        case 3: src = &h->bp.ABK; break;
        case 4: src = &h->bp.ls; break;
        case 5: src = &h->bp.sqrtQ; break;
        case 6: src = &h->bp.sqrtR; break;
        case 7: src = &h->bp.sqrtQN; break;
        default: return fail(SPOCK_ERR_BAD_ARG, "unknown build product");
    }
    if (len) *len = (int32_t)src->size();
    if (out && h->dyn_on_device && which <= 3) {   // rebuilt on the device since: read the pool
        const Dims &dm = h->bp.dm;
        const int off = which == 0 ? dm.mP : which == 1 ? dm.mK : which == 2 ? dm.mRinv : dm.mABK;
        const size_t k = src->size() < (size_t)cap ? src->size() : (size_t)cap;
        DeviceGuard guard(h->device);
        RTCHECK(down(h, out, h->d_mats + off, k));
        RTCHECK(rt::sync(h->stream));
        return SPOCK_OK;
    }
    if (out) { const size_t k = src->size() < (size_t)cap ? src->size() : (size_t)cap; std::memcpy(out, src->data(), k * sizeof(double)); }
    return SPOCK_OK;
}

// The dynamics-dependent part of the build step on the device (spock_build.hpp): ricatti_offline (implicit_l.jl:46-97) and the composite
// sweep matrices for new A_w, B_w of the same shape, written into the handle's matrix pool(s).  n_draws = 1: one (A, B) for every
// instance; n_draws = batch (per_instance_dynamics handles): instance b gets A[b], B[b] -- a batch of Monte-Carlo draws of the dynamics
// builds side by side, one workgroup per instance.  The kernel rewrites the dynamics-dependent regions of the LIVE pool in place (A, B,
// K, Rtilde^-1, P, ABK, U / W / V, the sweep blocks); a failed build (Rtilde not positive definite) must leave an instance as it was:
// the pools are snapshot first, the pools of failed instances restored, and the host-side copies touched only on full success.
static int update_dynamics_impl(spock_handle_t h, const double *A, const double *B, int n_draws, int32_t *status_out) {
    DeviceGuard guard(h->device);
    const Dims &dm = h->bp.dm;
    const size_t na = (size_t)dm.d * dm.nx * dm.nx, nb = (size_t)dm.d * dm.nx * dm.nu;
    for (size_t k = 0; k < na * n_draws; ++k) if (!std::isfinite(A[k])) return fail(SPOCK_ERR_BAD_ARG, "A must be finite");
    for (size_t k = 0; k < nb * n_draws; ++k) if (!std::isfinite(B[k])) return fail(SPOCK_ERR_BAD_ARG, "B must be finite");
    const size_t n_pools = h->mats_stride ? (size_t)h->batch : 1, pool_n = h->bp.mats.size();
    // one draw for a handle with per-instance pools: every instance gets it
    std::vector<double> Arep, Brep;
    if (n_pools > 1 && n_draws == 1) {
        Arep.resize(na * n_pools); Brep.resize(nb * n_pools);
        for (size_t b = 0; b < n_pools; ++b) { std::memcpy(&Arep[b * na], A, na * sizeof(double)); std::memcpy(&Brep[b * nb], B, nb * sizeof(double)); }
        A = Arep.data(); B = Brep.data();
    }
    void *dA = nullptr, *dB = nullptr, *dws = nullptr, *dst = nullptr, *dsnap = nullptr;
    const size_t pool_bytes = pool_n * n_pools * sizeof(real);
    int e = rt::dmalloc(&dA, na * n_pools * sizeof(double));
    if (!e) e = rt::dmalloc(&dB, nb * n_pools * sizeof(double));
    if (!e) e = rt::dmalloc(&dws, (size_t)build_ws_doubles(dm) * n_pools * sizeof(double));
    if (!e) e = rt::dmalloc(&dst, n_pools * sizeof(int));
    if (!e) e = rt::dmalloc(&dsnap, pool_bytes);
    std::vector<int> status(n_pools, 0);
    if (!e) e = rt::d2d(dsnap, h->d_mats, pool_bytes, h->stream);
    if (!e) e = rt::h2d(dA, A, na * n_pools * sizeof(double), h->stream);
    if (!e) e = rt::h2d(dB, B, nb * n_pools * sizeof(double), h->stream);
    bool launched = false;
    if (!e) {
        BuildArgs a;
        std::memset(&a, 0, sizeof(a));
        a.dm = dm; a.mats = h->d_mats; a.A = (const double *)dA; a.B = (const double *)dB; a.ws = (double *)dws; a.status = (int *)dst;
        a.mats_stride = (long)pool_n; a.n_inst = (int)n_pools;
        e = launch_build_riccati(a, h->stream);
        launched = true;
    }
    if (!e) e = rt::d2h(status.data(), dst, n_pools * sizeof(int), h->stream);
    if (!e) e = rt::sync(h->stream);
    bool any_failed = false;
    for (size_t b = 0; b < n_pools; ++b) any_failed = any_failed || status[b] != 0;
    if (launched && (e || any_failed)) {   // roll back: a failed instance's pool is again the one of its last successful build
        int e2 = 0;
        for (size_t b = 0; b < n_pools && !e2; ++b)
            if (e || status[b] != 0) e2 = rt::d2d(h->d_mats + b * pool_n, (const real *)dsnap + b * pool_n, pool_n * sizeof(real), h->stream);
        if (!e2) e2 = rt::sync(h->stream);
        if (e2 && !e) e = e2;
    }
    for (void *q : {dA, dB, dws, dst, dsnap}) if (q) rt::dfree(q);
    if (status_out) for (size_t b = 0; b < n_pools; ++b) status_out[b] = status[b] ? SPOCK_ERR_BUILD : SPOCK_OK;
    if (e) return fail(SPOCK_ERR_HIP, std::string("spock_update_dynamics: ") + rt::err(e));
    if (h->graph) { rt::graph_destroy(h->graph); h->graph = nullptr; }        // (the captured launches carry no pool contents, but stay on the safe side)
    if (h->graph_log) { rt::graph_destroy(h->graph_log); h->graph_log = nullptr; }
    if (h->graph_mpc) { rt::graph_destroy(h->graph_mpc); h->graph_mpc = nullptr; }
    if (any_failed) return fail(SPOCK_ERR_BUILD, n_pools > 1 ? "Rtilde is not positive definite for at least one instance (those instances keep their previous dynamics)"
                                                               : "Rtilde is not positive definite (the model keeps its previous dynamics)");
    h->hp.A.assign(A, A + na); h->hp.B.assign(B, B + nb);   // (instance 0's, what spock_get_build_product reports on)
    h->dyn_on_device = true;
    return SPOCK_OK;
}
int spock_update_dynamics(spock_handle_t h, const double *A, const double *B) {
    if (!h || !A || !B) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    return update_dynamics_impl(h, A, B, 1, nullptr);
}
int spock_update_dynamics_batch(spock_handle_t h, const double *A, const double *B, int32_t *status) {
    if (!h || !A || !B) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    if (!h->mats_stride && h->batch > 1) return fail(SPOCK_ERR_BAD_ARG, "the handle was created without per_instance_dynamics: its instances share one matrix pool");
    return update_dynamics_impl(h, A, B, h->batch, status);
}

static int profile_impl(spock_handle_t h, int32_t reps, double *avg_ms, int32_t *launches, int32_t *seq_modes, int32_t *seq_levels,
                        double *seq_ms, int32_t seq_cap, int32_t *seq_n);
int spock_profile_kernels(spock_handle_t h, int32_t reps, double *avg_ms, int32_t *launches) {
    if (!avg_ms) return fail(SPOCK_ERR_BAD_ARG, "bad argument");
    return profile_impl(h, reps, avg_ms, launches, nullptr, nullptr, nullptr, 0, nullptr);
}
int spock_profile_sequence(spock_handle_t h, int32_t reps, int32_t *modes, int32_t *levels, double *avg_ms, int32_t cap, int32_t *n) {
    if (!modes || !levels || !avg_ms || !n) return fail(SPOCK_ERR_BAD_ARG, "bad argument");
    double kinds[4];
    return profile_impl(h, reps, kinds, nullptr, modes, levels, avg_ms, cap, n);
}
static int profile_impl(spock_handle_t h, int32_t reps, double *avg_ms, int32_t *launches, int32_t *seq_modes, int32_t *seq_levels,
                        double *seq_ms, int32_t seq_cap, int32_t *seq_n) {
    if (!h || !avg_ms || reps < 1) return fail(SPOCK_ERR_BAD_ARG, "bad argument");
    if (h->alg != SPOCK_ALG_CP) return fail(SPOCK_ERR_UNSUPPORTED, "kernel profiling is implemented for CP handles");
    DeviceGuard guard(h->device);
    // `reps` further CP iterations on the current state, launched one kernel at a time with a HIP event before
    // and after each launch on the handle's stream; per kind: 0 SUB_A, 1 CROWN, 2 SUB_B or the fused SUB_BA, 3 FULL.
    // (With the fused sequence SUB_A runs once, ahead of the first iteration.)
    const size_t B = (size_t)h->batch;
    const double g = 0.99 / std::sqrt(h->L_norm);
    Ctl c = fresh_ctl(h, g, g, 0., SPOCK_SILENT);
    c.max_iter = 1 << 30;
    std::vector<Ctl> cs(B, c);
    RTCHECK(rt::h2d(h->d_ctl, cs.data(), B * sizeof(Ctl), h->stream));
    const bool split = h->bp.dm.t0 > 0;
    unsigned dbg = 0;
    if (const char *e = std::getenv("SPOCK_AMD_DEBUG_FLAGS")) dbg = (unsigned)std::atoi(e);
    std::vector<int> seq, seq_it, seq_lv;   // (mode, iteration, layer of mid groups) of every launch, in order
    auto push = [&](int mode, int it, int lv = 0) { seq.push_back(mode); seq_it.push_back(it); seq_lv.push_back(lv); };
    const Dims &pdm = h->bp.dm;
    const bool stream_lv = h->ilv && pdm.ilv_stream;
    if (split && h->fused) {
        if (stream_lv) for (int t = pdm.N - 2; t >= pdm.t0; --t) push(MODE_LA0, 0, t);
        else push(h->flat_local ? MODE_FLAT_A : MODE_SUB_A, 0);
        if (h->ilv) push(MODE_ICROWN_A, 0);
    }
    const size_t n_prologue = seq.size();
    for (int it = 0; it < reps; ++it) {
        if (!split) { push(MODE_FULL, it); continue; }
        if (h->flat_local) {
            if (!h->fused) push(MODE_FLAT_A, it);
            push(MODE_SUB_A, it); push(MODE_CROWN, it); push(MODE_SUB_B, it); push(h->fused ? MODE_FLAT_BA : MODE_FLAT_B, it);
            continue;
        }
        if (!h->fused) push(MODE_SUB_A, it);
        if (h->ilv) for (int lv = pdm.n_lv - 2; lv >= 1; --lv) push(MODE_I_UP, it, lv);
        push(MODE_CROWN, it);
        if (h->ilv) for (int lv = 1; lv <= pdm.n_lv - 2; ++lv) push(MODE_I_DOWN, it, lv);
        if (stream_lv) {
            const int t_lb = pdm.ilv_fuse2 ? pdm.N - 4 : pdm.N - 3;
            for (int t = pdm.t0; t <= t_lb; ++t) push(MODE_LB, it, t);
            if (pdm.ilv_fuse2) push(MODE_LBA2, it, pdm.N - 3); else push(MODE_LBA, it, pdm.N - 2);
            for (int t = t_lb; t >= pdm.t0; --t) push(MODE_LA, it, t);
        } else push(h->fused ? (h->rlane ? MODE_SUB_BA_R : MODE_SUB_BA) : (h->lat_regime ? MODE_SUB_B_LAT : MODE_SUB_B), it);
    }
    std::vector<void *> ev(seq.size() * 2, nullptr);
    for (auto &e : ev) RTCHECK(rt::event_create(&e));
    for (size_t k = 0; k < seq.size(); ++k) {
        KArgs a = make_args(h, false, seq_it[k] % 6, dbg);
        a.grp_level = seq_lv[k];
        RTCHECK(rt::event_record(ev[2 * k], h->stream));
        RTCHECK(launch_group(seq[k], a, h->stream));
        RTCHECK(rt::event_record(ev[2 * k + 1], h->stream));
    }
    RTCHECK(rt::sync(h->stream));
    double sum[4] = {0., 0., 0., 0.};
    int cnt[4] = {0, 0, 0, 0};
    // per launch of ONE iteration (the last one's sequence): average over the iterations
    const size_t per_it = reps > 0 ? (seq.size() - n_prologue) / (size_t)reps : 0;
    std::vector<double> it_sum(per_it, 0.);
    for (size_t k = 0; k < seq.size(); ++k) {
        float ms = 0.f;
        RTCHECK(rt::event_elapsed(&ms, ev[2 * k], ev[2 * k + 1]));
        if (k >= n_prologue && per_it) it_sum[(k - n_prologue) % per_it] += ms;
        // (element-major regime: the crown's node-local kernels count with the crown, kind 1)
        // (level-streaming kernels: the bottom stage's LBA is kind 2, the other stages' LB / LA count with the crown)
        const int kind = (seq[k] == MODE_SUB_A || seq[k] == MODE_ICROWN_A || seq[k] == MODE_LA0 || seq[k] == MODE_FLAT_A) ? 0
                         : (seq[k] == MODE_CROWN || seq[k] == MODE_I_UP || seq[k] == MODE_I_DOWN || seq[k] == MODE_LB || seq[k] == MODE_LA) ? 1 : seq[k] == MODE_FULL ? 3 : 2;
        sum[kind] += ms; cnt[kind] += 1;
    }
    for (auto e : ev) rt::event_destroy(e);
    for (int q = 0; q < 4; ++q) { avg_ms[q] = cnt[q] ? sum[q] / cnt[q] : 0.; if (launches) launches[q] = cnt[q]; }
    if (seq_n) {
        *seq_n = (int32_t)per_it;
        for (size_t k = 0; k < per_it && (int32_t)k < seq_cap; ++k) {
            seq_modes[k] = seq[n_prologue + k]; seq_levels[k] = seq_lv[n_prologue + k]; seq_ms[k] = it_sum[k] / reps;
        }
    }
    // leave a consistent state behind: the iterate after `reps` iterations goes back to slot 0
    if (h->ilv) {
        if (reps % 3) RTCHECK(rt::d2d(h->i_z[0], h->i_z[reps % 3], (size_t)h->bpad * h->bp.dm.nz * sizeof(real), h->stream));
        if (reps % 2) RTCHECK(rt::d2d(h->i_v[0], h->i_v[1], (size_t)h->bpad * h->bp.dm.nv * sizeof(real), h->stream));
        h->stage_valid = false;
    } else {
        if (reps % 3) RTCHECK(rt::d2d(h->d_z[0], h->d_z[reps % 3], B * h->bp.dm.nz * sizeof(real), h->stream));
        if (reps % 2) RTCHECK(rt::d2d(h->d_v[0], h->d_v[1], B * h->bp.dm.nv * sizeof(real), h->stream));
    }
    for (auto &x : cs) x.done = 1;
    RTCHECK(rt::h2d(h->d_ctl, cs.data(), B * sizeof(Ctl), h->stream));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}

int spock_debug_stamps(spock_handle_t h, uint64_t *out /* [3][64][2] */) {
    if (!h || !out) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    RTCHECK(rt::d2h(out, h->d_dbg, sizeof(uint64_t) * 3 * 64 * 2, h->stream));
    RTCHECK(rt::sync(h->stream));
    return SPOCK_OK;
}

int spock_L_norm(spock_handle_t h, double *in_use, double *closed_form) {
    if (!h) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    if (in_use) *in_use = h->L_norm;
    if (closed_form) *closed_form = h->bp.L_norm_closed;
    return SPOCK_OK;
}

int spock_get_timing(spock_handle_t h, double *iter_ms_avg, int32_t *iters_timed) {
    if (!h) return fail(SPOCK_ERR_BAD_ARG, "NULL argument");
    if (iter_ms_avg) *iter_ms_avg = h->last_iters > 0 ? h->last_ms / h->last_iters : 0.;
    if (iters_timed) *iters_timed = h->last_iters;
    return SPOCK_OK;
}

}  // extern "C"
