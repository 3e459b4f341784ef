// spock_host.cpp -- host-side build step of the product: everything build_cpock / build_spock compute
// once per problem structure (/root/reference/src/models/cpock.jl:1-163, src/models/spock.jl:1-210):
//   * sizes and offsets              get_nz / get_nv (implicit_l.jl:5-44), cpock.jl:51-60
//   * offline Riccati recursion      ricatti_offline (implicit_l.jl:46-97)
//   * kernel projector               cpock.jl:71-81, :125-128
//   * square roots of the costs      cpock.jl:119-121
// plus the tiling of the tree into per-workgroup subtrees.  The results are packed into one matrix pool
// that the kernels address through offsets in `Dims`.
//
// Because the reference's ricatti_offline only uses A_w, B_w and the (uniform) tree shape, P_i, K_i and
// Rtilde_i depend on the STAGE of node i only and ABK_j on (stage, child slot); they are stored per stage.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "spock_internal.hpp"

namespace spock {
namespace {

struct Mat {  // dense column-major
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.) {}
    double &operator()(int i, int j) { return a[(size_t)i + (size_t)j * r]; }
    double operator()(int i, int j) const { return a[(size_t)i + (size_t)j * r]; }
    static Mat from(const double *p, int r_, int c_) { Mat m(r_, c_); std::memcpy(m.a.data(), p, sizeof(double) * r_ * c_); return m; }
    static Mat eye(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.; return m; }
};
Mat T(const Mat &x) { Mat y(x.c, x.r); for (int i = 0; i < x.r; ++i) for (int j = 0; j < x.c; ++j) y(j, i) = x(i, j); return y; }
Mat mul(const Mat &x, const Mat &y) {
    Mat z(x.r, y.c);
    for (int j = 0; j < y.c; ++j)
        for (int i = 0; i < x.r; ++i) {
            double s = 0.;
            for (int k = 0; k < x.c; ++k) s += x(i, k) * y(k, j);
            z(i, j) = s;
        }
    return z;
}
Mat add(const Mat &x, const Mat &y) { Mat z = x; for (size_t k = 0; k < z.a.size(); ++k) z.a[k] += y.a[k]; return z; }

// symmetric eigen-decomposition by cyclic Jacobi rotations: S = V diag(w) V'
void sym_eig(const Mat &S, Mat &V, std::vector<double> &w) {
    const int n = S.r;
    Mat a = S;
    V = Mat::eye(n);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += a(p, q) * a(p, q);
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a(p, q);
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (a(q, q) - a(p, p)) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = a(k, p), y = a(k, q); a(k, p) = c * x - s * y; a(k, q) = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = a(p, k), y = a(q, k); a(p, k) = c * x - s * y; a(q, k) = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = V(k, p), y = V(k, q); V(k, p) = c * x - s * y; V(k, q) = s * x + c * y; }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = a(i, i);
}

// principal square root of a symmetric PSD matrix; diagonal input -> element-wise (Julia's sqrt(::Diagonal))
Mat sym_sqrt(const Mat &S) {
    const int n = S.r;
    bool diag = true;
    for (int j = 0; j < n && diag; ++j) for (int i = 0; i < n; ++i) if (i != j && S(i, j) != 0.) { diag = false; break; }
    Mat out(n, n);
    if (diag) { for (int i = 0; i < n; ++i) out(i, i) = std::sqrt(S(i, i)); return out; }
    Mat V; std::vector<double> w;
    sym_eig(S, V, w);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0.;
            for (int k = 0; k < n; ++k) s += V(i, k) * std::sqrt(w[k] > 0. ? w[k] : 0.) * V(j, k);
            out(i, j) = s;
        }
    return out;
}

// inverse of a symmetric positive definite matrix through its Cholesky factor; false if not SPD
bool spd_inverse(const Mat &S, Mat &inv) {
    const int n = S.r;
    Mat L(n, n);
    for (int j = 0; j < n; ++j) {
        double s = S(j, j);
        for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
        if (!(s > 0.)) return false;
        const double dj = std::sqrt(s);
        L(j, j) = dj;
        for (int i = j + 1; i < n; ++i) {
            double t = S(i, j);
            for (int k = 0; k < j; ++k) t -= L(i, k) * L(j, k);
            L(i, j) = t / dj;
        }
    }
    inv = Mat(n, n);
    std::vector<double> x(n);
    for (int c = 0; c < n; ++c) {
        for (int i = 0; i < n; ++i) {
            double s = (i == c) ? 1. : 0.;
            for (int k = 0; k < i; ++k) s -= L(i, k) * x[k];
            x[i] = s / L(i, i);
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = x[i];
            for (int k = i + 1; k < n; ++k) s -= L(k, i) * x[k];
            x[i] = s / L(i, i);
        }
        for (int i = 0; i < n; ++i) inv(i, c) = x[i];
    }
    for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) { const double m = 0.5 * (inv(i, j) + inv(j, i)); inv(i, j) = m; inv(j, i) = m; }
    return true;
}

int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

}  // namespace

int group_threads(int d, int stages) { return stages < 2 ? 0 : (ipow(d, stages - 1) - 1) / (d - 1); }

// Stage at which the tree is cut.  A group (subtree or crown) runs as ONE workgroup with one thread per
// non-leaf node and nx doubles of LDS per thread for the q / x exchange of the S1 sweeps.
int choose_split_stage(int N, int d, int nx, int nu, int batch) {
    int max_thr = 512;
    const int lds_thr = (144 * 1024 / 8) / (nx > 0 ? nx : 1);
    if (lds_thr < max_thr) max_thr = lds_thr;
    // one workgroup for the whole tree only while the tree is small: beyond ~128 non-leaf nodes the subtree kernels on
    // many CUs win even for a single instance (measured N = 10, nx = 5: 54 us whole-tree vs 42 us split)
    if (group_threads(d, N) <= (max_thr < 128 ? max_thr : 128)) return 0;
    int t0 = (N - 1) / 2;
    while (t0 < N - 2 && group_threads(d, N - t0) > max_thr) ++t0;
    if (group_threads(d, N - t0) > max_thr || group_threads(d, t0 + 1) > max_thr) return -1;
    // Batched regime (thousands of workgroups in flight, one thread per node): subtrees of two waves instead of one are
    // ~7 % faster (measured: N = 13, d = 2, 256 instances, 490 k -> 527 k iterations/s at t0 = 5; t0 = 4 the same, t0 = 3 slower;
    // for a single instance t0 = 6 stays best: 42.7 vs 46 us per iteration)
    if (nx <= 8 && nu <= 8 && (long)batch * ipow(d, t0) >= 1024)
        while (t0 > 1 && group_threads(d, N - (t0 - 1)) <= 128) --t0;
    return t0;
}

// Element-major batched regime: a subtree workgroup keeps one q / x slot of nx x 64 doubles per non-leaf node in LDS; subtrees
// of up to ~40 KB of slots leave room for three workgroups per CU.  The crown (stages <= t0) exchanges through global memory.
int choose_split_stage_ilv(int N, int d, int nx) {
    if (!(std::getenv("SPOCK_AMD_ILV_STREAM") && std::getenv("SPOCK_AMD_ILV_STREAM")[0] == '0')) {
        // level-streaming kernels below the split: a small crown (one top group, no mid layers) and everything else streamed.
        // Measured on MI355X (config 2, 2048 instances, one stream): split 3 / 4 / 5 / 6 / 8 -> 2.50 / 2.57 / 2.58 / 2.62 / 2.61 ms per
        // iteration (bench: 830 k against 809 k iterations/s for 3 against 4).  The smaller the crown the better; what stops at 3
        // is that the instance-major kernels of the operator entry points share the split and their subtrees must fit one workgroup.
        int t0 = N - 2 < 3 ? N - 2 : 3;
        while (t0 < N - 2 && group_threads(d, N - t0) > 512) ++t0;
        return t0 < 1 ? 1 : t0;
    }
    int S = 2;
    while (S + 1 <= N - 1 && (long)group_threads(d, S + 1) * nx * 64 * 8 <= 40 * 1024) ++S;
    return N - S;
}

// non-leaf levels a group of the element-major regime may span (LDS slots of nx x 64 doubles per non-leaf node, <= ~40 KB)
static int ilv_group_span(int d, int nx) {
    int S = 2;
    while (S < 12 && (long)group_threads(d, S + 1) * nx * 64 * 8 <= 40 * 1024) ++S;
    return S - 1;
}

// level-streaming kernels: waves per workgroup and LDS exchange rows (role waves hand their partial results to the finishing
// waves through rows of 64 doubles, double-buffered over the nodes of the chunk)
int ilevel_waves(int mode, const Dims &dm) { return (mode == MODE_LBA || mode == MODE_LBA2) ? 2 + 2 * dm.d : (mode == MODE_LB ? 2 + dm.d : 4); }   // X, R, C_c (, L_c)
int ilevel_ex_doubles(int mode, const Dims &dm, int want_xi) {
    if (mode == MODE_LBA2)   // + x and q of the children, c_i / e_i and [y; s_ch; tau_ch] of the parent (double-buffered over the parents)
        return (2 * lba_exchange_rows(dm.nx, dm.nu, dm.d, want_xi != 0) + 2 * dm.d * dm.nx + 2 * (dm.nx + dm.nu) + 2 * (dm.ny + 2 * dm.d)) * 64;
    if (mode == MODE_LBA) return 2 * lba_exchange_rows(dm.nx, dm.nu, dm.d, want_xi != 0) * 64;
    if (mode == MODE_LB) return 2 * lb_exchange_rows(dm.nx, dm.nu, dm.d, want_xi != 0) * 64;
    return 0;
}

IlvGeo ilv_geometry(int mode, const Dims &dm, int level, int want_xi) {
    IlvGeo g;
    const int d = dm.d, nx = dm.nx;
    const int sub_nl = group_threads(d, dm.sub_stages), crown_nl = group_threads(d, dm.t0 + 1);
    g.cchunk = sub_nl > 0 ? sub_nl : 1;
    g.n_cchunks = (crown_nl + g.cchunk - 1) / g.cchunk;
    int nW = 4;
    if (const char *e = std::getenv("SPOCK_AMD_ILV_WAVES")) { nW = std::atoi(e); if (nW < 1) nW = 1; if (nW > 4) nW = 4; }
    g.ex_doubles = 0;
    g.threads = 64 * nW;
    if (mode == MODE_FINALIZE) { g.threads = 64; g.grid_x = 1; }
    else if (mode == MODE_ICROWN_A) g.grid_x = g.n_cchunks;
    else if (mode == MODE_SUB_A || mode == MODE_SUB_BA) { g.grid_x = dm.nsub; g.ex_doubles = sub_nl * nx * 64; }
    else if (mode == MODE_LB || mode == MODE_LBA || mode == MODE_LA || mode == MODE_LA0 || mode == MODE_LBA2) {   // `level` = the tree stage
        const int chunk = dm.ls_chunk_t[level] > 0 ? dm.ls_chunk_t[level] : 1;
        g.grid_x = (ipow(d, level) + chunk - 1) / chunk;
        g.ls_chunk_k = chunk;
        g.ex_doubles = ilevel_ex_doubles(mode, dm, want_xi);
        g.threads = 64 * ilevel_waves(mode, dm);
        g.ls_cnt = ipow(d, level); g.ls_first = dm.first[level]; g.ls_first_child = dm.first[level + 1];
        g.ls_slot = dm.ilv_stream ? dm.ls_slot0[level] : 0;
    }
    else {   // CROWN (top group, level 0), I_UP / I_DOWN (mid groups of `level`)
        const int lv = mode == MODE_CROWN ? 0 : level;
        g.grid_x = ipow(d, dm.lv_root[lv]);
        g.ex_doubles = group_threads(d, dm.lv_root[lv + 1] - dm.lv_root[lv] + 1) * nx * 64;
    }
    // [exchange slots][maxima: waves x 3 x 64][64 live flags]; level kernels: the maxima reuse the exchange rows (spock_ilevel.hpp)
    const size_t red = (size_t)(g.threads / 64) * 3 * 64 * 8;
    const bool lvk = (mode == MODE_LB || mode == MODE_LBA || mode == MODE_LA || mode == MODE_LA0 || mode == MODE_LBA2);
    g.lds_bytes = (lvk ? ((size_t)g.ex_doubles * 8 > red ? (size_t)g.ex_doubles * 8 : red) : (size_t)g.ex_doubles * 8 + red) + 64 * 4 + 64;
    return g;
}

// Role-lane subtree kernel (spock_rgroup.hpp) in place of MODE_SUB_BA: the fused iteration of a compile-time shape with shared costs,
// groups of at most 64 non-leaf nodes (lane = node) and so few workgroups in flight that a node-local phase is one thread's latency
// rather than the chip's throughput (measured crossover: see DESIGN.md).  SPOCK_AMD_RLANE=0 / 1 overrides the batch rule.
bool rlane_applies(const Dims &dm, int batch, bool fused) {
    const char *e = std::getenv("SPOCK_AMD_RLANE");
    if (e && e[0] == '0') return false;
    if (!fused || dm.t0 <= 0) return false;
    if (!have_static_variant(dm.nx, dm.nu, dm.d, dm.ny)) return false;
    if (!(dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1)) return false;
    if (group_threads(dm.d, dm.sub_stages) > 64 || group_threads(dm.d, dm.t0 + 1) > 64) return false;   // lane = node; 8 lanes per node in the sweeps
    if (dm.nx > 8 || dm.nu > 8 || 64 * (2 + 2 * dm.d) > kRgroupThreads) return false;
    if ((long)rgroup_lds_doubles(dm.nx, dm.nu, dm.d, true) * 8 + 64 > 128 * 1024) return false;
    if (e && e[0] == '1') return true;
    // measured (profiles/r03/rlane_batch_crossover.log; 4096 leaves = 65 workgroups per instance, 512 leaves = 17): against SUB_BA
    // (nx + nu <= 8) 1.3-1.6 x faster up to ~400 workgroups in flight, slower from ~520 on; against the three-launch sequence of the
    // larger shapes still 1.07-1.16 x faster at 520-750
    return (long)batch * (dm.nsub + 1) <= (dm.nx + dm.nu <= 8 ? 400 : 800);
}

bool team_local_applies(const Dims &dm) {
    if (std::getenv("SPOCK_AMD_NO_TEAM_LOCAL")) return false;
    const bool big = (dm.nx > 8 || dm.nu > 8) || std::getenv("SPOCK_AMD_FORCE_TEAM_LOCAL");   // (the override is for experiments)
    return big && dm.diag_costs && dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1 && !dm.risk_has_soc;
}

// Flat node-local kernels (spock_flat_local.hpp): T lanes per node as in the subtree kernels' lane teams; a team is part of ONE wave
// and the waves of a workgroup share nothing but the final reduction, so workgroups are small -- 256 threads: 4 nodes at T = 64
// (nx = 64 / 50), 8 at T = 32 (nx = 20) -- and a launch has thousands of them.
FlatGeo flat_geometry(const Dims &dm) {
    FlatGeo g;
    int T = 8;
    while (T < dm.nx || T < dm.nu) T *= 2;
    g.team = T;
    int thr = 256;
    if (const char *e = std::getenv("SPOCK_AMD_FLAT_THREADS")) { thr = std::atoi(e); if (thr < 64) thr = 64; if (thr > 256) thr = 256; thr = (thr / 64) * 64; }
    g.threads = thr;
    g.nteams = g.threads / T;
    g.chunks = (dm.nn + g.nteams - 1) / g.nteams;
    const int m = dm.ny + 2 * dm.d;
    // per team: flat_scratch_doubles, sized for the host emulation's extra rows too (the GPU uses the first m + nu + d doubles)
    g.ex_doubles = g.nteams * (m + dm.nu + dm.d + (2 + 3 * dm.d) * T) + (dm.nx * dm.nu + dm.nu * dm.nu + m * m);   // + the staged U_t, Rt^-1_t, LS (flat_mats_doubles)
    g.lds_bytes = (size_t)g.ex_doubles * 8 + (4 + 128 + 64) * 8;          // + the reduction scratch / stamps / SP sums of the group kernels' carve
    return g;
}
bool flat_local_applies(const Dims &dm, int batch) {
    if (const char *e = std::getenv("SPOCK_AMD_FLAT_LOCAL")) { if (e[0] == '0') return false; }
    if (dm.t0 <= 0) return false;                                  // whole tree in one workgroup: nothing to flatten
    if (!team_local_applies(dm)) return false;
    // with the dedicated sweep kernels (spock_skernels.hip) the subtree / crown kernels need no lane-team scratch at all: groups whose
    // sweep vectors alone fill the LDS (nx = nu = 50 with 127-node groups: examples/server_heat/scaling.jl at N = 14, 15) qualify too
    if (sweep_kernels_apply(dm)) return true;
    return launch_geometry(MODE_SUB_A, dm, batch).team_local != 0 && launch_geometry(MODE_CROWN, dm, batch).team_local != 0;
}

LaunchGeo launch_geometry(int mode, const Dims &dm, int batch) {
    LaunchGeo g;
    if (mode == MODE_SUB_B_SP || mode == MODE_SUB_B_LAT) mode = MODE_SUB_B;
    if (mode == MODE_FULL_SP) mode = MODE_FULL;
    const int d = dm.d, nx = dm.nx, nu = dm.nu;
    auto r64 = [](int t) { return t < 64 ? 64 : ((t + 63) / 64) * 64; };
    int stage_sets[2], n_sets = 0;
    switch (mode) {
        case MODE_FULL: stage_sets[n_sets++] = dm.N; g.grid_x = 1; break;
        case MODE_SUB_A: case MODE_SUB_B: case MODE_SUB_BA: stage_sets[n_sets++] = dm.sub_stages; stage_sets[n_sets++] = dm.t0 + 1; g.grid_x = dm.nsub + 1; break;
        case MODE_CROWN: stage_sets[n_sets++] = dm.t0 + 1; g.grid_x = 1; break;
        default: stage_sets[n_sets++] = 2; g.grid_x = 1; break;
    }
    int nl = 0, widest = 1, max_stages = 2;
    long team_doubles = 0, team_doubles_noxl = 0, xl_doubles = 0;
    for (int k = 0; k < n_sets; ++k) {
        const int st = stage_sets[k];
        const int nonleaf = group_threads(d, st), leaves = ipow(d, st - 1);
        if (nonleaf > nl) nl = nonleaf;
        if (st > max_stages) max_stages = st;
        if (st >= 2 && ipow(d, st - 2) > widest) widest = ipow(d, st - 2);
        const long td = (long)nonleaf * (nx + 2 * nu) + (long)leaves * nx;
        if (td > team_doubles) team_doubles = td;
        if ((long)nonleaf * (nx + 2 * nu) > team_doubles_noxl) team_doubles_noxl = (long)nonleaf * (nx + 2 * nu);
        if ((long)leaves * nx > xl_doubles) xl_doubles = (long)leaves * nx;
    }
    int T = 8;
    while (T < nx || T < nu) T *= 2;
    const long mats_bytes = ((long)(max_stages - 1) * dm.sw_stride + (long)d * nx * nu) * 8;
    // 160 KiB of LDS per CU on MI355X; the launcher raises the dynamic-LDS limit of the kernel when more than 64 KiB is used
    // (config 5's shape: 12 KiB more let the subtree kernels run 1024 threads = 16 lane teams next to the sweep vectors; for the
    //  other shapes the measured optimum of the geometry rules below was found with 144 KiB)
    const long budget = ((nx == 64 && nu == 16 && d == 2) ? 156 : 144) * 1024, tail = (4 + 128 + 64) * 8;   // reduction scratch + flag, 64 diagnostic clock stamps, SP sum reduction
    // Lane teams pay off when a node's mat-vecs are big (measured on MI355X: nx = nu = 20, N = 10: 3058 -> 561 us per
    // iteration).  For the small compile-time shapes (nx <= 8) one thread per node with the vectors in registers and
    // the matrices in SGPRs is as fast per level and keeps workgroups at one wave, which the batched regime needs
    // (team sweeps there: 360k -> 218k iterations/s at 512 instances).  SPOCK_AMD_TEAM=1 / 0 overrides.
    // Small shapes in the LATENCY regime (few workgroups in flight) use the register-resident team sweep: thread =
    // (node, matrix row), the rows live in registers, a tree level is ~0.1-0.2 us instead of 0.7-1.0 us.  It needs
    // T x (non-leaf nodes of the group) threads; the crown kernel (one workgroup per instance) always qualifies.
    bool want_team = (nx > 8 || nu > 8);
    const bool small = (nx <= 8 && nu <= 8);
    g.xl_global = 0; g.xl_stride = (int)xl_doubles;
    // large nx, deep groups (config 5: nx = 64, 127-node groups = 163 KB): keep c / e / u in LDS and put the group-leaf
    // vector (half of the footprint, touched only at the bottom level) in global memory rather than fall back to the
    // one-thread-per-node kernels (34 ms per iteration there)
    const bool force_xl = std::getenv("SPOCK_AMD_FORCE_XL_GLOBAL") != nullptr;   // tests: exercise the path on small problems
    if (want_team && (team_doubles * 8 + tail > 144 * 1024 || force_xl) && !std::getenv("SPOCK_AMD_NO_XL_GLOBAL")) { g.xl_global = 1; team_doubles = team_doubles_noxl; }
    const long wgs = (long)batch * g.grid_x;
    bool want_reg = small && (long)T * nl <= 512 && (mode == MODE_CROWN || wgs < 512);
    if (const char *e = std::getenv("SPOCK_AMD_TEAM")) want_team = (e[0] == '1');
    if (const char *e = std::getenv("SPOCK_AMD_TEAM_REG")) want_reg = small && (long)T * nl <= 512 && (e[0] == '1');
    g.team_reg = 0;
    if (mode != MODE_FINALIZE && want_reg && team_doubles * 8 + tail <= budget) { g.team = T; g.team_reg = 1; }
    else g.team = (mode != MODE_FINALIZE && want_team && T <= 64 && team_doubles * 8 + tail <= budget) ? T : 0;
    g.threads = r64(nl);
    g.team_local = 0;
    if (g.team && !g.team_reg && mode != MODE_FINALIZE && team_local_applies(dm)) {
        // lane-team node-local phases: 512 threads = 512 / T teams (fewer rounds over the group's nodes: configs 3 / 4 /
        // 5-small 381 / 188 / 335 -> 325 / 152 / 289 us per iteration against 256 threads), each team with its scratch
        // behind the sweep vectors; 256 threads when that scratch does not fit
        // (1024 threads = one round for 31-node subtrees at T = 32: a further 2-4 %; the crown kernel has no node-local
        // phases and its level barriers are cheaper with 512)
        // (crown: 512 threads for the scalar level kernels, whose barriers are cheaper then; 1024 = 16 waves for the MFMA levels,
        //  which have 5 / 9 row tiles of work items per level, times the shares of the node tiles)
        const bool mfma_lv = (nx == 64 && nu == 16 && d == 2) && !std::getenv("SPOCK_AMD_NO_MFMA");
        // (many workgroups in flight -- a batch of closed-loop simulations: 512-thread workgroups pack the CUs better than one-round
        //  1024-thread ones.  Measured, config 4, 64 simulations x 6 steps: 501 -> 455 ms; 8 simulations 133 -> 138 ms, one
        //  instance 103 -> 119 us per iteration, hence only from two workgroups per CU on)
        const bool many_wgs = wgs >= 512 && !mfma_lv;
        for (int thr = (((mode == MODE_CROWN && !mfma_lv) || many_wgs) ? 512 : 1024); thr >= 256 && !g.team_local; thr /= 2) {
            if (const char *e = std::getenv("SPOCK_AMD_TL_THREADS")) thr = std::atoi(e);
            const int nteams = thr / g.team;
            const long scr = (long)nteams * ((dm.ny + 2 * d) + nu + (2 + 2 * d) * g.team);
            if (nteams >= 1 && (team_doubles + scr) * 8 + tail <= budget) { g.team_local = 1; team_doubles += scr; g.threads = thr; }
            if (std::getenv("SPOCK_AMD_TL_THREADS")) break;
        }
    }
    if (g.team_local) {}
    // (Lane-team node-local phases on the register-resident sweep's (node, element) thread mapping were tried for the
    // compile-time shapes in the single-instance regime: no gain -- phase B is bound by its two cold memory round trips
    // either way -- so those shapes keep one thread per node in the node-local phases.)
    else if (g.team_reg) g.threads = r64(g.team * nl);
    else if (g.team) {
        int want = r64(g.team * widest);
        if (want > 512) want = 512;
        if (want > g.threads) g.threads = want;
    }
    if (mode == MODE_FINALIZE) g.threads = 64;
    g.ex_doubles = g.team ? (int)team_doubles : g.threads * nx;
    const long base = (long)g.ex_doubles * 8 + tail;
    g.lds_mats = (mode != MODE_FINALIZE && base + mats_bytes <= budget) ? 1 : 0;
    g.lds_bytes = (size_t)(base + (g.lds_mats ? mats_bytes : 0));
    return g;
}

std::string host_build(const HostProblem &hp, int split_stage, BuildProducts &out, bool &unsupported, int batch) {
    unsupported = false;
    const int N = hp.N, d = hp.d, nx = hp.nx, nu = hp.nu, ny = hp.ny;
    if (d <= 1) return "Branching factor d must be larger than 1";       // scenario_tree.jl:62-64
    if (N < 2 || N > kMaxStages) { unsupported = N > kMaxStages; return "N (stages including the root) must be in [2, 24]"; }
    if (nx < 1 || nu < 1 || ny < 1) return "nx, nu, ny must be positive";
    if (nx > kDynMaxNx || nu > kDynMaxNu || d > kDynMaxD || ny > kDynMaxNy) {
        unsupported = true;
        return "dimensions exceed this build's limits (nx, nu <= 64, d <= 4, ny <= 16)";
    }
    if ((int)hp.cone_kind.size() > kMaxCones) { unsupported = true; return "more than 8 sub-cones"; }
    double nodes = (std::pow((double)d, N) - 1.) / (d - 1);
    if (nodes * (nx + nu + ny + 8) * 3. > 2.0e9) { unsupported = true; return "tree too large for 32-bit offsets"; }

    Dims &dm = out.dm;
    std::memset(&dm, 0, sizeof(dm));
    dm.N = N; dm.d = d; dm.nx = nx; dm.nu = nu; dm.ny = ny;
    dm.n = (ipow(d, N) - 1) / (d - 1);
    dm.nl = ipow(d, N - 1);
    dm.nn = (ipow(d, N - 1) - 1) / (d - 1);
    const int n = dm.n, nn = dm.nn, nl = dm.nl;
    dm.zx = 0; dm.zu = n * nx; dm.zs = dm.zu + nn * nu; dm.zt = dm.zs + n; dm.zy = dm.zt + (n - 1);
    dm.nz = dm.zy + nn * ny;
    dm.o2 = nn * ny; dm.o3 = dm.o2 + nn; dm.o4 = dm.o3 + nx * (n - 1); dm.o5 = dm.o4 + nu * (n - 1);
    dm.o6 = dm.o5 + (n - 1); dm.o7 = dm.o6 + (n - 1); dm.o11 = dm.o7 + nn * (nx + nu);
    dm.o12 = dm.o11 + nl * nx; dm.o13 = dm.o12 + nl; dm.o14 = dm.o13 + nl;
    dm.nv = dm.o14 + nl * nx;
    dm.nQ = hp.nQ; dm.nR = hp.nR; dm.nQN = hp.nQN;
    if (!(hp.nQ == 1 || hp.nQ == n - 1) || !(hp.nR == 1 || hp.nR == n - 1) || !(hp.nQN == 1 || hp.nQN == nl))
        return "cost matrix counts must be 1 or n-1 / n-1 / n_leaf";
    dm.n_cones = (int)hp.cone_kind.size();
    int csum = 0;
    for (int k = 0; k < dm.n_cones; ++k) {
        const int kind = hp.cone_kind[k];
        if (kind < 0 || kind > 4 || hp.cone_dim[k] < 0) return "invalid cone";
        dm.cone_kind[k] = kind == 0 ? 3 : (kind == 3 ? 0 : kind);       // MOI.dual_set: Zeros <-> Reals
        dm.cone_dim[k] = hp.cone_dim[k];
        csum += hp.cone_dim[k];
    }
    if (csum != ny) return "cone dimensions must add up to length(b)";
    dm.risk_has_soc = 0; dm.elem_kind_mask = 0ull;
    for (int k = 0, e = 0; k < dm.n_cones; ++k)
        for (int q = 0; q < dm.cone_dim[k]; ++q, ++e) {
            if (dm.cone_kind[k] == 4) dm.risk_has_soc = 1;
            dm.elem_kind_mask |= (unsigned long long)(dm.cone_kind[k] & 3) << (2 * e);
        }
    dm.x_min = hp.x_min; dm.x_max = hp.x_max; dm.u_min = hp.u_min; dm.u_max = hp.u_max;
    for (int t = 0; t <= N; ++t) dm.first[t] = (ipow(d, t) - 1) / (d - 1);

    int t0 = split_stage > 0 ? split_stage : choose_split_stage(N, d, nx, nu, batch);
    if (t0 < 0) { unsupported = true; return "tree does not fit the two-level subtree tiling"; }
    {   // Small trees of the shapes with matrix-core sweep kernels (spock_skernels.hip): split them as well.  The whole-tree kernel runs
        // its sweep levels as scalar lane-team levels with the matrices fetched from L2 (nx = nu = 50, N = 7: 1.7 ms per SP iteration) and,
        // once the tree's sweep vectors pass the LDS (N = 8 at nx = 50), falls back to one thread per node (scaling.jl N = 8: 3.6 s);
        // flat node-local kernels + UP / CROWN / DOWN cost five launches and ~0.2 ms per SP iteration whatever the tree's size.
        auto shared_diag = [&](const std::vector<double> &m, int cnt, int n) {
            if (cnt != 1) return false;
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) if (i != j && m[(size_t)i + (size_t)j * n] != 0.) return false;
            return true;
        };
        bool soc = false;
        for (int k : hp.cone_kind) soc = soc || k == 4;
        const bool sweep_shape = (nx == 64 && nu == 16 && d == 2) || (nx == 20 && nu == 20 && (d == 2 || d == 3)) || (nx == 50 && nu == 50 && d == 2);
        if (split_stage <= 0 && t0 == 0 && N >= 3 && sweep_shape && !soc && !std::getenv("SPOCK_AMD_NO_SMALL_SPLIT") &&
            shared_diag(hp.Q, hp.nQ, nx) && shared_diag(hp.R, hp.nR, nu) && shared_diag(hp.QN, hp.nQN, nx))
            t0 = (N - 1) / 2 < 1 ? 1 : (N - 1) / 2;
    }
    if (t0 > 0) {
        if (t0 > N - 2) return "split_stage must leave subtrees of at least two stages";
        const int lim = (144 * 1024 / 8) / nx < 512 ? (144 * 1024 / 8) / nx : 512;
        if (group_threads(d, N - t0) > lim || group_threads(d, t0 + 1) > lim) { unsupported = true; return "split_stage gives groups larger than one workgroup"; }
    }
    dm.t0 = t0;
    dm.nsub = t0 > 0 ? ipow(d, t0) : 1;
    dm.sub_stages = N - t0;
    {   // hierarchy of groups of the element-major regime: bottom layer at t0, layers of `span` non-leaf levels above it
        const int span = ilv_group_span(d, nx);
        int roots[8], n = 0;
        for (int t = t0; t > 0 && n < 7; t -= span) roots[n++] = t;   // bottom layer first
        roots[n++] = 0;                                                // the top group starts at the tree root
        dm.n_lv = n;
        for (int k = 0; k < n; ++k) dm.lv_root[k] = roots[n - 1 - k];
        dm.n_crown_groups = 0;
        for (int k = 0; k + 1 < n; ++k) dm.n_crown_groups += ipow(d, dm.lv_root[k]);
        // level-streaming kernels for the stages t0 .. N-2 (spock_ilevel.hpp); SPOCK_AMD_ILV_STREAM=0: the bottom-group kernels
        dm.ilv_stream = t0 > 0 ? 1 : 0;
        if (const char *e = std::getenv("SPOCK_AMD_ILV_STREAM")) dm.ilv_stream = (t0 > 0 && e[0] != '0') ? 1 : 0;
        // nodes per workgroup: up to ls_chunk on the wide stages (amortises the workgroup's start), fewer on the narrow ones so that
        // a launch has at least ~2 workgroups per CU (512) over the batch's tiles
        dm.ls_chunk = 16;
        if (const char *e = std::getenv("SPOCK_AMD_ILV_CHUNK")) { dm.ls_chunk = std::atoi(e); if (dm.ls_chunk < 1) dm.ls_chunk = 1; }
        dm.n_bslots = dm.nsub;
        // the two bottom stages in one launch (MODE_LBA2) when both are streamed; SPOCK_AMD_ILV_FUSE2=0: three launches (LB, LBA, LA)
        dm.ilv_fuse2 = (dm.ilv_stream && N - 3 >= t0) ? 1 : 0;
        if (const char *e = std::getenv("SPOCK_AMD_ILV_FUSE2")) { if (e[0] == '0') dm.ilv_fuse2 = 0; }
        if (dm.ilv_stream) {
            const long tiles = ((long)batch + 63) / 64;
            int slot = 0;
            for (int t = t0; t <= N - 2; ++t) {
                long c = (long)ipow(d, t) * tiles / 512;
                if (c < 1) c = 1;
                if (c > dm.ls_chunk) c = dm.ls_chunk;
                dm.ls_chunk_t[t] = (int)c;
                dm.ls_slot0[t] = slot;
                if (dm.ilv_fuse2 && t == N - 3) {   // parents per workgroup: the chunk of the bottom stage counted in parents
                    long cb = (long)ipow(d, N - 2) * tiles / 512;
                    if (cb < 1) cb = 1;
                    if (cb > dm.ls_chunk) cb = dm.ls_chunk;
                    c = cb / d < 1 ? 1 : cb / d;
                    dm.ls_chunk_t[t] = (int)c;
                }
                if (dm.ilv_fuse2 && t == N - 2) continue;   // its workgroups are those of stage N-3
                slot += (ipow(d, t) + (int)c - 1) / (int)c;
            }
            dm.n_bslots = slot;
        }
    }

    // ---- offline Riccati recursion (implicit_l.jl:46-97), per stage --------------------------------
    std::vector<Mat> A(d), B(d);
    for (int w = 0; w < d; ++w) {
        A[w] = Mat::from(hp.A.data() + (size_t)w * nx * nx, nx, nx);
        B[w] = Mat::from(hp.B.data() + (size_t)w * nx * nu, nx, nu);
    }
    std::vector<Mat> P(N), K(N - 1), Rinv(N - 1), ABK((size_t)N * d, Mat(nx, nx));
    // Restructured recursion.  The reference computes, per non-leaf node i with children j (implicit_l.jl:584-637),
    //   d_i = Rt^-1 (ubar_i - sum_j B_j' q_j),   q_i = sum_j ABK_j' (P_j B_j d_i + q_j) + K_i' (d_i - ubar_i) - xbar_i,
    // a chain of five dependent mat-vecs per tree level.  Both are linear in (ubar_i, xbar_i, q_j); with
    //   G = sum_j ABK_j' P_j B_j + K',  U = G Rt^-1 - K',  W_j = ABK_j' - G Rt^-1 B_j',  V_j = Rt^-1 B_j'
    // they become  q_i = (U ubar_i - xbar_i) + sum_j W_j q_j  and  d_i = Rt^-1 ubar_i - sum_j V_j q_j:
    // one mat-vec per level on the critical path, the bracketed terms are node-local.
    std::vector<Mat> Um(N - 1), Wm((size_t)(N - 1) * d), Vm((size_t)(N - 1) * d);
    P[N - 1] = Mat::eye(nx);
    for (int t = N - 2; t >= 0; --t) {
        Mat sum_r(nu, nu), sum_k(nu, nx);
        for (int c = 0; c < d; ++c) {
            const int w = (2 + c) % d;
            const Mat BtP = mul(T(B[w]), P[t + 1]);
            sum_r = add(sum_r, mul(BtP, B[w]));
            sum_k = add(sum_k, mul(BtP, A[w]));
        }
        Mat rt(nu, nu);
        for (int j = 0; j < nu; ++j) for (int i = 0; i < nu; ++i) rt(i, j) = (i == j ? 1. : 0.) + 0.5 * (sum_r(i, j) + sum_r(j, i));
        if (!spd_inverse(rt, Rinv[t])) return "Rtilde is not positive definite";
        Mat negk = sum_k;
        for (auto &x : negk.a) x = -x;
        K[t] = mul(Rinv[t], negk);
        Mat sum_p(nx, nx);
        for (int c = 0; c < d; ++c) {
            const int w = (2 + c) % d;
            Mat abk = add(A[w], mul(B[w], K[t]));
            sum_p = add(sum_p, mul(mul(T(abk), P[t + 1]), abk));
            ABK[(size_t)(t + 1) * d + c] = abk;
        }
        P[t] = add(add(Mat::eye(nx), mul(T(K[t]), K[t])), sum_p);
        Mat G = T(K[t]);
        for (int c = 0; c < d; ++c) {
            const int w = (2 + c) % d;
            G = add(G, mul(mul(T(ABK[(size_t)(t + 1) * d + c]), P[t + 1]), B[w]));
        }
        const Mat GR = mul(G, Rinv[t]);
        Mat negKt = T(K[t]);
        for (auto &x : negKt.a) x = -x;
        Um[t] = add(GR, negKt);
        for (int c = 0; c < d; ++c) {
            const int w = (2 + c) % d;
            Mat GRB = mul(GR, T(B[w]));
            for (auto &x : GRB.a) x = -x;
            Wm[(size_t)t * d + c] = add(T(ABK[(size_t)(t + 1) * d + c]), GRB);
            Vm[(size_t)t * d + c] = mul(Rinv[t], T(B[w]));
        }
    }

    // ---- projector onto ker M, M = [E' -I -I ; F' 0 0]: Pi = I - M' (M M')^+ M (cpock.jl:71-81,125-128), one per risk measure
    const int m = ny + 2 * d;
    const int nRm = hp.n_rms > 1 ? hp.n_rms : 1;
    if (nRm > 1 && nRm != dm.nn) return "n_rms must be 1 or the number of non-leaf nodes";
    dm.nRm = nRm;
    std::vector<double> ls_all;
    ls_all.reserve((size_t)nRm * m * m);
    for (int q = 0; q < nRm; ++q) {
        const double *Eq = hp.E.data() + (size_t)q * ny * d, *Fq = hp.F.data() + (size_t)q * ny * d;
        Mat Mk(2 * d, m);
        for (int k = 0; k < d; ++k) {
            for (int i = 0; i < ny; ++i) { Mk(k, i) = Eq[(size_t)i + (size_t)k * ny]; Mk(d + k, i) = Fq[(size_t)i + (size_t)k * ny]; }
            Mk(k, ny + k) = -1.; Mk(k, ny + d + k) = -1.;
        }
        Mat G = mul(Mk, T(Mk)), V;
        std::vector<double> w;
        sym_eig(G, V, w);
        double wmax = 0.;
        for (double x : w) wmax = std::fabs(x) > wmax ? std::fabs(x) : wmax;
        Mat Gp(2 * d, 2 * d);
        for (int k = 0; k < 2 * d; ++k) {
            if (!(w[k] > 1e-12 * wmax)) continue;
            for (int j = 0; j < 2 * d; ++j) for (int i = 0; i < 2 * d; ++i) Gp(i, j) += V(i, k) * V(j, k) / w[k];
        }
        Mat ls = Mat::eye(m), MtGM = mul(mul(T(Mk), Gp), Mk);
        for (size_t k = 0; k < ls.a.size(); ++k) ls.a[k] -= MtGM.a[k];
        for (int j = 0; j < m; ++j) for (int i = 0; i < j; ++i) { const double s2 = 0.5 * (ls(i, j) + ls(j, i)); ls(i, j) = s2; ls(j, i) = s2; }
        ls_all.insert(ls_all.end(), ls.a.begin(), ls.a.end());
    }

    // ---- pack -----------------------------------------------------------------------------------
    auto &pool = out.mats;
    pool.clear();
    auto push = [&](const std::vector<double> &v) { int off = (int)pool.size(); pool.insert(pool.end(), v.begin(), v.end()); return off; };
    dm.mA = (int)pool.size();
    for (int w2 = 0; w2 < d; ++w2) push(A[w2].a);
    dm.mB = (int)pool.size();
    for (int w2 = 0; w2 < d; ++w2) push(B[w2].a);
    dm.mK = (int)pool.size();
    out.K.clear();
    for (int t = 0; t < N - 1; ++t) { push(K[t].a); out.K.insert(out.K.end(), K[t].a.begin(), K[t].a.end()); }
    dm.mRinv = (int)pool.size();
    out.Rinv.clear();
    for (int t = 0; t < N - 1; ++t) { push(Rinv[t].a); out.Rinv.insert(out.Rinv.end(), Rinv[t].a.begin(), Rinv[t].a.end()); }
    dm.mP = (int)pool.size();
    out.P.clear();
    for (int t = 0; t < N; ++t) { push(P[t].a); out.P.insert(out.P.end(), P[t].a.begin(), P[t].a.end()); }
    dm.mABK = (int)pool.size();
    out.ABK.clear();
    for (size_t k = 0; k < ABK.size(); ++k) { push(ABK[k].a); out.ABK.insert(out.ABK.end(), ABK[k].a.begin(), ABK[k].a.end()); }
    dm.mU = (int)pool.size();
    for (int t = 0; t < N - 1; ++t) push(Um[t].a);
    dm.mW = (int)pool.size();
    for (size_t k = 0; k < Wm.size(); ++k) push(Wm[k].a);
    dm.mV = (int)pool.size();
    for (size_t k = 0; k < Vm.size(); ++k) push(Vm[k].a);
    // per-stage "sweep blocks" [W_t,c | V_t,c | ABK_{t+1},c | K_t], contiguous over the stages, so that a workgroup can
    // stage the blocks of its own stages in LDS with one coalesced copy (spock_group.hpp)
    dm.sw_oW = 0; dm.sw_oV = d * nx * nx; dm.sw_oABK = dm.sw_oV + d * nu * nx; dm.sw_oK = dm.sw_oABK + d * nx * nx;
    dm.sw_stride = dm.sw_oK + nu * nx;
    dm.mSweep = (int)pool.size();
    for (int t = 0; t < N - 1; ++t) {
        for (int c = 0; c < d; ++c) push(Wm[(size_t)t * d + c].a);
        for (int c = 0; c < d; ++c) push(Vm[(size_t)t * d + c].a);
        for (int c = 0; c < d; ++c) push(ABK[(size_t)(t + 1) * d + c].a);
        push(K[t].a);
    }
    dm.mLs = push(ls_all);
    out.ls.assign(ls_all.begin(), ls_all.begin() + (size_t)m * m);
    dm.mb = push(hp.b);
    out.sqrtQ.clear(); out.sqrtR.clear(); out.sqrtQN.clear();
    dm.mSqQ = (int)pool.size();
    for (int k = 0; k < hp.nQ; ++k) { Mat s = sym_sqrt(Mat::from(hp.Q.data() + (size_t)k * nx * nx, nx, nx)); push(s.a); out.sqrtQ.insert(out.sqrtQ.end(), s.a.begin(), s.a.end()); }
    dm.mSqR = (int)pool.size();
    for (int k = 0; k < hp.nR; ++k) { Mat s = sym_sqrt(Mat::from(hp.R.data() + (size_t)k * nu * nu, nu, nu)); push(s.a); out.sqrtR.insert(out.sqrtR.end(), s.a.begin(), s.a.end()); }
    dm.mSqQN = (int)pool.size();
    for (int k = 0; k < hp.nQN; ++k) { Mat s = sym_sqrt(Mat::from(hp.QN.data() + (size_t)k * nx * nx, nx, nx)); push(s.a); out.sqrtQN.insert(out.sqrtQN.end(), s.a.begin(), s.a.end()); }
    auto all_diag = [](const std::vector<double> &v, int n) {
        const size_t cnt = v.size() / ((size_t)n * n);
        for (size_t k = 0; k < cnt; ++k)
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) if (i != j && v[k * n * n + (size_t)i + (size_t)j * n] != 0.) return false;
        return true;
    };
    dm.diag_costs = (all_diag(out.sqrtQ, nx) && all_diag(out.sqrtR, nu) && all_diag(out.sqrtQN, nx)) ? 1 : 0;
    // ---- ||L||^2 in closed form.  L acts separately on x (non-leaf: v7 and the children's v3; leaf: v14 and v11), on u
    // (v7, children's v4), on tau (v5, v6), on s_leaf (v12, v13) and on (s_i, y_i) (v1, v2), so L'L is block diagonal:
    //   ||L||^2 = max{ 1 + lmax(sum_ch Q_j), 1 + lmax(QN_i), 1 + lmax(sum_ch R_j), 1/2, 1 + (beta + sqrt(beta^2 + 4 beta))/2 },
    // beta = |b|^2.  The reference leaves this as a `Todo` and hard-codes 10.0 (cpock.jl:62-63) / 3.28 (models/spock.jl:63-64).
    {
        auto lmax = [&](const Mat &S) { Mat V; std::vector<double> w; sym_eig(S, V, w); double m = 0.; for (double x : w) m = x > m ? x : m; return m; };
        double best = 0.5;
        for (int q = 0; q < nRm; ++q) {   // the (s_i, y_i) block of node i: beta = |b_i|^2
            double beta = 0.;
            for (int k = 0; k < ny; ++k) beta += hp.b[(size_t)q * ny + k] * hp.b[(size_t)q * ny + k];
            best = std::fmax(best, 1. + 0.5 * (beta + std::sqrt(beta * beta + 4. * beta)));
        }
        auto Qat = [&](int jn) { return Mat::from(hp.Q.data() + (size_t)(hp.nQ == 1 ? 0 : jn) * nx * nx, nx, nx); };
        auto Rat = [&](int jn) { return Mat::from(hp.R.data() + (size_t)(hp.nR == 1 ? 0 : jn) * nu * nu, nu, nu); };
        const int n_par = (hp.nQ == 1 && hp.nR == 1) ? 1 : nn;     // shared costs: every parent has the same sums
        for (int i = 0; i < n_par; ++i) {
            Mat sq(nx, nx), sr(nu, nu);
            for (int c = 0; c < d; ++c) { sq = add(sq, Qat(i * d + c)); sr = add(sr, Rat(i * d + c)); }
            best = std::fmax(best, 1. + lmax(sq));
            best = std::fmax(best, 1. + lmax(sr));
        }
        for (int i = 0; i < hp.nQN; ++i) best = std::fmax(best, 1. + lmax(Mat::from(hp.QN.data() + (size_t)i * nx * nx, nx, nx)));
        out.L_norm_closed = best;
    }
    for (double x : pool) if (!(x == x)) return "build step produced NaN";
    return "";
}

}  // namespace spock
