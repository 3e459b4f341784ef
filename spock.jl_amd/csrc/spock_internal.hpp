// spock_internal.hpp -- types shared by the host build step, the kernel launchers and the C ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace spock {

// Arithmetic / storage type of the device path.  The product library (libspock_amd.so) is built with double -- "Float64 is baked
// in" in the reference (src/model.jl:304-307).  libspock_amd_f32.so is the same source built with -DSPOCK_REAL=float (and
// single-precision literals): the fp32 variant of BASELINE config 5's tolerance study.  The C ABI is double either way; the
// control blocks, residuals, partial reductions and the host build step stay double.
#ifndef SPOCK_REAL
#define SPOCK_REAL double
#endif
typedef SPOCK_REAL real;
constexpr bool kRealIsDouble = sizeof(real) == 8;

constexpr int kMaxStages = 24;
constexpr int kMaxCones = 8;
// bounds of the run-time-dimension kernel variant (compile-time variants exist for the benchmark shapes)
constexpr int kDynMaxNx = 64, kDynMaxNu = 64, kDynMaxD = 4, kDynMaxNy = 16;

// Problem geometry + data layout, passed BY VALUE to every kernel.
// z layout x|u|s|tau|y: /root/reference/src/model_dynamics/implicit_l.jl:106-158
// v layout v1..v7,v11..v14: /root/reference/src/models/cpock.jl:51-60
struct Dims {
    int N, d, nx, nu, ny;
    int n, nn, nl, nz, nv;
    int zx, zu, zs, zt, zy;
    int o2, o3, o4, o5, o6, o7, o11, o12, o13, o14;
    int nQ, nR, nQN;
    int nRm;                   // 1: one risk measure (b, S2 projector) for all non-leaf nodes, else one per non-leaf node
    int n_cones;
    int cone_kind[kMaxCones];  // already the DUAL cone kinds (MOI.dual_set, implicit_l.jl:841)
    int cone_dim[kMaxCones];
    int risk_has_soc;          // a second-order sub-cone is present (general projection path)
    unsigned long long elem_kind_mask;  // 2 bits per element of y: dual-cone kind (0 zero, 1 >= 0, 2 <= 0, 3 free), ny <= 16
    double x_min, x_max, u_min, u_max;
    // tree tiling: the tree is cut at stage t0 into nsub = d^t0 subtrees of sub_stages = N - t0 stages
    // (one workgroup each) plus the crown (stages 0..t0, one workgroup).  t0 == 0: one workgroup per instance.
    int t0, nsub, sub_stages;
    int first[kMaxStages + 1];  // first node of each stage (scenario_tree.jl:107)
    // offsets (in doubles) into the matrix pool
    int mB, mK, mRinv, mP, mABK, mLs, mb, mSqQ, mSqR, mSqQN;
    // composite matrices of the restructured S1 recursion (spock_host.cpp): per stage t
    //   U_t (nx x nu), per child slot W_{t,c} (nx x nx), V_{t,c} (nu x nx)
    int mU, mW, mV;
    // per-stage sweep blocks (offsets inside one block, block stride, pool offset of stage 0)
    int mSweep, sw_stride, sw_oW, sw_oV, sw_oABK, sw_oK;
    // element-major regime: hierarchy of groups (spock_igroup.hpp): root stages of the layers, top (0) first, bottom (t0) last
    int n_lv, lv_root[8], n_crown_groups;
    // element-major regime, level-streaming kernels (spock_ilevel.hpp) for the stages t0 .. N-2: one launch per tree stage, a
    // workgroup works on ls_chunk consecutive nodes of the stage and owns one slot of residual partials
    int ilv_stream;              // 1: stages >= t0 run as level kernels (LB / LBA / LA) instead of the bottom-group kernels
    int ilv_fuse2;               // 1: stages N-3 and N-2 run as ONE launch (MODE_LBA2); ls_chunk_t[N-3] = parents per workgroup, its slots cover both
    int ls_chunk;                // nodes of a stage per workgroup: upper bound (wide stages)
    int ls_chunk_t[kMaxStages];  // ... per stage: narrow stages take fewer nodes per workgroup so that the launch still fills the chip
    int ls_slot0[kMaxStages];    // first partial slot of every streamed stage
    int n_bslots;                // partial slots of the bottom part (old kernels: nsub); the crown groups' slots follow
    int mA;          // A_w, d matrices (closed-loop state update of the MPC loop)
    int diag_costs;  // every sqrtQ / sqrtR / sqrtQN is diagonal -> element-wise products in the kernels
};

// Per-instance control block in device memory: the driver's scalars live here so that a captured
// hipGraph can be replayed for any tol / gamma / sigma and so that kernels can stop on their own.
struct Ctl {
    int done;           // 1: every later kernel of this solve exits at once
    int status;         // SPOCK_CONVERGED / SPOCK_MAX_ITER / SPOCK_NAN
    int evals;          // termination tests run so far (= T evaluations checked)
    int iter;           // the reference's `iter` (cp.jl:203,227)
    int pending;        // an iteration has been issued whose termination test has not run yet
    int result_parity;  // which buffers hold the final iterate: z buffer result_parity % 3, v buffer result_parity % 2
    int max_iter;
    int log_mode;       // SPOCK_LOG -> absolute xi rule (cp.jl:83-99), else res_0 rule (cp.jl:102-112)
    int alg;
    int pad;
    double res0[2];     // cp.jl:114-119
    double xi1, xi2, xi;
    double gamma, sigma, tol;
};

// per workgroup: xi1, xi2, xi, |L'd_v - d_z/gamma| | SP: sum_z dz xi_1, sum_v dv xi_2, <r, P d>, |L d_z - d_v/sigma|
constexpr int kPartialStride = 8;
constexpr int kHistStride = 4;     // xi, xi1, xi2, backtrack

enum KFlags : unsigned {
    F_RAW_LT = 1u,       // phase A: write z - gamma L'v for every owned entry, no S2 / s_1 shift (operator L')
    F_NO_SWEEP = 2u,     // skip the S1 sweeps (phase B then reads x, u from z_out)
    F_NO_BLOCAL = 4u,    // skip phase B's node-local part (operator prox_f)
    F_W_IS_ZBAR = 8u,    // phase B: w = zbar instead of 2 zbar - z (operator L)
    F_NO_PROX = 16u,     // phase B: v_out = sigma L w + v, no prox (operator L)
    F_NO_CTL = 32u,      // ignore the control block's done / pending logic (operators)
    F_NO_ALOCAL = 64u,   // skip phase A's node-local part (operator prox_h_conj)
    F_DBG_NO_BWD = 128u, // measurement only (spock_profile_kernels + SPOCK_AMD_DEBUG_FLAGS): skip the backward sweep
    F_DBG_NO_FWD = 256u, // measurement only: skip the forward sweep
    F_SP = 1024u,        // SP driver: no CP termination prologue in the crown / whole-tree kernel, only the done check
    F_DBG_STAMPS = 512u, // measurement only: thread 0 of every workgroup of instance 0 writes clock stamps to KArgs::dbg
};

struct KArgs {
    Dims dm;
    const real *mats;           // the matrix pool; instance b reads mats + b * mats_stride (0: one pool shared by the batch)
    long mats_stride;
    const real *z_in, *v_in;  // input iterate; instance b starts at z_in + b*stride_z, v_in + b*stride_v
    real *z_out, *v_out;      // zbar, vbar (same strides)
    real *z_next;             // SUB_BA only: where phase A of the NEXT iteration puts its part of the next zbar (third z buffer)
    const real *dir;          // SP, sp_mode 2: the direction d (z | v concatenated per instance like z_in), for <r, P d>
    int sp_mode;                // SP: 0 none, 1 accumulate <r, P r>, 2 also <r, P d>  (spock_device.hpp: b_local SPM)
    real *v2x;                // SUB_BA only: [batch][nsub] vbar.v2 of the subtree roots as recomputed by the crown's workgroup
    long stride_z, stride_v;    // CP: nz, nv (separate arrays); SP: nz + nv for both (z|v concatenated per instance)
    real *dbuf;               // [batch][nn*nu]      S1's d_i between phase A and phase B
    real *qroot;              // [batch][nsub*nx]    S1's q at the subtree roots
    real *cbuf;               // [batch][crown non-leaf nodes][nx+nu]  node-local terms of the crown's sweep
    double *partials;           // [batch][(nsub+1)*kPartialStride]
    Ctl *ctl;                   // [batch]
    const real *x0;           // [batch][nx]
    double *hist;               // [batch][hist_cap*kHistStride]
    unsigned long long *dbg;    // [3 kernels][64 stamps][2]: (s_memtime, s_memrealtime) pairs, diagnostic builds of the timing only
    int hist_cap;
    int parity;
    unsigned flags;
    int batch;
    int lds_mats;               // 1: the sweep matrices of the workgroup's stages are staged in LDS
    int ex_doubles;             // size of the LDS exchange region (thread path: threads*nx; team path: TeamLds vectors)
    int team_reg;               // lane-team sweep with the matrix rows in registers (small compile-time shapes)
    int team;                   // > 0: lane-team sweeps with this many lanes per node (all sweep vectors in LDS)
    real *xlbuf;              // [batch][groups][xl_stride]: the group-leaf sweep vector when it does not fit in LDS
    int xl_global, xl_stride;
    // element-major ("interleaved") batched regime (spock_igroup.hpp): every per-instance array is [element][bp], bp = batch
    // rounded up to a multiple of 64; one wave = one tree node x one tile of 64 instances
    int ilv;
    long bp;
    real *cq;                 // [crown non-leaf nodes][nx][bp]: q / x exchange of the crown's sweep (global memory)
    int cchunk, n_cchunks;      // crown nodes per workgroup of the node-local crown phases, number of such workgroups
    int want_xi;                // element-major regime: LOG-mode CP (combined residual xi, cp.jl:83-86) selects the kernel variant
    int grp_level;              // element-major regime, I_UP / I_DOWN: the layer of mid groups this launch works on
    int tile0, n_tiles;         // element-major regime: the launch covers tiles tile0 .. tile0 + n_tiles - 1 (n_tiles 0 = all)
    int ls_cnt, ls_first, ls_first_child, ls_slot, ls_chunk_k;   // level kernels (grp_level = the stage): filled by the launcher from IlvGeo
    // flat node-local kernels (spock_flat_local.hpp): 1 = SUB_A / CROWN / SUB_B run the sweeps only (c_i, e_i, q_leaf come from
    // zbar's x / u entries, x_i, u_i go back there), the node-local phases run as MODE_FLAT_A / MODE_FLAT_B over `flat_chunks`
    // workgroups per instance, which also own the residual partial slots
    int flat_local, flat_chunks;
};

// ---- SP (SuperMann + Anderson) driver: plain data shared by the kernels (spock_sp.hpp) and the driver ----
struct SpCtl {          // per instance, device memory; survives between solves where the reference's state does
    int k;              // Anderson calls in this solve (anderson!'s `iter`), reset per solve
    int head;           // physical column (0..2) holding the newest logical column of MP / Q
    int accept;         // decision of the current iteration: 1 K1, 2 K2, 0 plain CP step
    int backtrack;      // 0 or 9
    double R[9];        // aa_R, column-major 3x3 upper triangular; persists between solves
    double r_safe, rnorm, rtilde, rho_dot, coef;
    double r2, r3, rho, g[3];
    double xi_fb[2];    // residuals of the plain CP step z <- T(z) (the fall-back), kept from the first T evaluation
    double crho[3];     // norm of each PHYSICAL column of Q: the columns are stored unnormalised and divided where they are read
                        // (spock_sp.hpp: FLAT1 / FLAT2); 1 initially
};

constexpr int kSpSums = 10;   // partial sums per workgroup of the flat / LOP kernels

struct SpArgs {
    Dims dm;
    const real *mats;
    long mats_stride;          // as in KArgs
    Ctl *ctl;
    SpCtl *sp;
    // every vector is [batch][nr], nr = nz + nv, z part first
    real *Z, *ZB, *ZOLD, *R, *ROLD, *DP, *W, *WB, *RW, *D;
    real *MP, *Q;            // [batch][3][nr], circular columns
    double *partF;             // [batch][n_partF][kSpSums] partial sums of the flat kernels
    double *partL;             // [batch][n_partL][kSpSums] partial sums (0,1) and maxima (2,3) of the LOP kernels
    double *hist;
    int hist_cap;
    int n_partF, n_partL;      // workgroups per instance of the kernels that fill partF / partL
    int batch;
    long nr;
    int op;
    int team;                  // lanes per node of the lane-team LOP (large nx), else 0
    // fused inner products: <r, P r>, <rt, P rt>, <rt, P d> come out of the CP kernels' phase B (KArgs::sp_mode) as
    // per-workgroup partial sums in partG[instance][n_partG][kPartialStride] (slots 4, 5, 6) instead of out of LOP kernels
    const double *partG;
    int n_partG, fused_lop;
    // corrected variant (spock_options_t.sp_variant = 1): least-squares Anderson, working line search (spock_sp.hpp)
    int variant;
    int trial, last_trial;     // line search: index of the candidate this launch belongs to, 1 if no further candidate follows
    double tau;                // its step: beta^trial
    // closed loop chained on the device (SP_CTL_M; examples/server_heat/mpc_simulation.jl:47-54,164-177): per instance the step it is at,
    // its disturbance sequence, x0 (advanced in place), the trajectories and one record per step [status, evals, iter, xi1, xi2, xi]
    const int *mpc_w;
    real *mpc_x0, *mpc_xt, *mpc_ut;
    double *mpc_res;
    int *mpc_t;
    int mpc_steps;
};
constexpr int kMpcRec = 6;

enum SpOp { SP_FLAT1 = 1, SP_FLAT2, SP_FLAT3, SP_FLAT4, SP_FLAT5, SP_LOP_R, SP_LOP_RT, SP_LOP_DP,
            SP_CTL_A, SP_CTL_B, SP_CTL_C, SP_CTL_D, SP_CTL_CD,
            SP_FLATW = 20,    // corrected variant: the next line-search candidate W = Z + tau D
            SP_CTL_E = 21,    // corrected variant: end of an iteration, instances parked by an accepted candidate (done == 3) run on
            SP_CTL_M = 22 };  // closed loop on the device, end of an iteration: an instance whose solve has just ended (done == 2, final update
                              // applied) records the step, applies u, advances x0 and re-arms itself for the next warm-started solve

// MODE_SUB_B_SP / MODE_FULL_SP: SUB_B / FULL whose phase B also accumulates the SP driver's inner products (KArgs::sp_mode);
// separate kernels so that the CP kernels do not carry that code and its registers
enum GroupMode { MODE_FULL = 0, MODE_SUB_A = 1, MODE_CROWN = 2, MODE_SUB_B = 3, MODE_FINALIZE = 4, MODE_SUB_BA = 5,
                 MODE_SUB_B_SP = 6, MODE_FULL_SP = 7,
                 MODE_SUB_B_LAT = 8,   // SUB_B of the single-instance regime: the leaf-children blocks of phase B on an idle wave
                 N_GROUP_MODES = 9,
                 // element-major regime only (spock_igroup.hpp): the node-local phases of the crown's nodes, chunked over workgroups
                 MODE_ICROWN_A = 9, MODE_I_UP = 10, MODE_I_DOWN = 11,
                 // element-major regime, level-streaming kernels (spock_ilevel.hpp); KArgs::grp_level = the tree stage of the launch
                 MODE_LB = 12,    // forward step + phase B of one non-bottom stage (top-down)
                 MODE_LBA = 13,   // bottom stage (children = leaves): forward step, phase B, next phase A, backward step
                 MODE_LA = 14,    // next phase A + backward step of one non-bottom stage (bottom-up), on (zbar, vbar) -> znext
                 MODE_LA0 = 15,   // prologue: phase A + backward step of iteration 0 on (z, v) -> zbar, any streamed stage
                 // flat node-local kernels of the lane-team shapes (large nx / nu; spock_flat_local.hpp): the node-local phases leave the
                 // one-workgroup-per-subtree kernels and run over chunks of consecutive non-leaf nodes, ~1000 workgroups per instance
                 MODE_FLAT_A = 16,     // phase A of every non-leaf node: c_i, e_i and the leaves' q = -xbar parked in zbar's x / u entries
                 MODE_FLAT_B = 17,     // phase B of every non-leaf node on the zbar the sweeps completed; residual maxima per workgroup
                 MODE_FLAT_B_SP = 18,  // ... plus the SP driver's inner products (KArgs::sp_mode)
                 MODE_FLAT_BA = 19,    // phase B of iteration k and phase A of iteration k + 1 on the iterate just produced, one launch
                 // element-major regime: the two bottom stages in one launch (spock_ilevel.hpp), grp_level = N-3: LB of stage N-3, LBA of stage
                 // N-2 and LA of stage N-3 per parent -- what the three launches hand to one another stays in registers / LDS
                 MODE_LBA2 = 20,
                 // single-instance regime: MODE_SUB_BA with the node-local phases split over role lanes (spock_rgroup.hpp): wave = role, lane = node
                 MODE_SUB_BA_R = 21,
                 N_ALL_MODES = 22 };

// ---- host build step (spock_host.cpp) -------------------------------------------------------------
struct HostProblem {
    int N, d, nx, nu, ny;
    std::vector<double> A, B, Q, R, QN, E, F, b;
    int nQ, nR, nQN, n_rms = 1;
    std::vector<int> cone_kind, cone_dim;
    double x_min, x_max, u_min, u_max;
};
struct BuildProducts {
    Dims dm;
    std::vector<double> mats;  // the matrix pool uploaded to the device
    // copies kept for spock_get_build_product
    std::vector<double> P, K, Rinv, ABK, ls, sqrtQ, sqrtR, sqrtQN;
    double L_norm_closed = 0.;   // ||L||^2 in closed form; the reference hard-codes 10.0 / 3.28
};
// returns "" on success, else an error message; split_stage 0 = choose
std::string host_build(const HostProblem &hp, int split_stage, BuildProducts &out, bool &unsupported, int batch = 1);
int choose_split_stage(int N, int d, int nx, int nu = 0, int batch = 1);
// launch geometry shared by the GPU launcher and the emulation harness
// xl_global: the lane-team sweep's vector over the GROUP LEAVES (q in, x out) lives in global memory instead of LDS
// (xl_stride doubles per workgroup): large nx with deep groups, where all four sweep vectors do not fit 144 KiB
struct LaunchGeo { int threads, grid_x, team, team_reg, team_local, lds_mats, ex_doubles, xl_global, xl_stride; size_t lds_bytes; };
bool team_local_applies(const Dims &dm);   // lane-team node-local phases: large nx / nu, shared diagonal costs, no SOC risk cone
LaunchGeo launch_geometry(int mode, const Dims &dm, int batch);
// flat node-local kernels: lanes per node, threads per workgroup (= one round of threads / team nodes), workgroups per instance
struct FlatGeo { int team, threads, nteams, chunks, ex_doubles; size_t lds_bytes; };
FlatGeo flat_geometry(const Dims &dm);
bool flat_local_applies(const Dims &dm, int batch);   // the handle runs its node-local phases as flat kernels
bool sweep_kernels_apply(const Dims &dm);             // dedicated matrix-core sweep kernels exist for this shape (spock_skernels.hip; false in the host emulation)
int group_threads(int d, int stages);  // non-leaf nodes of a complete d-ary tree with `stages` stages

// element-major batched regime: split stage (subtrees of at most `max_sub_stages` stages, LDS-bound), launch geometry
int choose_split_stage_ilv(int N, int d, int nx);
struct IlvGeo {
    int threads, grid_x, cchunk, n_cchunks; size_t lds_bytes; int ex_doubles;
    int ls_cnt = 0, ls_first = 0, ls_first_child = 0, ls_slot = 0, ls_chunk_k = 1;   // level kernels: nodes of the stage, its first node, first node of the next stage, first partial slot
};
IlvGeo ilv_geometry(int mode, const Dims &dm, int level = 0, int want_xi = 0);
// level-streaming kernels (spock_ilevel.hpp): role waves per workgroup, rows (64 doubles each) of ONE LDS exchange buffer
// (the kernels double-buffer it over the nodes of a chunk).  LBA rows: per child, role C_c [L'dv | L'vbar | L'xi_2 on (x_i, u_i) | tau_c],
// role L_c [W_c q_c | V_c q_c | s_c];  LB rows: per child, role C_c [L'dv | L'vbar | L'xi_2 on (x_i, u_i)]
int ilevel_waves(int mode, const Dims &dm);
// (the L'xi_2 rows exist in the LOG-mode kernel variants only: want_xi)
constexpr int lba_exchange_rows(int nx, int nu, int d, bool wx) { return d * ((wx ? 3 : 2) * (nx + nu) + 1) + d * (nx + nu + 1); }
constexpr int lb_exchange_rows(int nx, int nu, int d, bool wx) { return d * (wx ? 3 : 2) * (nx + nu); }
int ilevel_ex_doubles(int mode, const Dims &dm, int want_xi);
// role-lane subtree kernel (spock_rgroup.hpp), LDS doubles: [sweep vectors XQ 64 nx | ED 64 nu | UL 64 nu | XL 64 d nx][given rows (nx + nu + d nx) 64]
// [role exchange rows 64][s entries 64]; threads per workgroup: one wave per role, and T = 8 lanes per node for the lane-team sweeps
constexpr int rgroup_lds_doubles(int nx, int nu, int d, bool wx) {
    return 64 * nx + 2 * 64 * nu + 64 * d * nx + (nx + nu + d * nx) * 64 + lba_exchange_rows(nx, nu, d, wx) * 64 + 64
           + 7 * (2 * d * nx * nx + d * nu * nx + nu * nx);   // + the sweep blocks [W | V | ABK | K] of up to seven stages (groups of <= 64 nodes)
}
constexpr int kRgroupThreads = 512;

// ---- kernel launchers (spock_kernels.hip) ---------------------------------------------------------
// Each returns a hipError_t cast to int.  `stream` is a hipStream_t.
int launch_group(int mode, const KArgs &a, void *stream);
int launch_igroup(int mode, const KArgs &a, void *stream);   // element-major batched regime (a.ilv), spock_ikernels.hip
int launch_rgroup(const KArgs &a, void *stream);             // MODE_SUB_BA_R (spock_rkernels.hip)
bool rlane_applies(const Dims &dm, int batch, bool fused);   // the handle runs MODE_SUB_BA_R in place of MODE_SUB_BA
bool have_static_variant(int nx, int nu, int d, int ny);
// closed-loop step of examples/server_heat/mpc_simulation.jl:164-177 on the device
struct MpcArgs {
    Dims dm;
    const real *mats;
    long mats_stride;
    const real *z;        // element e of instance b at z[b * stride_z + e * stride_e] (the u block of node 0 is the applied input)
    long stride_z, stride_e;
    real *x0;             // [batch][nx], updated in place: x0 <- A_w x0 + B_w u
    const int *w_seq;       // [batch][steps] events
    real *x_traj, *u_traj;  // [batch][steps+1][nx], [batch][steps][nu]
    int t, steps, batch;
};
int launch_mpc_advance(const MpcArgs &a, void *stream);
// dst[c * ld_dst + r] = src[r * ld_src + c] for r < rows, c < cols (instance-major <-> element-major copies of the state)
int launch_transpose(real *dst, const real *src, long rows, long cols, long ld_src, long ld_dst, void *stream);
// element-major state: column b of z0 / v0 <- column b of the buffer its control block's result_parity names
int launch_ilv_collect(real *const *z3, real *const *v2, const Ctl *ctl, long nz, long nv, long bp, int batch, void *stream);
int launch_sp(const SpArgs &a, void *stream);   // a.op selects the kernel (spock_sp.hpp)
// on-device build step for new dynamics (spock_build.hpp); workspace size in doubles
struct BuildArgs {
    Dims dm;
    real *mats;          // the matrix pool (device)
    const double *A;     // [d][nx*nx] new dynamics (device)
    const double *B;     // [d][nx*nu]
    double *ws;          // workspace, build_ws_doubles(dm) doubles
    int *status;         // 0 ok, 1 Rtilde not positive definite
    // batched builds (spock_update_dynamics_batch): workgroup b builds pool mats + b * mats_stride from A + b * d nx^2, B + b * d nx nu
    // in workspace ws + b * build_ws_doubles(dm), status[b]
    long mats_stride;
    int n_inst;
};

int launch_build_riccati(const BuildArgs &a, void *stream);
inline long build_ws_doubles(const Dims &dm) { return 7L * dm.nx * dm.nx + 6L * dm.nu * dm.nx + 4L * dm.nu * dm.nu + (long)dm.d * dm.nx * dm.nx; }
int sp_flat_blocks(long nr);                    // workgroups per instance of the flat kernels
int sp_lop_blocks(const Dims &dm);              // workgroups per instance of the LOP kernels

}  // namespace spock
