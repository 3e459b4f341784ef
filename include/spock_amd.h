/*
 * spock_amd.h -- C ABI of libspock_amd.so: the MI355X (gfx950) implementation of the one hot path of
 * kul-optec/spock.jl, the Chambolle-Pock (CP) / SuperMann-accelerated CP (SP, "Spock") iteration for
 * multistage risk-averse optimal control over a scenario tree.
 *
 * The reference (pure Julia, /root/reference) has no FFI: its seam for this path is Julia multiple
 * dispatch on MODEL_IMPLICITL / MODEL_CP / MODEL_SP (src/model.jl:261-267).  Each entry point below
 * names the reference function it replaces.  A Julia `ccall` binding and the Python ctypes binding that
 * the tests and the benchmark use are shown in INTEGRATION.md.
 *
 * Conventions: extern "C"; plain pointers and sizes; every pointer is a HOST pointer owned by the
 * caller (the library copies what it needs into device memory owned by the handle); matrices are
 * column-major (Julia layout); indices are 0-based; no exceptions cross the boundary -- every call
 * returns a status code and spock_last_error() holds the message of the last failure on this thread.
 * A handle is bound to one GPU and one HIP stream and must not be used from two threads at once.
 * All arithmetic is fp64 ("Float64 is baked in" in the reference: src/model.jl:304-307).
 */
#ifndef SPOCK_AMD_H
#define SPOCK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPOCK_AMD_VERSION 300 /* round 3: spock_abi_struct_sizes, spock_sizes_t.layout_reason, element-major SP */

/* status codes */
#define SPOCK_OK 0
#define SPOCK_ERR_BAD_ARG 1      /* NULL / wrong length / wrong value ("z or v has the wrong length": implicit_l.jl:184-186) */
#define SPOCK_ERR_UNSUPPORTED 2  /* valid in the reference, not (yet) on this path (model.jl:295) */
#define SPOCK_ERR_HIP 3          /* HIP runtime failure, no GPU, kernel image missing */
#define SPOCK_ERR_BUILD 4        /* build step failed (Rtilde not positive definite, ...) */

/* solve status (spock_result_t.status) */
#define SPOCK_CONVERGED 0
#define SPOCK_MAX_ITER 1 /* the reference just falls out of its loop (cp.jl:205, sp.jl:391) */
#define SPOCK_NAN 2      /* a residual became NaN */

/* AlgorithmOptions (src/model.jl:61-64) */
#define SPOCK_ALG_CP 1
#define SPOCK_ALG_SP 2

/* VERBOSE_LEVEL (src/model.jl:251-254): selects the termination rule (cp.jl:83-112, sp.jl:304-334) */
#define SPOCK_SILENT 1
#define SPOCK_LOG 2

/* ConvexBaseCone kinds (src/risk_measures.jl:6) */
#define SPOCK_CONE_ZEROS 0
#define SPOCK_CONE_NONNEG 1
#define SPOCK_CONE_NONPOS 2
#define SPOCK_CONE_REALS 3
#define SPOCK_CONE_SOC 4

typedef struct spock_handle_s *spock_handle_t;

/* The five positional arguments of build_model (src/model.jl:273-279) as flat read-only arrays.
 *   scen_tree   -> N, d (+ optional anc, w to be validated against the uniform breadth-first layout of
 *                  generate_scenario_tree_uniform_branching_factor_v2, src/scenario_tree.jl:61-109)
 *   dynamics    -> A, B indexed by event           (src/dynamics.jl:1-4)
 *   cost        -> Q, R per non-root node, QN per leaf; a count of 1 means "one matrix shared by all"
 *                  (src/cost.jl:3-7)
 *   rms         -> E, F, b and the cone list of K (src/risk_measures.jl:79-84): one risk measure shared by all non-leaf nodes
 *                  (n_rms = 1: get_uniform_rms_v2, :93-102) or one per non-leaf node (n_rms = n_nonleaf:
 *                  get_nonuniform_rms_avar_v2, :127-136); length(b) and K are common to all (implicit_l.jl:14,19)
 *   constraints -> UniformRectangle bounds          (src/constraints.jl:11-22)
 */
typedef struct {
    int32_t N;            /* number of stages INCLUDING the root: leaves = d^(N-1) */
    int32_t d;            /* branching factor (> 1) */
    int32_t nx, nu, ny;   /* state, input, length(b) */
    const int32_t *anc;   /* optional [n]: parent of each node, -1 for the root; NULL = uniform tree */
    const int32_t *w;     /* optional [n]: event of each node, -1 for the root;  NULL = uniform tree */
    const double *A;      /* [d][nx*nx] */
    const double *B;      /* [d][nx*nu] */
    int32_t nQ, nR, nQN;  /* 1, or n-1 / n-1 / n_leaf */
    const double *Q;      /* [nQ][nx*nx] */
    const double *R;      /* [nR][nu*nu] */
    const double *QN;     /* [nQN][nx*nx] */
    const double *E;      /* [n_rms][ny*d]  (ny x d) */
    const double *F;      /* [n_rms][ny*d]  (ny x d) */
    const double *b;      /* [n_rms][ny] */
    int32_t n_cones;      /* subcones of K, dimensions add up to ny */
    const int32_t *cone_kind;
    const int32_t *cone_dim;
    double x_min, x_max, u_min, u_max;
    int32_t n_rms;        /* 0 / 1: E, F, b hold one risk measure; n_nonleaf: one (ny x d), (ny x d), (ny) block per non-leaf node */
} spock_problem_t;

/* SolverOptions (src/model.jl:71-74) plus what a GPU needs to know.  Zero / NaN selects the default. */
typedef struct {
    int32_t algorithm;    /* SPOCK_ALG_CP | SPOCK_ALG_SP */
    int32_t device;       /* HIP device ordinal */
    int32_t batch;        /* independent instances (same problem data, own x0 and state) held by the handle; 0 -> 1 */
    int32_t split_stage;  /* tree stage at which the tree is cut into per-workgroup subtrees; 0 = choose */
    double L_norm;        /* NaN -> the reference's hard-coded value: 10.0 (CP, cpock.jl:63) / 3.28 (SP, models/spock.jl:64);
                           * negative -> ||L||^2 in closed form (the reference's `Todo: L norm must be computed efficiently!`) */
    int32_t max_iter;     /* 0 -> reference default: 5000 (cp.jl:190) / 1000 (sp.jl:360) */
    int32_t graph_iters;  /* iterations captured per hipGraph replay; 0 = choose */
    int32_t batch_layout; /* layout of the batch in HBM: 0 = choose (element-major from 320 instances on, CP and the compile-time
                           * shapes), 1 = instance-major z[b][nz] (one thread per tree node), 2 = element-major z[e][batch]
                           * (one lane per instance, one wave per tree node).  Invisible at this boundary: every array that
                           * crosses it is instance-major. */
    int32_t sp_variant;   /* SP only.  0: run_sp! / anderson! exactly as the reference has them, defects included (no-op line search
                           * sp.jl:439, coefficients written into row 1 of R anderson.jl:86); 1: the corrected variant -- least-squares
                           * Anderson and a line search that shortens the step (no reference counterpart; measures what the defects cost) */
    int32_t sp_max_backtrack; /* sp_variant 1: candidates tried after the first, 0 -> run_sp!'s MAX_BACKTRACK_COUNT = 8 (sp.jl:371) */
    int32_t per_instance_dynamics; /* 1: every instance of the batch gets a matrix pool of its own, so that spock_update_dynamics_batch can
                           * give each its own A_w, B_w (Monte-Carlo draws of the dynamics; instance-major layout only).  0: one pool. */
} spock_options_t;

typedef struct {
    int32_t status;       /* SPOCK_CONVERGED | SPOCK_MAX_ITER | SPOCK_NAN */
    int32_t iterations;   /* evaluations of the CP operator whose termination test ran */
    int32_t iter_ref;     /* what the reference prints: "Finished in $(iter) iterations." (cp.jl:230) */
    int32_t reserved;
    double xi1, xi2, xi;  /* last residuals (xi is NaN in SILENT mode) */
    double solve_ms;      /* device time of the whole solve call (HIP events on the handle's stream) */
} spock_result_t;

typedef struct {
    int32_t n, n_nonleaf, n_leaf, nz, nv, batch;
    /* 0-based offsets of the z blocks x|u|s|tau|y (implicit_l.jl:106-158) and of the v blocks (cpock.jl:51-60) */
    int32_t z_x, z_u, z_s, z_tau, z_y;
    int32_t v1, v2, v3, v4, v5, v6, v7, v11, v12, v13, v14;
    int32_t split_stage, n_subtrees, graph_iters;
    int32_t fused_iteration;  /* 1: a CP iteration runs as CROWN + SUB_BA (a deterministic rule on the shape), 0: SUB_A + CROWN + SUB_B / whole-tree kernel */
    int32_t interleaved;      /* 1: the batch is held element-major (spock_options_t.batch_layout) */
    int32_t layout_reason;    /* why `interleaved` is what it is: 0 requested / chosen by the batch-size rule; 1 element-major does not exist for
                               * this algorithm / shape / per-node data; 2 fell back to instance-major: an array would pass the 4 GiB of the
                               * kernels' 32-bit buffer offsets (batch too large); 3 fell back: subtree does not fit the LDS slots */
} spock_sizes_t;

/* build_model(scen_tree, cost, dynamics, rms, constraints, SolverOptions(alg, nothing))
 * (src/model.jl:273-297 -> src/models/cpock.jl:1-163 / src/models/spock.jl:1-210).  Runs the build step
 * (ricatti_offline implicit_l.jl:46-97, kernel projector cpock.jl:71-81,125-128, sqrt of the costs
 * cpock.jl:119-121) on the host, uploads it, allocates the solver state (z = v = 0, ...). */
int spock_create(const spock_problem_t *problem, const spock_options_t *options, spock_handle_t *out);
int spock_destroy(spock_handle_t h);
int spock_sizes(spock_handle_t h, spock_sizes_t *out);

/* solve_model!(model, x0; tol, verbose, z0, v0, gamma, sigma) (cpock.jl:169-189, models/spock.jl:231-251).
 * x0: [batch][nx].  z0/v0: both NULL (warm start from the state left in the handle) or both given,
 * [batch][nz] / [batch][nv].  gamma/sigma: both finite to override 0.99/sqrt(L_norm) (cp.jl:198-201), else NaN.
 * results: [batch].  Blocking. */
int spock_solve(spock_handle_t h, const double *x0, double tol, int32_t verbose,
                const double *z0, const double *v0, double gamma, double sigma, spock_result_t *results);

/* Closed-loop MPC simulation, the Spock branch of examples/server_heat/mpc_simulation.jl:47-54,164-177: `steps`
 * warm-started solves; after each one u = z[u_inds[1:nu]] is applied and x0 <- A_w x0 + B_w u is evaluated on the
 * device (no state leaves the GPU between steps).  w_seq: [batch][steps] events in [0, d) (the reference draws
 * rand(1:d)).  x_traj: [batch][steps+1][nx], u_traj: [batch][steps][nu], results: [batch][steps]; any may be NULL. */
int spock_mpc_simulate(spock_handle_t h, const double *x0, int32_t steps, const int32_t *w_seq, double tol, int32_t verbose,
                       double *x_traj, double *u_traj, spock_result_t *results);

/* model.solver_state.z / .v after a solve (mpc_simulation.jl:50-54): copy-out, [batch][nz] / [batch][nv] */
int spock_get_z(spock_handle_t h, double *z_out);
int spock_get_v(spock_handle_t h, double *v_out);
/* overwrite the persistent state of every instance (z, v); also resets what SP carries between calls */
int spock_set_state(spock_handle_t h, const double *z, const double *v);
/* back to the freshly built model: z = v = 0 and every carried buffer at its initial value */
int spock_reset(spock_handle_t h);
/* per-iteration log of the last solve for instance `inst` (what verbose=LOG appends to
 * examples/output/xi*.dat, cp.jl:88-96, sp.jl:306-317): xi, xi1, xi2 and (SP) the backtrack count.
 * Any output pointer may be NULL.  *len receives the number of entries (<= cap are written). */
int spock_get_history(spock_handle_t h, int32_t inst, double *xi, double *xi1, double *xi2, int32_t *backtrack,
                      int32_t cap, int32_t *len);

/* Operator-level seam (what cp.jl / sp.jl call), one instance, host in -> device kernels -> host out.
 * They run the same device code as the fused solve path and exist for parity testing. */
int spock_op_L(spock_handle_t h, const double *z, double *v_out);                 /* L!            implicit_l.jl:177-318 */
int spock_op_Lt(spock_handle_t h, const double *v, double *z_out);                /* L_transpose!  implicit_l.jl:320-449 */
int spock_op_prox_f(spock_handle_t h, const double *x0, double gamma, double *arg);      /* prox_f!      implicit_l.jl:729-748 (in place) */
int spock_op_prox_h_conj(spock_handle_t h, double sigma, double *arg);            /* prox_h_conj!  implicit_l.jl:911-951 (in place) */
int spock_op_dot(spock_handle_t h, const double *a_z, const double *a_v, const double *b_z, const double *b_v,
                 double gamma, double sigma, double *out);                        /* spock_dot     implicit_l.jl:485-516 */
/* one evaluation of the CP operator (update_zbar! + update_vbar!, cp.jl:5-32) */
int spock_op_T(spock_handle_t h, const double *x0, const double *z, const double *v, double gamma, double sigma,
               double *zbar_out, double *vbar_out);

/* products of the build step, for cross-checking against the oracle: which = 0 P, 1 K, 2 Rtilde^-1,
 * 3 ABK, 4 kernel projector, 5 sqrtQ, 6 sqrtR, 7 sqrtQN.  *len receives the element count. */
int spock_get_build_product(spock_handle_t h, int32_t which, double *out, int32_t cap, int32_t *len);

/* The dynamics-dependent part of the build step ON THE DEVICE: ricatti_offline (src/model_dynamics/implicit_l.jl:46-97: P, K,
 * Rtilde^-1, A + B K per stage) and the composite matrices of the S1 sweep, recomputed in the handle's matrix pool for new dynamics
 * of the same shape -- A: [d][nx*nx], B: [d][nx*nu], host pointers.  Costs, risk measures, constraints, state and step sizes are
 * kept.  The reference rebuilds the whole model (build_model) when the dynamics change; this replaces that for e.g. a model
 * re-linearised between the steps of a closed loop.  SPOCK_ERR_BUILD when Rtilde is not positive definite. */
int spock_update_dynamics(spock_handle_t h, const double *A, const double *B);
/* The same for a BATCH OF DRAWS: instance b of the handle gets A[b] ([d][nx*nx]) and B[b] ([d][nx*nu]); the device builds all of them
 * side by side (one workgroup each).  Needs spock_options_t.per_instance_dynamics.  status: optional [batch], SPOCK_OK / SPOCK_ERR_BUILD per
 * instance; an instance whose build fails keeps its previous dynamics, the call then returns SPOCK_ERR_BUILD.  The reference builds one model
 * per draw (build_model, src/model.jl:273-297). */
int spock_update_dynamics_batch(spock_handle_t h, const double *A, const double *B, int32_t *status);

/* L_norm in use (step sizes default to 0.99/sqrt(L_norm), cp.jl:198-201) and the closed-form ||L||^2 of this problem */
int spock_L_norm(spock_handle_t h, double *in_use, double *closed_form);
/* device time of the last solve divided by its iteration count, and that count */
int spock_get_timing(spock_handle_t h, double *iter_ms_avg, int32_t *iters_timed);
/* measurement aid (no reference counterpart): runs `reps` further CP iterations on the current state with a
 * HIP event before and after every kernel launch on the handle's stream.  avg_ms / launches: [4], average
 * duration and launch count per kernel kind: 0 subtree phase A, 1 crown, 2 subtree phase B, 3 whole-tree kernel. */
int spock_profile_kernels(spock_handle_t h, int32_t reps, double *avg_ms, int32_t *launches);
/* the same measurement per launch of one CP iteration, in launch order: kernel mode (spock_internal.hpp: GroupMode), its layer
 * of groups / tree stage, average duration over the `reps` iterations; *n receives the launches per iteration (cap: array sizes) */
int spock_profile_sequence(spock_handle_t h, int32_t reps, int32_t *modes, int32_t *levels, double *avg_ms, int32_t cap, int32_t *n);
/* measurement aid: clock stamps (shader clock, 100 MHz wall clock) written by workgroup 0 of instance 0 during the
 * last spock_profile_kernels run with SPOCK_AMD_DEBUG_FLAGS including 512; out: [3 kernels][64 stamps][2]. */
int spock_debug_stamps(spock_handle_t h, uint64_t *out);

const char *spock_last_error(void);
int spock_version(void);
/* sizeof of the four ABI structs as this library was compiled: out[0..3] = spock_problem_t, spock_options_t, spock_result_t,
 * spock_sizes_t (as many as `cap` allows).  A binding asserts them against its own mirrors when it loads the library, so that a
 * stale mirror fails loudly instead of making spock_create read past a struct (no reference counterpart: Julia has no FFI here). */
int spock_abi_struct_sizes(int32_t *out, int32_t cap);
/* bytes of the device path's arithmetic / storage type: 8 in libspock_amd.so (the reference is Float64 only, src/model.jl:304-307),
 * 4 in libspock_amd_f32.so, the same sources built in single precision for BASELINE config 5's tolerance study.  The ABI is
 * double in both; control blocks, residual reductions and the build step stay double. */
int spock_real_size(void);

#ifdef __cplusplus
}
#endif
#endif /* SPOCK_AMD_H */
