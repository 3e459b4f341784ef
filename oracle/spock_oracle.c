/*
 * spock_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A single-thread fp64 restatement, in plain C, of the Chambolle-Pock (CP) and
 * SuperMann-accelerated CP ("SP" = Spock) hot path of kul-optec/spock.jl.  It is
 * written from a reading of the reference's Julia sources and follows their loop
 * structure and summation order so that trajectories can be compared element by
 * element.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this file's shared object; the shipped GPU path never calls it.
 *
 * Parity pin: this oracle reproduces the residual trace the reference ships in
 * examples/server_heat/output/residuals.tikz (CP: all 3159 values; SP: the first
 * ~50 iterations, after which the SP trajectory is chaotically sensitive to
 * rounding) -- see tests/test_oracle_kat.py and tests/golden/.
 *
 * Reference files followed (all under /root/reference):
 *   src/model_dynamics/implicit_l.jl   get_nz :5-15, get_nv :17-44, ricatti_offline :46-97,
 *                                      L! :177-318, L_transpose! :320-449, spock_mul! :451-481,
 *                                      spock_dot :485-516, projection_S1! :559-686,
 *                                      projection_S2! :688-727, prox_f! :729-748,
 *                                      project_on_leaf_constraints! :752-810,
 *                                      project_on_nonleaf_constraints! :812-909, prox_h_conj! :911-951
 *   src/model_algorithms/cp.jl         update_zbar! :5-15, update_vbar! :17-32, should_terminate! :54-123,
 *                                      run_cp! :188-232
 *   src/model_algorithms/sp.jl         run_sp! :358-469 and helpers
 *   src/model_algorithms/qnewton_directions/anderson.jl  anderson! :1-123
 *   src/models/cpock.jl, src/models/spock.jl  build step, initial values, L_norm constants
 *   src/risk_measures.jl               project_onto_cone! :14-71
 *   src/constraints.jl                 box clamps and the (x,u) interleave maps :24-127
 *   src/scenario_tree.jl               generate_scenario_tree_uniform_branching_factor_v2 :61-109
 *
 * All matrices are column-major (Julia layout).  Indices are 0-based here; the
 * reference is 1-based.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_CP 1
#define ORC_SP 2
#define ORC_SILENT 1
#define ORC_LOG 2

enum { CONE_ZEROS = 0, CONE_NONNEG = 1, CONE_NONPOS = 2, CONE_REALS = 3, CONE_SOC = 4 };

typedef struct {
    int N, d, nx, nu, ny;
    const double *A, *B;       /* d matrices each, nx*nx and nx*nu, column-major */
    int nQ, nR, nQN;           /* 1 = one shared matrix, else n-1 / n-1 / nl */
    const double *Q, *R, *QN;
    const double *E, *F, *b;   /* n_rms blocks of ny x d, ny x d, ny */
    int n_cones;
    const int *cone_kind, *cone_dim;
    double x_min, x_max, u_min, u_max;
    int n_rms;                 /* 1: one risk measure shared by all non-leaf nodes (get_uniform_rms_v2, risk_measures.jl:93-102), else one
                                  per non-leaf node with a common length(b) and cone (get_nonuniform_rms_avar_v2, :127-136) */
} orc_problem;

typedef struct {
    int iterations;      /* number of T evaluations whose termination test ran */
    int iter_ref;        /* the count the reference prints ("Finished in k iterations") */
    int status;          /* 0 converged, 1 max iterations */
    double xi1, xi2, xi; /* last residuals */
} orc_result;

typedef struct {
    int alg;
    int N, d, nx, nu, ny, n, nn, nl, nz, nv;
    int zx, zu, zs, zt, zy;
    int o2, o3, o4, o5, o6, o7, o11, o12, o13, o14; /* v offsets; v1 at 0 */
    double *A, *B;
    int nQ, nR, nQN, n_rms;
    double *sqrtQ, *sqrtR, *sqrtQN;
    double *b;
    int n_cones, *ck, *cd;
    double x_min, x_max, u_min, u_max;
    double L_norm;
    int sp_variant, sp_max_backtrack;   /* 0 = the reference's run_sp! as it stands; 1 = corrected Anderson + working line search (below) */
    /* build-step products, stored per stage (identical for every node of a stage
       because ricatti_offline uses only A_w, B_w and the uniform tree shape) */
    double *P;    /* N stages x nx*nx            */
    double *K;    /* N-1 stages x nu*nx          */
    double *Rch;  /* N-1 stages x nu*nu (lower Cholesky factor of Rtilde) */
    double *ABK;  /* N stages x d child slots x nx*nx (stage of the child) */
    double *ls;   /* (ny+2d)^2 kernel projector   */
    int *stage;   /* stage of each node           */
    /* solver state (model.jl:23-44) */
    double *x0;
    double *z, *v, *zbar, *vbar, *z_workspace, *rz, *rv, *xi_1, *xi_2, *xi, *delta_z, *delta_v;
    double *z_old, *v_old, *delta_rz, *delta_rv;
    double res_0[2];
    /* internal workspaces (model.jl:80-125) */
    double *mul_x, *mul_x2, *mul_u, *buf_z, *buf_v, *ls_b, *ls_b2, *ric_q, *ric_d, *sum_for_d;
    double *prox_v_wsp, *proj_leaf, *proj_nleaf;
    /* SuperMann / Anderson (model.jl:137-226) */
    double *rho_z, *rho_v, *dz, *dv, *w, *u, *wbar, *ubar, *rw, *ru, *w_workspace, *wLz, *wLv;
    double *rz_old, *rv_old;
    double *MP, *MR, *aa_wsp, *aa_Q;
    double *dzo, *dvo, *drzo, *drvo; /* delta_*_old: written, never read (sp.jl:346-356) */
    double aa_R[9], aa_gamma[3];
} orc;

/* ---------------------------------------------------------------- small dense helpers */
static double *dalloc(size_t k) {
    double *p = (double *)calloc(k ? k : 1, sizeof(double));
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static double *dones(size_t k) {
    double *p = dalloc(k);
    for (size_t i = 0; i < k; i++) p[i] = 1.0;
    return p;
}
/* y = A x, A is m x n */
static void gemv_n(int m, int n, const double *A, const double *x, double *y) {
    for (int r = 0; r < m; r++) {
        double s = 0.;
        for (int k = 0; k < n; k++) s += A[r + k * m] * x[k];
        y[r] = s;
    }
}
/* y = A' x, A is m x n, y has n entries */
static void gemv_t(int m, int n, const double *A, const double *x, double *y) {
    for (int c = 0; c < n; c++) {
        double s = 0.;
        for (int r = 0; r < m; r++) s += A[r + c * m] * x[r];
        y[c] = s;
    }
}
/* C(m x n) = op(A) * op(B); A is stored (ta ? k x m : m x k), B is stored (tb ? n x k : k x n) */
static void gemm(int m, int n, int k, const double *A, int ta, const double *B, int tb, double *C) {
    for (int c = 0; c < n; c++)
        for (int r = 0; r < m; r++) {
            double s = 0.;
            for (int l = 0; l < k; l++) {
                double a = ta ? A[l + r * k] : A[r + l * m];
                double bb = tb ? B[c + l * n] : B[l + c * k];
                s += a * bb;
            }
            C[r + c * m] = s;
        }
}
/* lower Cholesky factor of symmetric positive definite M (n x n), in Lo; returns 0 on success */
static int chol_lower(int n, const double *M, double *Lo) {
    memset(Lo, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        double s = M[j + j * n];
        for (int k = 0; k < j; k++) s -= Lo[j + k * n] * Lo[j + k * n];
        if (s <= 0.) return 1;
        double dj = sqrt(s);
        Lo[j + j * n] = dj;
        for (int i = j + 1; i < n; i++) {
            double t = M[i + j * n];
            for (int k = 0; k < j; k++) t -= Lo[i + k * n] * Lo[j + k * n];
            Lo[i + j * n] = t / dj;
        }
    }
    return 0;
}
/* solve (Lo Lo') x = rhs in place */
static void chol_solve(int n, const double *Lo, double *x) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= Lo[i + k * n] * x[k];
        x[i] = s / Lo[i + i * n];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= Lo[k + i * n] * x[k];
        x[i] = s / Lo[i + i * n];
    }
}
/* cyclic Jacobi eigen-decomposition of symmetric S (n x n): S = V diag(w) V' */
static void jacobi_eig(int n, const double *S, double *V, double *w) {
    double *a = dalloc((size_t)n * n);
    memcpy(a, S, sizeof(double) * n * n);
    for (int i = 0; i < n * n; i++) V[i] = 0.;
    for (int i = 0; i < n; i++) V[i + i * n] = 1.;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += a[p + q * n] * a[p + q * n];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[p + q * n];
                if (fabs(apq) < 1e-300) continue;
                double theta = (a[q + q * n] - a[p + p * n]) / (2. * apq);
                double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                double c = 1. / sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = a[k + p * n], akq = a[k + q * n];
                    a[k + p * n] = c * akp - s * akq;
                    a[k + q * n] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = a[p + k * n], aqk = a[q + k * n];
                    a[p + k * n] = c * apk - s * aqk;
                    a[q + k * n] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k + p * n], vkq = V[k + q * n];
                    V[k + p * n] = c * vkp - s * vkq;
                    V[k + q * n] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = a[i + i * n];
    free(a);
}
/* principal square root of a symmetric PSD matrix (cpock.jl:119-121 `map(x -> sqrt(x), cost.Q)`).
   A diagonal input takes the element-wise path, like Julia's sqrt(::Diagonal). */
static void sym_sqrt(int n, const double *S, double *out) {
    int diag = 1;
    for (int c = 0; c < n && diag; c++)
        for (int r = 0; r < n; r++)
            if (r != c && S[r + c * n] != 0.) { diag = 0; break; }
    if (diag) {
        memset(out, 0, sizeof(double) * n * n);
        for (int i = 0; i < n; i++) out[i + i * n] = sqrt(S[i + i * n]);
        return;
    }
    double *V = dalloc((size_t)n * n), *w = dalloc(n);
    jacobi_eig(n, S, V, w);
    for (int c = 0; c < n; c++)
        for (int r = 0; r < n; r++) {
            double s = 0.;
            for (int k = 0; k < n; k++) s += V[r + k * n] * sqrt(w[k] > 0. ? w[k] : 0.) * V[c + k * n];
            out[r + c * n] = s;
        }
    free(V); free(w);
}

/* ---------------------------------------------------------------- tree helpers (scenario_tree.jl:61-109) */
static int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }
static inline int child0(const orc *o, int i) { return i * o->d + 1; }          /* first child, 0-based */
static inline int slot_event(const orc *o, int c) { return (2 + c) % o->d; }     /* 0-based event w of child slot c:
                                                                                   w = (j % d) + 1 with j = 2+(i-1)d+c (1-based) */

/* ---------------------------------------------------------------- build step */
static void ricatti_offline(orc *o) { /* implicit_l.jl:46-97 */
    int nx = o->nx, nu = o->nu, d = o->d, N = o->N;
    o->P = dalloc((size_t)N * nx * nx);
    o->K = dalloc((size_t)(N > 1 ? N - 1 : 1) * nu * nx);
    o->Rch = dalloc((size_t)(N > 1 ? N - 1 : 1) * nu * nu);
    o->ABK = dalloc((size_t)N * d * nx * nx);
    double *Pl = o->P + (size_t)(N - 1) * nx * nx;
    for (int i = 0; i < nx; i++) Pl[i + i * nx] = 1.;
    double *sum_r = dalloc((size_t)nu * nu), *sum_k = dalloc((size_t)nu * nx), *sum_p = dalloc((size_t)nx * nx);
    double *t1 = dalloc((size_t)nu * nx), *t2 = dalloc((size_t)nx * nx), *t3 = dalloc((size_t)nx * nx);
    double *rt = dalloc((size_t)nu * nu), *col = dalloc(nu);
    for (int t = N - 2; t >= 0; t--) {
        const double *Pc = o->P + (size_t)(t + 1) * nx * nx; /* P of the children */
        memset(sum_r, 0, sizeof(double) * nu * nu);
        memset(sum_k, 0, sizeof(double) * nu * nx);
        for (int c = 0; c < d; c++) {
            const double *Aw = o->A + (size_t)slot_event(o, c) * nx * nx;
            const double *Bw = o->B + (size_t)slot_event(o, c) * nx * nu;
            gemm(nu, nx, nx, Bw, 1, Pc, 0, t1);                 /* B' P            */
            gemm(nu, nu, nx, t1, 0, Bw, 0, rt);                 /* (B' P) B        */
            for (int k = 0; k < nu * nu; k++) sum_r[k] += rt[k];
            double *tk = dalloc((size_t)nu * nx);
            gemm(nu, nx, nx, t1, 0, Aw, 0, tk);                 /* (B' P) A        */
            for (int k = 0; k < nu * nx; k++) sum_k[k] += tk[k];
            free(tk);
        }
        for (int c = 0; c < nu; c++)                            /* 0.5 (S + S') then I + . */
            for (int r = 0; r < nu; r++)
                rt[r + c * nu] = (r == c ? 1. : 0.) + 0.5 * (sum_r[r + c * nu] + sum_r[c + r * nu]);
        double *Lo = o->Rch + (size_t)t * nu * nu;
        if (chol_lower(nu, rt, Lo)) { fprintf(stderr, "oracle: Rtilde not positive definite\n"); abort(); }
        double *Kt = o->K + (size_t)t * nu * nx;
        for (int c = 0; c < nx; c++) {                          /* K = chol \ (-sum_k) */
            for (int r = 0; r < nu; r++) col[r] = -sum_k[r + c * nu];
            chol_solve(nu, Lo, col);
            for (int r = 0; r < nu; r++) Kt[r + c * nu] = col[r];
        }
        memset(sum_p, 0, sizeof(double) * nx * nx);
        for (int c = 0; c < d; c++) {
            const double *Aw = o->A + (size_t)slot_event(o, c) * nx * nx;
            const double *Bw = o->B + (size_t)slot_event(o, c) * nx * nu;
            double *abk = o->ABK + ((size_t)(t + 1) * d + c) * nx * nx;
            gemm(nx, nx, nu, Bw, 0, Kt, 0, t2);                 /* B K */
            for (int k = 0; k < nx * nx; k++) abk[k] = Aw[k] + t2[k];
            gemm(nx, nx, nx, abk, 1, Pc, 0, t2);                /* ABK' P */
            gemm(nx, nx, nx, t2, 0, abk, 0, t3);                /* (ABK' P) ABK */
            for (int k = 0; k < nx * nx; k++) sum_p[k] += t3[k];
        }
        double *Pt = o->P + (size_t)t * nx * nx;
        gemm(nx, nx, nu, Kt, 1, Kt, 0, t2);                     /* K' K */
        for (int c = 0; c < nx; c++)
            for (int r = 0; r < nx; r++)
                Pt[r + c * nx] = ((r == c ? 1. : 0.) + t2[r + c * nx]) + sum_p[r + c * nx];
    }
    free(sum_r); free(sum_k); free(sum_p); free(t1); free(t2); free(t3); free(rt); free(col);
}

/* Orthogonal projector onto ker [E' -I -I ; F' 0 0]  (cpock.jl:71-81,125-128).
   The reference forms U*pinv(U) with U from an SVD of a null-space basis, i.e. the unique orthogonal
   projector; here: I - W W' with W an orthonormal basis of the row space (MGS with re-orthogonalisation). */
static void kernel_projector(orc *o, const double *E, const double *F, double *ls_out) {
    int d = o->d, ny = o->ny, m = ny + 2 * d;
    double *W = dalloc((size_t)2 * d * m), *r = dalloc(m);
    int rank = 0;
    for (int k = 0; k < 2 * d; k++) {
        memset(r, 0, sizeof(double) * m);
        if (k < d) {
            for (int i = 0; i < ny; i++) r[i] = E[i + k * ny];
            r[ny + k] = -1.; r[ny + d + k] = -1.;
        } else {
            for (int i = 0; i < ny; i++) r[i] = F[i + (k - d) * ny];
        }
        double n0 = 0.;
        for (int i = 0; i < m; i++) n0 += r[i] * r[i];
        n0 = sqrt(n0);
        if (n0 == 0.) continue;
        for (int pass = 0; pass < 2; pass++)
            for (int q = 0; q < rank; q++) {
                double s = 0.;
                for (int i = 0; i < m; i++) s += W[i + q * m] * r[i];
                for (int i = 0; i < m; i++) r[i] -= s * W[i + q * m];
            }
        double n1 = 0.;
        for (int i = 0; i < m; i++) n1 += r[i] * r[i];
        n1 = sqrt(n1);
        if (n1 <= 1e-12 * n0) continue;
        for (int i = 0; i < m; i++) W[i + rank * m] = r[i] / n1;
        rank++;
    }
    for (int c = 0; c < m; c++)
        for (int rr = 0; rr < m; rr++) {
            double s = 0.;
            for (int q = 0; q < rank; q++) s += W[rr + q * m] * W[c + q * m];
            ls_out[rr + c * m] = (rr == c ? 1. : 0.) - s;
        }
    free(W); free(r);
}

orc *orc_create(const orc_problem *p, int alg) {
    if (p->d <= 1 || p->N < 2) return NULL; /* scenario_tree.jl:62-64 */
    orc *o = (orc *)calloc(1, sizeof(orc));
    o->alg = alg;
    o->N = p->N; o->d = p->d; o->nx = p->nx; o->nu = p->nu; o->ny = p->ny;
    int d = o->d, N = o->N, nx = o->nx, nu = o->nu, ny = o->ny;
    o->n = (ipow(d, N) - 1) / (d - 1);
    o->nl = ipow(d, N - 1);
    o->nn = (ipow(d, N - 1) - 1) / (d - 1);
    int n = o->n, nn = o->nn, nl = o->nl;
    /* z layout x|u|s|tau|y  (implicit_l.jl:5-15,106-158) */
    o->zx = 0; o->zu = n * nx; o->zs = o->zu + nn * nu; o->zt = o->zs + n; o->zy = o->zt + (n - 1);
    o->nz = o->zy + nn * ny;
    /* v offsets (cpock.jl:51-60); nGamma_nonleaf = nn (nx+nu), nGamma_leaf = nl nx */
    o->o2 = nn * ny; o->o3 = o->o2 + nn; o->o4 = o->o3 + nx * (n - 1); o->o5 = o->o4 + nu * (n - 1);
    o->o6 = o->o5 + (n - 1); o->o7 = o->o6 + (n - 1); o->o11 = o->o7 + nn * (nx + nu);
    o->o12 = o->o11 + nl * nx; o->o13 = o->o12 + nl; o->o14 = o->o13 + nl;
    o->nv = o->o14 + nl * nx;
    o->L_norm = (alg == ORC_CP) ? 10. : 3.28; /* cpock.jl:63, models/spock.jl:64 */

    o->A = dalloc((size_t)d * nx * nx); memcpy(o->A, p->A, sizeof(double) * d * nx * nx);
    o->B = dalloc((size_t)d * nx * nu); memcpy(o->B, p->B, sizeof(double) * d * nx * nu);
    o->nQ = p->nQ; o->nR = p->nR; o->nQN = p->nQN;
    o->sqrtQ = dalloc((size_t)o->nQ * nx * nx);
    o->sqrtR = dalloc((size_t)o->nR * nu * nu);
    o->sqrtQN = dalloc((size_t)o->nQN * nx * nx);
    for (int k = 0; k < o->nQ; k++) sym_sqrt(nx, p->Q + (size_t)k * nx * nx, o->sqrtQ + (size_t)k * nx * nx);
    for (int k = 0; k < o->nR; k++) sym_sqrt(nu, p->R + (size_t)k * nu * nu, o->sqrtR + (size_t)k * nu * nu);
    for (int k = 0; k < o->nQN; k++) sym_sqrt(nx, p->QN + (size_t)k * nx * nx, o->sqrtQN + (size_t)k * nx * nx);
    o->n_rms = p->n_rms > 1 ? p->n_rms : 1;
    o->b = dalloc((size_t)o->n_rms * ny); memcpy(o->b, p->b, sizeof(double) * ny * o->n_rms);
    o->n_cones = p->n_cones;
    o->ck = (int *)malloc(sizeof(int) * (p->n_cones ? p->n_cones : 1));
    o->cd = (int *)malloc(sizeof(int) * (p->n_cones ? p->n_cones : 1));
    memcpy(o->ck, p->cone_kind, sizeof(int) * p->n_cones);
    memcpy(o->cd, p->cone_dim, sizeof(int) * p->n_cones);
    o->x_min = p->x_min; o->x_max = p->x_max; o->u_min = p->u_min; o->u_max = p->u_max;

    o->stage = (int *)malloc(sizeof(int) * n);
    { int first = 0, cnt = 1;
      for (int t = 0; t < N; t++) { for (int k = 0; k < cnt; k++) o->stage[first + k] = t; first += cnt; cnt *= d; } }

    ricatti_offline(o);
    o->ls = dalloc((size_t)o->n_rms * (ny + 2 * d) * (ny + 2 * d));
    for (int k = 0; k < o->n_rms; k++)   /* Ms[i] of cpock.jl:71-81 is built from rms[i].E, rms[i].F */
        kernel_projector(o, p->E + (size_t)k * ny * d, p->F + (size_t)k * ny * d, o->ls + (size_t)k * (ny + 2 * d) * (ny + 2 * d));

    int nz = o->nz, nv = o->nv, m = ny + 2 * d;
    o->x0 = dalloc(nx);
    o->z = dalloc(nz); o->v = dalloc(nv); o->zbar = dalloc(nz); o->vbar = dalloc(nv);
    o->z_workspace = dalloc(nz); o->rz = dalloc(nz); o->rv = dalloc(nv);
    o->xi_1 = dalloc(nz); o->xi_2 = dalloc(nv); o->xi = dalloc(nz);
    o->delta_z = dalloc(nz); o->delta_v = dalloc(nv);
    o->z_old = dones(nz); o->v_old = dones(nv);          /* ones(nz), ones(nv): cpock.jl:97-98 */
    o->delta_rz = dalloc(nz); o->delta_rv = dalloc(nv);
    o->res_0[0] = -INFINITY; o->res_0[1] = -INFINITY;
    o->mul_x = dalloc(nx); o->mul_x2 = dalloc(nx); o->mul_u = dalloc(nu);
    o->buf_z = dalloc(nz); o->buf_v = dalloc(nv); o->ls_b = dalloc(m); o->ls_b2 = dalloc(m);
    o->ric_q = dalloc((size_t)n * nx); o->ric_d = dalloc((size_t)n * nu); o->sum_for_d = dalloc(nu);
    o->prox_v_wsp = dalloc(nv); o->proj_leaf = dalloc(nx + 2); o->proj_nleaf = dalloc(nx + nu + 2);
    if (alg == ORC_SP) { /* models/spock.jl:159-201 */
        o->rho_z = dalloc(nz); o->rho_v = dalloc(nv); o->dz = dalloc(nz); o->dv = dalloc(nv);
        o->w = dalloc(nz); o->u = dalloc(nv); o->wbar = dalloc(nz); o->ubar = dalloc(nv);
        o->rw = dalloc(nz); o->ru = dalloc(nv); o->w_workspace = dalloc(nz);
        o->wLz = dalloc(nz); o->wLv = dalloc(nv);
        o->rz_old = dones(nz); o->rv_old = dones(nv);
        size_t nr = (size_t)nz + nv;
        o->MP = dalloc(nr * 3); o->MR = dalloc(nr * 3); o->aa_wsp = dalloc(nr); o->aa_Q = dalloc(nr * 3);
        o->dzo = dalloc(nz); o->dvo = dalloc(nv); o->drzo = dalloc(nz); o->drvo = dalloc(nv);
        for (int i = 0; i < 9; i++) o->aa_R[i] = 0.;
        o->aa_R[0] = o->aa_R[4] = o->aa_R[8] = 1.;
    }
    return o;
}

void orc_free(orc *o) {
    if (!o) return;
    double **ps[] = { &o->A, &o->B, &o->sqrtQ, &o->sqrtR, &o->sqrtQN, &o->b, &o->P, &o->K, &o->Rch, &o->ABK, &o->ls,
        &o->x0, &o->z, &o->v, &o->zbar, &o->vbar, &o->z_workspace, &o->rz, &o->rv, &o->xi_1, &o->xi_2, &o->xi,
        &o->delta_z, &o->delta_v, &o->z_old, &o->v_old, &o->delta_rz, &o->delta_rv, &o->mul_x, &o->mul_x2, &o->mul_u,
        &o->buf_z, &o->buf_v, &o->ls_b, &o->ls_b2, &o->ric_q, &o->ric_d, &o->sum_for_d, &o->prox_v_wsp, &o->proj_leaf,
        &o->proj_nleaf, &o->rho_z, &o->rho_v, &o->dz, &o->dv, &o->w, &o->u, &o->wbar, &o->ubar, &o->rw, &o->ru,
        &o->w_workspace, &o->wLz, &o->wLv, &o->rz_old, &o->rv_old, &o->MP, &o->MR, &o->aa_wsp, &o->aa_Q,
        &o->dzo, &o->dvo, &o->drzo, &o->drvo };
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); i++) free(*ps[i]);
    free(o->ck); free(o->cd); free(o->stage);
    free(o);
}

/* ---------------------------------------------------------------- accessors */
void orc_sizes(const orc *o, int *out /* n, nn, nl, nz, nv, ny */) {
    out[0] = o->n; out[1] = o->nn; out[2] = o->nl; out[3] = o->nz; out[4] = o->nv; out[5] = o->ny;
}
void orc_offsets(const orc *o, int *out /* zx zu zs zt zy | v1 v2 v3 v4 v5 v6 v7 v11 v12 v13 v14 */) {
    int t[] = { o->zx, o->zu, o->zs, o->zt, o->zy, 0, o->o2, o->o3, o->o4, o->o5, o->o6, o->o7, o->o11, o->o12, o->o13, o->o14 };
    memcpy(out, t, sizeof(t));
}
double *orc_z(orc *o) { return o->z; }
double *orc_v(orc *o) { return o->v; }
double orc_get_L_norm(const orc *o) { return o->L_norm; }
void orc_set_L_norm(orc *o, double L) { o->L_norm = L; }
/* build products for cross-checking the product's own host build step */
const double *orc_P(const orc *o) { return o->P; }
const double *orc_K(const orc *o) { return o->K; }
const double *orc_Rchol(const orc *o) { return o->Rch; }
const double *orc_ABK(const orc *o) { return o->ABK; }
const double *orc_ls(const orc *o) { return o->ls; }
const double *orc_sqrtQ(const orc *o) { return o->sqrtQ; }
const double *orc_sqrtR(const orc *o) { return o->sqrtR; }
const double *orc_sqrtQN(const orc *o) { return o->sqrtQN; }

/* ---------------------------------------------------------------- operators */
static inline const double *mat_sel(const double *base, int cnt, int idx, int sz) {
    return base + (size_t)(cnt == 1 ? 0 : idx) * sz;
}

void orc_L(orc *o, const double *z, double *v) { /* implicit_l.jl:177-318 */
    int nx = o->nx, nu = o->nu, ny = o->ny, nn = o->nn, nl = o->nl, d = o->d;
    for (int k = 0; k < nn * ny; k++) v[k] = z[o->zy + k];                                   /* v1 :208-210 */
    for (int i = 0; i < nn; i++) {                                                           /* v2 :214-222 */
        double dot_p = 0.;
        for (int k = 0; k < ny; k++) dot_p += mat_sel(o->b, o->n_rms, i, ny)[k] * z[o->zy + i * ny + k];   /* rms[i].b (:219) */
        v[o->o2 + i] = z[o->zs + i] - dot_p;
    }
    for (int i = 0; i < nn; i++)                                                             /* v3 :226-235 */
        for (int c = 0; c < d; c++) {
            int j = child0(o, i) + c, j_ind = j - 1; /* 0-based index among non-root nodes */
            gemv_n(nx, nx, mat_sel(o->sqrtQ, o->nQ, j_ind, nx * nx), z + o->zx + i * nx, o->mul_x);
            for (int k = 0; k < nx; k++) v[o->o3 + j_ind * nx + k] = o->mul_x[k];
        }
    for (int i = 0; i < nn; i++)                                                             /* v4 :239-248 */
        for (int c = 0; c < d; c++) {
            int j = child0(o, i) + c, j_ind = j - 1;
            gemv_n(nu, nu, mat_sel(o->sqrtR, o->nR, j_ind, nu * nu), z + o->zu + i * nu, o->mul_u);
            for (int k = 0; k < nu; k++) v[o->o4 + j_ind * nu + k] = o->mul_u[k];
        }
    for (int jn = 0; jn < o->n - 1; jn++) v[o->o5 + jn] = 0.5 * z[o->zt + jn];              /* v5 :252-260 */
    for (int jn = 0; jn < o->n - 1; jn++) v[o->o6 + jn] = 0.5 * z[o->zt + jn];              /* v6 :264-272 */
    for (int i = 0; i < nn; i++) {                                                           /* v7 :275-282, constraints.jl:111-127 */
        for (int k = 0; k < nx; k++) v[o->o7 + i * (nx + nu) + k] = z[o->zx + i * nx + k];
        for (int k = 0; k < nu; k++) v[o->o7 + i * (nx + nu) + nx + k] = z[o->zu + i * nu + k];
    }
    for (int i = 0; i < nl; i++) {                                                           /* v11 :288-294 */
        int j = nn + i;
        gemv_n(nx, nx, mat_sel(o->sqrtQN, o->nQN, i, nx * nx), z + o->zx + j * nx, o->mul_x);
        for (int k = 0; k < nx; k++) v[o->o11 + i * nx + k] = o->mul_x[k];
    }
    for (int i = 0; i < nl; i++) v[o->o12 + i] = 0.5 * z[o->zs + nn + i];                    /* v12 :298-301 */
    for (int i = 0; i < nl; i++) v[o->o13 + i] = 0.5 * z[o->zs + nn + i];                    /* v13 :305-308 */
    for (int k = 0; k < nl * nx; k++) v[o->o14 + k] = z[o->zx + nn * nx + k];                /* v14 :311-317 */
}

void orc_Lt(orc *o, double *z, const double *v) { /* implicit_l.jl:320-449 */
    int nx = o->nx, nu = o->nu, ny = o->ny, nn = o->nn, nl = o->nl, d = o->d;
    for (int i = 0; i < nn; i++)                                                             /* x <- Gamma' v7 :351-356 */
        for (int k = 0; k < nx; k++) z[o->zx + i * nx + k] = v[o->o7 + i * (nx + nu) + k];
    for (int i = 0; i < nn; i++)                                                             /* + sqrtQ v3 :365-373 */
        for (int c = 0; c < d; c++) {
            int j_ind = child0(o, i) + c - 1;
            gemv_n(nx, nx, mat_sel(o->sqrtQ, o->nQ, j_ind, nx * nx), v + o->o3 + j_ind * nx, o->mul_x);
            for (int k = 0; k < nx; k++) z[o->zx + i * nx + k] += o->mul_x[k];
        }
    for (int k = 0; k < nl * nx; k++) z[o->zx + nn * nx + k] = v[o->o14 + k];                /* leaf x <- v14 :376-381 */
    for (int i = 0; i < nl; i++) {                                                           /* + sqrtQN v11 :390-396 */
        gemv_n(nx, nx, mat_sel(o->sqrtQN, o->nQN, i, nx * nx), v + o->o11 + i * nx, o->mul_x);
        for (int k = 0; k < nx; k++) z[o->zx + (nn + i) * nx + k] += o->mul_x[k];
    }
    for (int i = 0; i < nn; i++)                                                             /* u <- Gamma' v7 :399-404 */
        for (int k = 0; k < nu; k++) z[o->zu + i * nu + k] = v[o->o7 + i * (nx + nu) + nx + k];
    for (int i = 0; i < nn; i++)                                                             /* + sqrtR v4 :412-420 */
        for (int c = 0; c < d; c++) {
            int j_ind = child0(o, i) + c - 1;
            gemv_n(nu, nu, mat_sel(o->sqrtR, o->nR, j_ind, nu * nu), v + o->o4 + j_ind * nu, o->mul_u);
            for (int k = 0; k < nu; k++) z[o->zu + i * nu + k] += o->mul_u[k];
        }
    for (int i = 0; i < nn; i++)                                                             /* y :423-429 */
        for (int k = 0; k < ny; k++) z[o->zy + i * ny + k] = v[i * ny + k] - mat_sel(o->b, o->n_rms, i, ny)[k] * v[o->o2 + i];   /* rms[i].b (:427) */
    for (int jn = 0; jn < o->n - 1; jn++) z[o->zt + jn] = 0.5 * v[o->o5 + jn] + 0.5 * v[o->o6 + jn]; /* tau :432-437 */
    for (int i = 0; i < nn; i++) z[o->zs + i] = v[o->o2 + i];                                /* s non-leaf :440-442 */
    for (int i = 0; i < nl; i++) z[o->zs + nn + i] = 0.5 * v[o->o12 + i] + 0.5 * v[o->o13 + i]; /* s leaf :445-448 */
}

/* C <- alpha op(L) B + beta C  (implicit_l.jl:451-481) */
static void spock_mul(orc *o, double *C, int trans, const double *Bv, double alpha, double beta) {
    if (!trans) {
        memcpy(o->buf_v, C, sizeof(double) * o->nv);
        orc_L(o, Bv, C);
        for (int i = 0; i < o->nv; i++) C[i] = alpha * C[i] + beta * o->buf_v[i];
    } else {
        memcpy(o->buf_z, C, sizeof(double) * o->nz);
        orc_Lt(o, C, Bv);
        for (int i = 0; i < o->nz; i++) C[i] = alpha * C[i] + beta * o->buf_z[i];
    }
}
void orc_mul(orc *o, double *C, int trans, const double *Bv, double alpha, double beta) { spock_mul(o, C, trans, Bv, alpha, beta); }

/* <a, P b> with P = [[I, -gamma L'],[-sigma L, I]]  (implicit_l.jl:485-516); needs the SP workspaces */
double orc_dot(orc *o, const double *a_z, const double *a_v, const double *b_z, const double *b_v, double gamma, double sigma) {
    double *Lz = o->wLz, *Lv = o->wLv;
    int own = 0;
    if (!Lz) { Lz = dalloc(o->nz); Lv = dalloc(o->nv); own = 1; }
    orc_L(o, b_z, Lv);
    orc_Lt(o, Lz, b_v);
    for (int k = 0; k < o->nz; k++) Lz[k] = b_z[k] - gamma * Lz[k];
    for (int k = 0; k < o->nv; k++) Lv[k] = -sigma * Lv[k] + b_v[k];
    double res = 0.;
    for (int k = 0; k < o->nz; k++) res += a_z[k] * Lz[k];
    for (int k = 0; k < o->nv; k++) res += a_v[k] * Lv[k];
    if (own) { free(Lz); free(Lv); }
    return res;
}
static double spock_norm(orc *o, const double *a_z, const double *a_v, double gamma, double sigma) {
    return sqrt(orc_dot(o, a_z, a_v, a_z, a_v, gamma, sigma));
}

static void projection_S1(orc *o, double *z1) { /* implicit_l.jl:559-686; z1 = (x, u) */
    int nx = o->nx, nu = o->nu, d = o->d, nn = o->nn, n = o->n;
    int u_off = n * nx;
    double *q = o->ric_q, *dd = o->ric_d;
    for (int i = nn; i < n; i++)
        for (int k = 0; k < nx; k++) q[i * nx + k] = -z1[i * nx + k];                         /* :578-582 */
    for (int i = nn - 1; i >= 0; i--) {                                                       /* :584-637 */
        int t = o->stage[i];
        for (int k = 0; k < nu; k++) o->sum_for_d[k] = 0.;
        for (int c = 0; c < d; c++) {
            int j = child0(o, i) + c;
            gemv_t(nx, nu, o->B + (size_t)slot_event(o, c) * nx * nu, q + j * nx, o->mul_u);  /* B' q_j */
            for (int k = 0; k < nu; k++) o->sum_for_d[k] += o->mul_u[k];
        }
        for (int k = 0; k < nu; k++) o->mul_u[k] = z1[u_off + i * nu + k] - o->sum_for_d[k];
        chol_solve(nu, o->Rch + (size_t)t * nu * nu, o->mul_u);                               /* ldiv!(R_chol[i], .) */
        for (int k = 0; k < nu; k++) dd[i * nu + k] = o->mul_u[k];
        for (int k = 0; k < nx; k++) q[i * nx + k] = 0.;
        for (int c = 0; c < d; c++) {
            int j = child0(o, i) + c;
            gemv_n(nx, nu, o->B + (size_t)slot_event(o, c) * nx * nu, dd + i * nu, o->mul_x); /* B d */
            gemv_n(nx, nx, o->P + (size_t)(t + 1) * nx * nx, o->mul_x, o->mul_x2);            /* P_j (B d) */
            for (int k = 0; k < nx; k++) o->mul_x2[k] += q[j * nx + k];
            gemv_t(nx, nx, o->ABK + ((size_t)(t + 1) * d + c) * nx * nx, o->mul_x2, o->mul_x);/* ABK_j' (.) */
            for (int k = 0; k < nx; k++) q[i * nx + k] += o->mul_x[k];
        }
        for (int k = 0; k < nu; k++) o->mul_u[k] = dd[i * nu + k] - z1[u_off + i * nu + k];
        gemv_t(nu, nx, o->K + (size_t)t * nu * nx, o->mul_u, o->mul_x);                       /* K' (d - ubar) */
        for (int k = 0; k < nx; k++) q[i * nx + k] += o->mul_x[k];
        for (int k = 0; k < nx; k++) q[i * nx + k] -= z1[i * nx + k];
    }
    for (int k = 0; k < nx; k++) z1[k] = o->x0[k];                                            /* :667 */
    for (int i = 0; i < nn; i++) {                                                            /* :669-681 */
        int t = o->stage[i];
        gemv_n(nu, nx, o->K + (size_t)t * nu * nx, z1 + i * nx, z1 + u_off + i * nu);
        for (int k = 0; k < nu; k++) z1[u_off + i * nu + k] += dd[i * nu + k];
        for (int c = 0; c < d; c++) {
            int j = child0(o, i) + c;
            gemv_n(nx, nx, o->ABK + ((size_t)(t + 1) * d + c) * nx * nx, z1 + i * nx, z1 + j * nx);
            gemv_n(nx, nu, o->B + (size_t)slot_event(o, c) * nx * nu, dd + i * nu, o->mul_x);
            for (int k = 0; k < nx; k++) z1[j * nx + k] += o->mul_x[k];
        }
    }
}

static void projection_S2(orc *o, double *z2) { /* implicit_l.jl:688-727; z2 = (s_2..s_n, tau_2..tau_n, y) */
    int n = o->n, ny = o->ny, d = o->d, m = ny + 2 * d;
    int tau_off = n - 1, y_off = 2 * (n - 1);
    for (int i = 0; i < o->nn; i++) {
        int c1 = child0(o, i) - 1; /* position of the first child in the s / tau sub-blocks */
        memcpy(o->ls_b, z2 + y_off + i * ny, sizeof(double) * ny);
        memcpy(o->ls_b + ny, z2 + c1, sizeof(double) * d);
        memcpy(o->ls_b + ny + d, z2 + tau_off + c1, sizeof(double) * d);
        gemv_n(m, m, mat_sel(o->ls, o->n_rms, i, m * m), o->ls_b, o->ls_b2);   /* ls_matrix[i] (:717) */
        memcpy(z2 + y_off + i * ny, o->ls_b2, sizeof(double) * ny);
        memcpy(z2 + c1, o->ls_b2 + ny, sizeof(double) * d);
        memcpy(z2 + tau_off + c1, o->ls_b2 + ny + d, sizeof(double) * d);
    }
}

void orc_prox_f(orc *o, double *arg, double gamma) { /* implicit_l.jl:729-748 */
    arg[o->zs] -= gamma;
    projection_S1(o, arg);
    projection_S2(o, arg + o->zs + 1);
}
void orc_set_x0(orc *o, const double *x0) { memcpy(o->x0, x0, sizeof(double) * o->nx); }

/* risk_measures.jl:14-71 */
static void project_soc(double *vec, int len) {
    double t = vec[0], x_norm = 0.;
    for (int i = 1; i < len; i++) x_norm += vec[i] * vec[i];
    x_norm = sqrt(x_norm);
    if (x_norm <= t) {
    } else if (x_norm <= -t) {
        for (int k = 0; k < len; k++) vec[k] = 0.;
    } else {
        vec[0] = (t + x_norm) / 2.;
        for (int k = 1; k < len; k++) vec[k] = vec[0] * vec[k] / x_norm;
    }
}
static void project_cone(double *vec, int kind, int dim) {
    switch (kind) {
    case CONE_ZEROS: for (int i = 0; i < dim; i++) vec[i] = 0.; break;
    case CONE_NONNEG: for (int i = 0; i < dim; i++) if (vec[i] < 0.) vec[i] = 0.; break;
    case CONE_NONPOS: for (int i = 0; i < dim; i++) if (vec[i] > 0.) vec[i] = 0.; break;
    case CONE_REALS: break;
    case CONE_SOC: project_soc(vec, dim); break;
    }
}
static int dual_kind(int kind) { /* MOI.dual_set: Zeros <-> Reals, the others are self-dual */
    return kind == CONE_ZEROS ? CONE_REALS : (kind == CONE_REALS ? CONE_ZEROS : kind);
}
static inline double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

static void project_on_leaf_constraints(orc *o, double *arg) { /* implicit_l.jl:752-810 */
    int nx = o->nx;
    double *w = o->proj_leaf;
    for (int i = 0; i < o->nl; i++) {
        w[0] = arg[o->o13 + i];
        for (int j = 0; j < nx; j++) w[j + 1] = arg[o->o11 + i * nx + j];
        w[nx + 1] = arg[o->o12 + i];
        project_soc(w, nx + 2);
        arg[o->o13 + i] = w[0];
        for (int j = 0; j < nx; j++) arg[o->o11 + i * nx + j] = w[j + 1];
        arg[o->o12 + i] = w[nx + 1];
    }
    for (int k = 0; k < o->nl * nx; k++) arg[o->o14 + k] = clampd(arg[o->o14 + k], o->x_min, o->x_max); /* constraints.jl:24-33 */
}
static void project_on_nonleaf_constraints(orc *o, double *arg) { /* implicit_l.jl:812-909 */
    int nx = o->nx, nu = o->nu, ny = o->ny, nn = o->nn, d = o->d;
    for (int i = 0; i < nn; i++) {                                                            /* v1 :833-845 */
        int off = i * ny;
        for (int k = 0; k < o->n_cones; k++) {
            project_cone(arg + off, dual_kind(o->ck[k]), o->cd[k]);
            off += o->cd[k];
        }
    }
    for (int k = 0; k < nn; k++) arg[o->o2 + k] = arg[o->o2 + k] >= 0. ? arg[o->o2 + k] : 0.; /* v2 :849-851 */
    double *w = o->proj_nleaf;
    for (int i = 0; i < nn; i++)                                                              /* v3-v6 :854-900 */
        for (int c = 0; c < d; c++) {
            int j_ind = child0(o, i) + c - 1;
            w[0] = arg[o->o6 + j_ind];
            for (int k = 0; k < nx; k++) w[k + 1] = arg[o->o3 + j_ind * nx + k];
            for (int k = 0; k < nu; k++) w[1 + nx + k] = arg[o->o4 + j_ind * nu + k];
            w[1 + nx + nu] = arg[o->o5 + j_ind];
            project_soc(w, nx + nu + 2);
            arg[o->o6 + j_ind] = w[0];
            for (int k = 0; k < nx; k++) arg[o->o3 + j_ind * nx + k] = w[k + 1];
            for (int k = 0; k < nu; k++) arg[o->o4 + j_ind * nu + k] = w[1 + nx + k];
            arg[o->o5 + j_ind] = w[1 + nx + nu];
        }
    for (int i = 0; i < nn; i++) {                                                            /* v7 :903-908, constraints.jl:38-54 */
        double *xu = arg + o->o7 + i * (nx + nu);
        for (int j = 0; j < nx; j++) xu[j] = clampd(xu[j], o->x_min, o->x_max);
        for (int j = 0; j < nu; j++) xu[nx + j] = clampd(xu[nx + j], o->u_min, o->u_max);
    }
}
void orc_prox_h_conj(orc *o, double *arg, double sigma) { /* implicit_l.jl:911-951 */
    int n1 = o->n - 1, nl = o->nl;
    for (int i = 0; i < o->nv; i++) arg[i] = arg[i] / sigma;
    for (int i = 0; i < n1; i++) arg[o->o5 + i] -= 0.5;
    for (int i = 0; i < n1; i++) arg[o->o6 + i] += 0.5;
    for (int i = 0; i < nl; i++) arg[o->o12 + i] -= 0.5;
    for (int i = 0; i < nl; i++) arg[o->o13 + i] += 0.5;
    memcpy(o->prox_v_wsp, arg, sizeof(double) * o->nv);
    project_on_leaf_constraints(o, arg);
    project_on_nonleaf_constraints(o, arg);
    for (int i = 0; i < o->nv; i++) arg[i] = sigma * (o->prox_v_wsp[i] - arg[i]);
}

/* ---------------------------------------------------------------- CP (cp.jl) */
static double norm_inf(const double *x, int k) {
    double m = 0.;
    for (int i = 0; i < k; i++) { double a = fabs(x[i]); if (a > m || a != a) m = a; }
    return m;
}
/* one evaluation of the CP operator: (zin, vin) -> (zout, vout); cp.jl:5-32 / sp.jl:118-147 */
static void cp_operator(orc *o, const double *zin, const double *vin, double *zout, double *vout, double *zwork, double gamma, double sigma) {
    memcpy(zout, zin, sizeof(double) * o->nz);
    spock_mul(o, zout, 1, vin, -gamma, 1.);
    orc_prox_f(o, zout, gamma);
    memcpy(zwork, zin, sizeof(double) * o->nz);
    for (int i = 0; i < o->nz; i++) zwork[i] = 2. * zout[i] + -1. * zwork[i]; /* BLAS.axpby!(2, zbar, -1, w) */
    memcpy(vout, vin, sizeof(double) * o->nv);
    spock_mul(o, vout, 0, zwork, sigma, 1.);
    orc_prox_h_conj(o, vout, sigma);
}
/* public single-operator entry for tests */
void orc_T(orc *o, const double *zin, const double *vin, double *zout, double *vout, double gamma, double sigma) {
    double *zw = dalloc(o->nz);
    cp_operator(o, zin, vin, zout, vout, zw, gamma, sigma);
    free(zw);
}

typedef struct { double *xi, *xi1, *xi2; int *bt; int cap, len; } orc_hist;

/* cp.jl:54-123 and sp.jl:270-344 (they differ only in the LOG-mode xi) */
static int should_terminate(orc *o, double a1, double a2, double tol, int verbose, int backtrack, orc_hist *h, orc_result *res) {
    for (int i = 0; i < o->nz; i++) o->delta_z[i] = o->z[i] - o->z_old[i];
    for (int i = 0; i < o->nv; i++) o->delta_v[i] = o->v[i] - o->v_old[i];
    memcpy(o->xi_1, o->delta_z, sizeof(double) * o->nz);
    memcpy(o->xi_2, o->delta_v, sizeof(double) * o->nv);
    spock_mul(o, o->xi_1, 1, o->delta_v, 1., -1. / a1);
    spock_mul(o, o->xi_2, 0, o->delta_z, 1., -1. / a2);
    double xi_1 = norm_inf(o->xi_1, o->nz), xi_2 = norm_inf(o->xi_2, o->nv), xi = NAN;
    int r = 0;
    if (verbose == ORC_LOG) {
        if (o->alg == ORC_CP) {
            memcpy(o->xi, o->xi_1, sizeof(double) * o->nz);
            spock_mul(o, o->xi, 1, o->xi_2, 1., 1.);
            xi = norm_inf(o->xi, o->nz);
        } else {
            xi = xi_1 > xi_2 ? xi_1 : xi_2;
        }
        if (xi <= tol) r = 1;
    } else {
        double t1 = tol * o->res_0[0] > tol ? tol * o->res_0[0] : tol;
        double t2 = tol * o->res_0[1] > tol ? tol * o->res_0[1] : tol;
        if (xi_1 <= t1 && xi_2 <= t2) r = 1;
    }
    if (h && h->len < h->cap) {
        if (h->xi) h->xi[h->len] = xi;
        if (h->xi1) h->xi1[h->len] = xi_1;
        if (h->xi2) h->xi2[h->len] = xi_2;
        if (h->bt) h->bt[h->len] = backtrack;
        h->len++;
    }
    if (o->res_0[0] == -INFINITY) o->res_0[0] = xi_1;
    if (o->res_0[1] == -INFINITY) o->res_0[1] = xi_2;
    res->xi1 = xi_1; res->xi2 = xi_2; res->xi = xi;
    return r;
}

static void run_cp(orc *o, int max_iter, double tol, int verbose, double gamma, double sigma, orc_hist *h, orc_result *res) { /* cp.jl:188-232 */
    int iter = 0, evals = 0, conv = 0;
    while (iter < max_iter) {
        memcpy(o->z_old, o->z, sizeof(double) * o->nz);
        memcpy(o->v_old, o->v, sizeof(double) * o->nv);
        cp_operator(o, o->z, o->v, o->zbar, o->vbar, o->z_workspace, gamma, sigma);
        memcpy(o->z, o->zbar, sizeof(double) * o->nz); /* lambda = 1: z <- zbar exactly */
        memcpy(o->v, o->vbar, sizeof(double) * o->nv);
        evals++;
        if (should_terminate(o, gamma, sigma, tol, verbose, 0, h, res)) { conv = 1; break; }
        iter++;
    }
    res->iterations = evals; res->iter_ref = iter; res->status = conv ? 0 : 1;
}

/* ---------------------------------------------------------------- Anderson (anderson.jl:1-123), bug-compatible */
static double col_norm(const double *c, size_t nr) { double s = 0.; for (size_t i = 0; i < nr; i++) s += c[i] * c[i]; return sqrt(s); }
static void anderson(orc *o, int iter) {
    size_t nz = o->nz, nv = o->nv, nr = nz + nv;
    double *MP = o->MP, *MR = o->MR, *Q = o->aa_Q, *R = o->aa_R, *g = o->aa_gamma;
    for (int j = 1; j >= 0; j--) memcpy(MP + (size_t)(j + 1) * nr, MP + (size_t)j * nr, sizeof(double) * nr);     /* :13-17 */
    for (int j = 1; j >= 0; j--) memcpy(MR + (size_t)(j + 1) * nr, MR + (size_t)j * nr, sizeof(double) * nr);     /* :19-23 */
    for (size_t i = 0; i < nz; i++) { MR[i] = o->delta_rz[i]; MP[i] = o->delta_z[i] - MR[i]; }                    /* :25-28 */
    for (size_t i = 0; i < nv; i++) { MR[nz + i] = o->delta_rv[i]; MP[nz + i] = o->delta_v[i] - MR[nz + i]; }     /* :29-32 */
    for (size_t i = 0; i < nz; i++) o->aa_wsp[i] = o->rz[i];                                                      /* :34-39 */
    for (size_t i = 0; i < nv; i++) o->aa_wsp[nz + i] = o->rv[i];
    /* shift R diagonally (:46-50); R is UpperTriangular in the reference: reads below the diagonal give 0 */
    for (int j = 1; j >= 0; j--)
        for (int i = 1; i >= 0; i--)
            R[(i + 1) + (j + 1) * 3] = (i <= j) ? R[i + j * 3] : 0.;
    for (int j = 1; j >= 0; j--) memcpy(Q + (size_t)(j + 1) * nr, Q + (size_t)j * nr, sizeof(double) * nr);       /* :52-56 */
    memcpy(Q, MR, sizeof(double) * nr);                                                                           /* :63-65 */
    if (iter == 1) {                                                                                              /* :68-75 */
        double r = col_norm(Q, nr);
        for (size_t i = 0; i < nr; i++) Q[i] /= r;
        R[0] = r;
    } else {                                                                                                      /* :76-99 */
        int jmax = iter < 3 ? iter : 3;
        for (int j = 1; j < jmax; j++) {
            const double *Qj = Q + (size_t)j * nr;
            double r = 0.;
            for (size_t i = 0; i < nr; i++) r += Qj[i] * Q[i];
            r /= col_norm(Qj, nr);
            R[0 + j * 3] = r;                                   /* written into ROW 1 (anderson.jl:86) */
            for (size_t i = 0; i < nr; i++) Q[i] = Q[i] - r * Qj[i];
        }
        double r = col_norm(Q, nr);
        for (size_t i = 0; i < nr; i++) Q[i] /= r;
        R[0] = r;
    }
    for (int j = 0; j < 3; j++) {                                                                                 /* gamma = Q' wsp :110 */
        const double *Qj = Q + (size_t)j * nr;
        double s = 0.;
        for (size_t i = 0; i < nr; i++) s += Qj[i] * o->aa_wsp[i];
        g[j] = s;
    }
    for (int j = 2; j >= 0; j--) {                                                                                /* ldiv!(UpperTriangular R, gamma) :113 */
        g[j] = g[j] / R[j + j * 3];
        for (int i = j - 1; i >= 0; i--) g[i] -= g[j] * R[i + j * 3];
    }
    for (size_t i = 0; i < nr; i++) {                                                                             /* wsp = MP gamma :116 */
        double s = 0.;
        for (int j = 0; j < 3; j++) s += MP[i + (size_t)j * nr] * g[j];
        o->aa_wsp[i] = s;
    }
    for (size_t i = 0; i < nz; i++) o->dz[i] = -o->rz[i] - o->aa_wsp[i];                                          /* :117-122 */
    for (size_t i = 0; i < nv; i++) o->dv[i] = -o->rv[i] - o->aa_wsp[nz + i];
}

/* ---------------------------------------------------------------- corrected variant (sp_variant = 1; no reference counterpart)
 * What the reference's anderson! / run_sp! would compute without their defects:
 *   - Anderson: gamma = argmin |r - MR gamma|_2 over the min(iter, 3) columns present, by the normal equations MR'MR gamma = MR'r
 *     (3 x 3 Cholesky; a column whose pivot falls below 1e-14 of its diagonal entry is dropped), d = -r - MP gamma.  The
 *     reference orthogonalises the new column against the stored ones but writes the coefficients into row 1 of R instead of
 *     column 1 and divides by column norms that should be 1 (anderson.jl:76-99).
 *   - line search: tau <- beta tau after a rejected candidate (the reference discards the result of perform_linesearch!, sp.jl:439,
 *     and re-evaluates the same candidate nine times).
 * K0 stays disabled as in the reference (sp.jl:80).  The GPU option spock_options_t.sp_variant = 1 implements exactly this. */
void orc_set_sp_variant(orc *o, int variant, int max_backtrack) { o->sp_variant = variant; o->sp_max_backtrack = max_backtrack > 0 ? max_backtrack : 8; }

/* gamma of the normal equations G gamma = rhs on the leading m x m block; G symmetric, column-major 3 x 3 */
static void ls3_solve(int m, const double *G, const double *rhs, double *g) {
    double L[9] = {0}, y[3] = {0, 0, 0};
    int valid[3] = {0, 0, 0};
    for (int j = 0; j < m; j++) {
        double s = G[j + 3 * j];
        for (int k = 0; k < j; k++) if (valid[k]) s -= L[j + 3 * k] * L[j + 3 * k];
        if (!(s > 1e-14 * G[j + 3 * j])) continue;
        valid[j] = 1;
        L[j + 3 * j] = sqrt(s);
        for (int i = j + 1; i < m; i++) {
            double t = G[i + 3 * j];
            for (int k = 0; k < j; k++) if (valid[k]) t -= L[i + 3 * k] * L[j + 3 * k];
            L[i + 3 * j] = t / L[j + 3 * j];
        }
    }
    for (int j = 0; j < m; j++) {
        if (!valid[j]) continue;
        double t = rhs[j];
        for (int k = 0; k < j; k++) if (valid[k]) t -= L[j + 3 * k] * y[k];
        y[j] = t / L[j + 3 * j];
    }
    for (int j = 2; j >= 0; j--) {
        g[j] = 0.;
        if (j >= m || !valid[j]) continue;
        double t = y[j];
        for (int k = j + 1; k < m; k++) if (valid[k]) t -= L[k + 3 * j] * g[k];
        g[j] = t / L[j + 3 * j];
    }
}

static void anderson_ls(orc *o, int iter) {
    size_t nz = o->nz, nv = o->nv, nr = nz + nv;
    double *MP = o->MP, *MR = o->MR, *g = o->aa_gamma;
    for (int j = 1; j >= 0; j--) memcpy(MP + (size_t)(j + 1) * nr, MP + (size_t)j * nr, sizeof(double) * nr);
    for (int j = 1; j >= 0; j--) memcpy(MR + (size_t)(j + 1) * nr, MR + (size_t)j * nr, sizeof(double) * nr);
    for (size_t i = 0; i < nz; i++) { MR[i] = o->delta_rz[i]; MP[i] = o->delta_z[i] - MR[i]; }
    for (size_t i = 0; i < nv; i++) { MR[nz + i] = o->delta_rv[i]; MP[nz + i] = o->delta_v[i] - MR[nz + i]; }
    for (size_t i = 0; i < nz; i++) o->aa_wsp[i] = o->rz[i];
    for (size_t i = 0; i < nv; i++) o->aa_wsp[nz + i] = o->rv[i];
    const int m = iter < 3 ? iter : 3;
    double G[9], rhs[3];
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b <= a; b++) {
            double s = 0.;
            for (size_t i = 0; i < nr; i++) s += MR[i + (size_t)a * nr] * MR[i + (size_t)b * nr];
            G[a + 3 * b] = s; G[b + 3 * a] = s;
        }
        double s = 0.;
        for (size_t i = 0; i < nr; i++) s += MR[i + (size_t)a * nr] * o->aa_wsp[i];
        rhs[a] = s;
    }
    ls3_solve(m, G, rhs, g);
    for (size_t i = 0; i < nr; i++) {
        double s = 0.;
        for (int j = 0; j < 3; j++) s += MP[i + (size_t)j * nr] * g[j];
        o->aa_wsp[i] = s;
    }
    for (size_t i = 0; i < nz; i++) o->dz[i] = -o->rz[i] - o->aa_wsp[i];
    for (size_t i = 0; i < nv; i++) o->dv[i] = -o->rv[i] - o->aa_wsp[nz + i];
}

/* ---------------------------------------------------------------- SP (sp.jl:358-469) */
static void run_sp(orc *o, int max_iter, double tol, int verbose, double gamma, double sigma, orc_hist *h, orc_result *res) {
    const double c1 = 0.99, q = 0.99, sigma_k2 = 0.1, lambda_sp = 1.;
    const int MAX_BACKTRACK = (o->sp_variant == 1 && o->sp_max_backtrack > 0) ? o->sp_max_backtrack : 8;
    const double beta = 0.5;                                                                         /* sp.jl:370 */
    int nz = o->nz, nv = o->nv;
    int iter = 0, evals = 0, conv = 0, broyden_k = 0;
    double rnorm, rtilde_norm, r_safe = INFINITY, tau;
    while (iter < max_iter) {
        cp_operator(o, o->z, o->v, o->zbar, o->vbar, o->z_workspace, gamma, sigma);              /* :392-393 */
        for (int i = 0; i < nz; i++) o->rz[i] = o->z[i] - o->zbar[i];                               /* :34-51 */
        for (int i = 0; i < nv; i++) o->rv[i] = o->v[i] - o->vbar[i];
        rnorm = spock_norm(o, o->rz, o->rv, gamma, sigma);
        for (int i = 0; i < nz; i++) o->delta_rz[i] = o->rz[i] - o->rz_old[i];                      /* :257-268 */
        for (int i = 0; i < nv; i++) o->delta_rv[i] = o->rv[i] - o->rv_old[i];
        memcpy(o->z_old, o->z, sizeof(double) * nz); memcpy(o->v_old, o->v, sizeof(double) * nv);    /* :232-255 */
        memcpy(o->rz_old, o->rz, sizeof(double) * nz); memcpy(o->rv_old, o->rv, sizeof(double) * nv);
        broyden_k += 1;                                                                              /* models/spock.jl:212-224 */
        if (o->sp_variant == 1) anderson_ls(o, broyden_k); else anderson(o, broyden_k);
        tau = 1.;
        int backtrack_count = 0, loop = 1;
        /* K0 is unreachable: should_perform_k0 returns `... && false` (sp.jl:80) */
        while (loop && backtrack_count <= MAX_BACKTRACK) {                                           /* :411 */
            for (int i = 0; i < nz; i++) o->w[i] = o->z[i] + tau * o->dz[i];                         /* :118-131 */
            for (int i = 0; i < nv; i++) o->u[i] = o->v[i] + tau * o->dv[i];
            cp_operator(o, o->w, o->u, o->wbar, o->ubar, o->w_workspace, gamma, sigma);             /* :133-164 */
            for (int i = 0; i < nz; i++) o->rw[i] = o->w[i] - o->wbar[i];
            for (int i = 0; i < nv; i++) o->ru[i] = o->u[i] - o->ubar[i];
            rtilde_norm = spock_norm(o, o->rw, o->ru, gamma, sigma);
            if (rnorm <= r_safe && rtilde_norm <= c1 * rnorm) {                                      /* K1 :166-191 */
                memcpy(o->z, o->w, sizeof(double) * nz); memcpy(o->v, o->u, sizeof(double) * nv);
                r_safe = rtilde_norm + pow(q, iter);
                loop = 0;
                continue;
            }
            for (int k = 0; k < nz; k++) o->rho_z[k] = o->rw[k] - tau * o->dz[k];                    /* :420-434 */
            for (int k = 0; k < nv; k++) o->rho_v[k] = o->ru[k] - tau * o->dv[k];
            double rho = orc_dot(o, o->rw, o->ru, o->rho_z, o->rho_v, gamma, sigma);
            if (rho >= sigma_k2 * rnorm * rtilde_norm) {                                             /* K2 :193-222 */
                rho /= rtilde_norm * rtilde_norm;
                for (int i = 0; i < nz; i++) o->z[i] -= lambda_sp * rho * o->rw[i];
                for (int i = 0; i < nv; i++) o->v[i] -= lambda_sp * rho * o->ru[i];
                loop = 0;
                continue;
            }
            /* perform_linesearch!(tau, beta): the returned tau*beta is discarded (sp.jl:439) -> tau stays 1 */
            if (o->sp_variant == 1) tau *= beta;                                                     /* corrected variant: it is kept */
            backtrack_count += 1;
        }
        if (loop) {                                                                                  /* :443-446 */
            memcpy(o->z, o->zbar, sizeof(double) * nz); memcpy(o->v, o->vbar, sizeof(double) * nv);
        }
        memcpy(o->dzo, o->delta_z, sizeof(double) * nz); memcpy(o->dvo, o->delta_v, sizeof(double) * nv);       /* :346-356 */
        memcpy(o->drzo, o->delta_rz, sizeof(double) * nz); memcpy(o->drvo, o->delta_rv, sizeof(double) * nv);
        evals++;
        if (should_terminate(o, gamma, sigma, tol, verbose, backtrack_count, h, res)) { conv = 1; break; }
        iter++;
    }
    res->iterations = evals; res->iter_ref = iter; res->status = conv ? 0 : 1;
}

/* solve_model! (cpock.jl:169-189, models/spock.jl:231-251).  gamma/sigma: both must be finite to override
   the default 0.99/sqrt(L_norm) (cp.jl:198-201).  max_iter <= 0 selects the reference default (5000 / 1000). */
int orc_solve(orc *o, const double *x0, double tol, int verbose, const double *z0, const double *v0,
              double gamma, double sigma, int max_iter,
              double *hist_xi, double *hist_xi1, double *hist_xi2, int *hist_bt, int hist_cap, int *hist_len,
              orc_result *res) {
    if (z0 || v0) {
        if (!z0 || !v0) return -1; /* copyto!(.., nothing) throws in the reference */
        memcpy(o->z, z0, sizeof(double) * o->nz); memcpy(o->v, v0, sizeof(double) * o->nv);
    }
    memcpy(o->x0, x0, sizeof(double) * o->nx);
    o->res_0[0] = -INFINITY; o->res_0[1] = -INFINITY;
    if (!(isfinite(gamma) && isfinite(sigma))) { sigma = 0.99 / sqrt(o->L_norm); gamma = sigma; }
    orc_hist h = { hist_xi, hist_xi1, hist_xi2, hist_bt, hist_cap, 0 };
    orc_result r = { 0, 0, 0, NAN, NAN, NAN };
    if (o->alg == ORC_CP) run_cp(o, max_iter > 0 ? max_iter : 5000, tol, verbose, gamma, sigma, &h, &r);
    else run_sp(o, max_iter > 0 ? max_iter : 1000, tol, verbose, gamma, sigma, &h, &r);
    if (hist_len) *hist_len = h.len;
    if (res) *res = r;
    return 0;
}
