// spock_device.hpp -- the per-thread bodies of the fused Chambolle-Pock kernels.
//
// One thread owns one NON-LEAF tree node i together with the edge quantities of its d children
// (and, when the children are leaves, the leaves themselves).  With that ownership every piece of the
// CP operator T(z, v) = (prox_f(z - gamma L'v), prox_h*(v + sigma L(2 zbar - z))) except the S1 tree
// sweep is thread-local:
//   phase A  a_local   : zbar0 = z - gamma L'v on the owned entries  (L_transpose!, implicit_l.jl:320-449)
//                        + risk-kernel projection S2                 (projection_S2!, implicit_l.jl:688-727)
//   sweep    bwd_level : Riccati-style backward recursion of S1      (projection_S1!, implicit_l.jl:584-637)
//            fwd_level : forward roll-out                            (implicit_l.jl:667-681)
//   phase B  b_local   : L(2 zbar - z) (L!, implicit_l.jl:177-318), Moreau step around the cone / box
//                        projections (prox_h_conj!, implicit_l.jl:911-951, :752-909,
//                        risk_measures.jl:14-71, constraints.jl:24-54) and the residuals of
//                        should_terminate! (cp.jl:54-123)
// The sweeps exchange q (child -> parent) and x (parent -> child) through LDS between tree levels.
//
// This header compiles for gfx950 (included by spock_kernels.hip) and, with SPOCK_HOST_EMU defined,
// as plain C++ for the CPU-side debugging harness under tests/emu (never part of the product).
#pragma once
#include <cmath>

#include "spock_internal.hpp"

#ifdef SPOCK_HOST_EMU
#define SPK_DEV inline
#define SPK_UNROLL
#else
#define SPK_DEV __device__ __forceinline__
#define SPK_UNROLL _Pragma("unroll P::UNR")
#endif
#ifdef SPOCK_HOST_EMU
#define SPK_UNROLL4
#define SPK_UNROLL8
#else
#define SPK_UNROLL4 _Pragma("unroll 4")   // run-time trip counts: keep four independent loads in flight
#define SPK_UNROLL8 _Pragma("unroll 8")   // ... eight: rows of the small per-stage matrices fetched from L2 inside a mat-vec loop
#endif

namespace spock {

// The matrix pool is read-only for the lifetime of a handle.  Addressing it through the CONSTANT address
// space lets the compiler fetch every wave-uniform matrix element with scalar loads (s_load_dwordx16 into
// SGPRs that feed v_fma_f64 directly) instead of 64 identical per-lane vector loads; per-lane (divergent)
// indices still work and fall back to vector loads.
#ifdef SPOCK_HOST_EMU
typedef const real *MatPtr;
#else
typedef const __attribute__((address_space(4))) real *MatPtr;
#endif

// cost-matrix mode (template parameter CM of the dimension policies)
constexpr int CM_GENERAL = 0;      // per-node sqrtQ / sqrtR / sqrtQN (indexed by node: divergent loads)
constexpr int CM_SHARED = 1;       // one matrix of each kind shared by every node (uniform: scalar loads)
constexpr int CM_SHARED_DIAG = 2;  // shared and diagonal: the mat-vecs are element-wise products

// For finite data the element-wise (diagonal) product is bit-identical to the dense one: the skipped terms
// are exact zeros.
template <int NX, int NU, int D, int NY, int CM>
struct StaticDims {
    static constexpr int MAXNX = NX, MAXNU = NU, MAXD = D, MAXNY = NY, MAXM = NY + 2 * D, UNR = 128;
    static constexpr bool DIAG = (CM == CM_SHARED_DIAG), SHARED = (CM != CM_GENERAL);
    static constexpr bool REGROWS = (NX <= 8 && NU <= 8);   // register-resident team sweep available
    static constexpr bool TEAMLOCAL = false, MFMA_SWEEP = false;
    SPK_DEV explicit StaticDims(const Dims &) {}
    SPK_DEV constexpr int nx() const { return NX; }
    SPK_DEV constexpr int nu() const { return NU; }
    SPK_DEV constexpr int d() const { return D; }
    SPK_DEV constexpr int ny() const { return NY; }
};
// TL: lane-team node-local phases (spock_team_local.hpp) instead of the per-thread ones: no per-thread vectors at all
// MF: the sweep levels run on the matrix cores (bwd_level_mfma / fwd_level_mfma; nx = 64, nu = 16, d = 2).  A policy of its
// own: the fragment registers of those levels must not set the register allocation of the other shapes' kernels.
template <int CM, bool TL = false, bool MF = false>
struct DynDims {
    static constexpr int MAXNX = TL ? 1 : kDynMaxNx, MAXNU = TL ? 1 : kDynMaxNu, MAXD = kDynMaxD, MAXNY = TL ? 1 : kDynMaxNy,
                         MAXM = TL ? 1 : kDynMaxNy + 2 * kDynMaxD, UNR = 1;
    static constexpr bool DIAG = (CM == CM_SHARED_DIAG), SHARED = (CM != CM_GENERAL);
    static constexpr bool REGROWS = false, TEAMLOCAL = TL, MFMA_SWEEP = MF;
    int nx_, nu_, d_, ny_;
    SPK_DEV explicit DynDims(const Dims &dm) : nx_(dm.nx), nu_(dm.nu), d_(dm.d), ny_(dm.ny) {}
    SPK_DEV int nx() const { return nx_; }
    SPK_DEV int nu() const { return nu_; }
    SPK_DEV int d() const { return d_; }
    SPK_DEV int ny() const { return ny_; }
};

// Lane-team policy with the branching factor at COMPILE time, for the flat node-local kernels (spock_flat_local.hpp): the loops over
// the children unroll and the per-child values a lane carries from part 1 to part 2 of a phase (TLReg) stay in registers.  With a
// run-time d they are dynamically indexed arrays, i.e. scratch memory -- 672 B per lane in the first build of those kernels, and
// the reason the lane-team phases of the subtree kernels were slow.  nx, nu, ny stay run-time values (no per-lane vectors exist).
template <int D>
struct FlatDims {
    static constexpr int MAXNX = 1, MAXNU = 1, MAXD = D, MAXNY = 1, MAXM = 1, UNR = 8;
    static constexpr bool DIAG = true, SHARED = true, REGROWS = false, TEAMLOCAL = true, MFMA_SWEEP = false;
    int nx_, nu_, ny_;
    SPK_DEV explicit FlatDims(const Dims &dm) : nx_(dm.nx), nu_(dm.nu), ny_(dm.ny) {}
    SPK_DEV int nx() const { return nx_; }
    SPK_DEV int nu() const { return nu_; }
    SPK_DEV constexpr int d() const { return D; }
    SPK_DEV int ny() const { return ny_; }
};

// lane-team node-local phases (spock_team_local.hpp): per-lane registers carried from part 1 to part 2 of a phase
template <class P>
struct TLReg {
    real xb, ub;                        // A: xbar_r, ubar_r         B: x_i[r] - z (dz), see below
    real wx, dzx, wu, dzu, wy, dzy;     // B: w = 2 zbar - z and dz = zbar - z on x_i[r], u_i[r], y_i[r]
    real t3[P::MAXD], l3[P::MAXD], t4[P::MAXD], l4[P::MAXD];       // child blocks: pre-prox value / (L dz)
    real t11[P::MAXD], l11[P::MAXD], t14[P::MAXD], wl[P::MAXD], dl[P::MAXD];   // leaf blocks
};

constexpr int kStageRegs = 4;   // sweep inputs a thread can hold in registers while the control block is in flight

// Per-thread state that lives across the phases of one kernel (registers on the GPU).
template <class P>
struct TState {
    int lvl, pos, node, c0;   // level and position inside the group, global node, first child
    int stage;                // global stage of the node
    bool active;
    bool kids_leaf;           // children are leaves of the TREE (this thread also owns them)
    bool kids_gleaf;          // children are leaves of the GROUP (crown: the subtree roots)
    real xb[P::MAXNX];      // phase A: c_i = U ubar_i - xbar_i  -> x_i (after the forward sweep)
    real ub[P::MAXNU];      // phase A: e_i = Rt^-1 ubar_i       -> u_i
    real dd[P::MAXNU];      // d_i of the Riccati recursion
    real xl[P::MAXD][P::MAXNX];  // leaf children: xbar_j -> x_j
    real stg[kStageRegs];        // early-issued loads of the sweep inputs (spock_group.hpp)
    TLReg<P> tlr;                  // lane-team local phases
    real mx3[3];                 // residual maxima accumulated over the rounds of the lane-team phase B
    real sp3[3];                 // ... and the SP inner products (tb_p2 SPM)
    // register-resident lane-team sweep (small compile-time shapes): this thread = (node tn_li, matrix row tn_r)
    // and keeps ITS rows of the sweep matrices of its node's stage for the whole kernel
    int tn_li, tn_r, tn_lvl, tn_pos;
    real tW[P::REGROWS ? P::MAXD : 1][P::REGROWS ? P::MAXNX : 1], tV[P::REGROWS ? P::MAXD : 1][P::REGROWS ? P::MAXNX : 1];
    real tA[P::REGROWS ? P::MAXD : 1][P::REGROWS ? P::MAXNX : 1], tB[P::REGROWS ? P::MAXD : 1][P::REGROWS ? P::MAXNU : 1];
    real tK[P::REGROWS ? P::MAXNX : 1];
};

template <class T>
SPK_DEV T spk_max_nan(T m, T a) { return (a > m || a != a) ? a : m; }  // NaN-propagating max
SPK_DEV real spk_clamp(real x, real lo, real hi) { return x > hi ? hi : (x < lo ? lo : x); }

// event of child slot c: w = (j % d) + 1 with j = 2 + (i-1) d + c, 1-based (scenario_tree.jl:77)
SPK_DEV int slot_event(int c, int d) { return (2 + c) % d; }

// y (+)= S x for a symmetric cost square root S (n x n)
template <class P, bool ACC, class XV>
SPK_DEV void sym_mv(int n, MatPtr S, XV x, real *y) {
    SPK_UNROLL for (int r = 0; r < n; ++r) {
        real t;
        if (P::DIAG) t = S[r + r * n] * x[r];
        else {
            t = 0.;
            SPK_UNROLL for (int k = 0; k < n; ++k) t += S[r + k * n] * x[k];
        }
        if (ACC) y[r] += t; else y[r] = t;
    }
}

// (t; x) second-order-cone projection, MOI ordering t first (risk_measures.jl:47-71)
template <int MAXLEN>
SPK_DEV void project_soc(real *vec, int len) {
    real t = vec[0], s = 0.;
    for (int i = 1; i < len; ++i) s += vec[i] * vec[i];
    real x_norm = sqrt(s);
    if (x_norm <= t) {
    } else if (x_norm <= -t) {
        for (int k = 0; k < len; ++k) vec[k] = 0.;
    } else {
        const real t2 = (t + x_norm) / 2.;
        const real sc = t2 / x_norm;   // the reference evaluates (t2 * vec[k]) / x_norm: same up to one rounding
        vec[0] = t2;
        for (int k = 1; k < len; ++k) vec[k] = sc * vec[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Pointer-like views.  The node-local bodies below are written against `operator[]` on their vector arguments, so the same
// code runs on two layouts of the iterate:
//   * instance-major  z[b][nz]  (plain pointers): one THREAD per tree node -- latency regime, operators, SP
//   * element-major   z[e][Bp]  (IVec): one LANE per problem instance, one WAVE per tree node -- the batched regime of
//     BASELINE config 2, where every load / store of a wave is 512 contiguous bytes and every matrix element is wave-uniform
// IVec element e of this lane's instance lives at base[e * stride] + lane, `base` wave-uniform.
#ifdef SPOCK_HOST_EMU
struct IVec {
    real *base;
    long stride;
    unsigned lb;
    SPK_DEV real &operator[](long e) const { return *(real *)((char *)(base + e * stride) + lb); }
    SPK_DEV IVec operator+(long off) const { return IVec{base + off * stride, stride, lb}; }
    SPK_DEV explicit operator bool() const { return base != nullptr; }
};
// view of `n_elems` element rows starting at the tile's first column `base_tile`, for lane `lane`
SPK_DEV IVec ivec_make(const real *base_tile, long stride, int lane, long) { return IVec{const_cast<real *>(base_tile), stride, (unsigned)lane * (unsigned)sizeof(real)}; }
SPK_DEV IVec ivec_null() { return IVec{nullptr, 0, 0}; }
#else
// gfx950: buffer addressing.  address = resource base (4 SGPRs, the array's tile column) + soffset (one SGPR: the element row,
// e * stride * 8 bytes, scalar arithmetic) + voffset (one VGPR for the whole kernel: lane * 8): no per-row vector or 64-bit
// scalar address arithmetic at all.  Offsets are 32-bit: an array must stay below 4 GiB (checked at spock_create).
typedef unsigned int spk_u32x2 __attribute__((ext_vector_type(2)));
template <class T> SPK_DEV T spk_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> SPK_DEV double spk_buf_load<double>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
template <> SPK_DEV float spk_buf_load<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <class T> SPK_DEV void spk_buf_store(T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> SPK_DEV void spk_buf_store<double>(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(spk_u32x2, v), r, voff, soff, 0);
}
template <> SPK_DEV void spk_buf_store<float>(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
struct IBufRef {
    __amdgpu_buffer_rsrc_t r;
    unsigned voff, soff;
    SPK_DEV operator real() const { return spk_buf_load<real>(r, voff, soff); }
    SPK_DEV void operator=(real v) const { spk_buf_store<real>(v, r, voff, soff); }
};
struct IVec {
    __amdgpu_buffer_rsrc_t r;
    unsigned stride8, voff, soff;
    bool ok;
    SPK_DEV IBufRef operator[](long e) const { return IBufRef{r, voff, soff + (unsigned)e * stride8}; }
    SPK_DEV IVec operator+(long off) const { return IVec{r, stride8, voff, soff + (unsigned)off * stride8, ok}; }
    SPK_DEV explicit operator bool() const { return ok; }
};
SPK_DEV IVec ivec_make(const real *base_tile, long stride, int lane, long n_elems) {
    const unsigned long bytes = ((unsigned long)(n_elems > 0 ? n_elems - 1 : 0) * (unsigned long)stride + 64ul) * sizeof(real);
    return IVec{__builtin_amdgcn_make_buffer_rsrc((void *)base_tile, 0, (int)(bytes > 0xfffffffful ? 0xfffffffful : bytes), 0x00020000),
                (unsigned)stride * (unsigned)sizeof(real), (unsigned)lane * (unsigned)sizeof(real), 0u, base_tile != nullptr};
}
SPK_DEV IVec ivec_null() { return IVec{__builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000), 0u, 0u, 0u, false}; }
#endif
typedef IVec ICVec;
// LDS slots of the sweep: element e of this lane at p[e * 64]
struct LVec {
    real *p;
    SPK_DEV real &operator[](long e) const { return p[e * 64]; }
    SPK_DEV LVec operator+(long off) const { return LVec{p + off * 64}; }
    SPK_DEV explicit operator bool() const { return p != nullptr; }
};
// Block fence of the element-major instantiation: the node-local bodies are one long straight-line block there, and the
// machine scheduler, left alone, hoists every load of a node to the top (hundreds of live registers, scalar and vector).  A
// scheduling barrier between the blocks of a phase keeps a block's loads next to its arithmetic; memory-level parallelism then
// comes from the waves a CU holds rather than from one wave's depth.  No effect on the one-thread-per-node instantiations.
template <class V> struct IsIlvView { static constexpr bool value = false; };
template <> struct IsIlvView<IVec> { static constexpr bool value = true; };
template <class V>
SPK_DEV void spk_block_fence() {
#ifndef SPOCK_HOST_EMU
    if constexpr (IsIlvView<V>::value) __builtin_amdgcn_sched_barrier(0);
#endif
}

// Residual maxima |xi|_inf with NaN propagation (a NaN residual must end the solve with SPOCK_NAN).  One-thread-per-node kernels:
// a NaN-propagating compare-and-select per entry (spk_max_nan).  Element-major kernels, where the vector ALU is the scarcer
// resource: a plain maximum (one instruction) plus one FMA per entry into a `poison` accumulator that turns NaN as soon as any
// entry is NaN or infinite (0 * x); the workgroup's maxima are replaced by NaN when it did (spock_igroup.hpp: reduce_maxima).
template <class V>
SPK_DEV void spk_resmax(real &m, real x, real &poison) {
#ifndef SPOCK_HOST_EMU
    if constexpr (IsIlvView<V>::value) {
        m = fmax(m, fabs(x));
        poison = fma(x, (real)0., poison);
        return;
    }
#endif
    m = (fabs(x) > m || x != x) ? fabs(x) : m;
}

// vbar.v2 of a node's children handed over in registers (crown chunks of the element-major fused kernel)
template <int MAXD>
struct V2Reg {
    real v[MAXD];
    int base;
    SPK_DEV real operator[](int i) const {
        real r = v[0];
        for (int c = 1; c < MAXD; ++c) if (i - base == c) r = v[c];
        return r;
    }
};
SPK_DEV bool v2x_applies(const real *p, bool kids_gleaf) { return p && kids_gleaf; }
template <int MAXD>
SPK_DEV bool v2x_applies(const V2Reg<MAXD> &, bool) { return true; }

// ------------------------------------------------------------------------------------------------
// group geometry: thread tau <-> local BFS index tau of a complete d-ary group tree
template <class P>
SPK_DEV void init_geometry(const P &p, const Dims &dm, int tau, int root_stage, int root_pos, int stages, bool crown,
                           TState<P> &s) {
    const int d = p.d();
    int lvl = 0, first = 0, cnt = 1;
    while (lvl < stages - 1 && tau >= first + cnt) { first += cnt; cnt *= d; ++lvl; }
    s.active = (lvl < stages - 1);
    s.lvl = lvl;
    s.pos = tau - first;
    if (!s.active) { s.pos = 0; s.lvl = 0; }
    int span = 1;
    for (int k = 0; k < s.lvl; ++k) span *= d;
    s.node = dm.first[root_stage + s.lvl] + root_pos * span + s.pos;
    s.c0 = s.node * d + 1;
    s.stage = root_stage + s.lvl;
    s.kids_gleaf = s.active && (s.lvl == stages - 2);
    s.kids_leaf = s.kids_gleaf && !crown && (root_stage + stages == dm.N);
    if (crown) s.kids_leaf = false;
}

// ------------------------------------------------------------------------------------------------
// phase A, node-local:  zbar0 = (-gamma) L'v + z on the entries owned by node i, then S2, then the
// node-local terms of the S1 recursion: c_i = U_t ubar_i - xbar_i and e_i = Rt^-1 ubar_i (spock_host.cpp).
template <class P, class ZO>
SPK_DEV void a_finish(const P &p, const Dims &dm, MatPtr M, ZO zo, TState<P> &s, const real *xbar, const real *ubar, const real *vec);

template <class P, class ZI, class VI, class ZO, class V2>
SPK_DEV void a_local_g(const P &p, const Dims &dm, MatPtr M, ZI z, VI v, ZO zo, TState<P> &s, real gamma, unsigned flags, V2 v2x) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny();
    const int i = s.node, nxu = nx + nu;
    const real mg = -gamma;
    const bool raw = (flags & F_RAW_LT) != 0;
    real xbar[P::MAXNX], ubar[P::MAXNU];
    // x_i <- Gamma' v7 + sum_j sqrtQ_j v3_j  (implicit_l.jl:351-373)
    {
        real acc[P::MAXNX];
        SPK_UNROLL for (int k = 0; k < nx; ++k) acc[k] = v[dm.o7 + i * nxu + k];
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int jn = s.c0 + c - 1;
            sym_mv<P, true>(nx, M + dm.mSqQ + (P::SHARED ? 0 : jn) * nx * nx, v + dm.o3 + jn * nx, acc);
        }
        SPK_UNROLL for (int k = 0; k < nx; ++k) xbar[k] = mg * acc[k] + z[dm.zx + i * nx + k];
    }
    spk_block_fence<ZI>();
    // u_i <- Gamma' v7 + sum_j sqrtR_j v4_j  (implicit_l.jl:399-420)
    {
        real acc[P::MAXNU];
        SPK_UNROLL for (int k = 0; k < nu; ++k) acc[k] = v[dm.o7 + i * nxu + nx + k];
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int jn = s.c0 + c - 1;
            sym_mv<P, true>(nu, M + dm.mSqR + (P::SHARED ? 0 : jn) * nu * nu, v + dm.o4 + jn * nu, acc);
        }
        SPK_UNROLL for (int k = 0; k < nu; ++k) ubar[k] = mg * acc[k] + z[dm.zu + i * nu + k];
    }
    spk_block_fence<ZI>();
    // [y_i ; s_children ; tau_children]  (implicit_l.jl:423-448)
    real vec[P::MAXM];
    const real v2i = v[dm.o2 + i];
    const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? i * ny : 0);          // rms[i].b (implicit_l.jl:427)
    SPK_UNROLL for (int k = 0; k < ny; ++k) vec[k] = mg * (v[i * ny + k] - bb[k] * v2i) + z[dm.zy + i * ny + k];
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const int j = s.c0 + c;
        real ls;
        if (v2x_applies(v2x, s.kids_gleaf)) ls = v2x[s.pos * d + c];   // crown workgroup of the fused B+A kernel, children = subtree roots
        else if (j < dm.nn) ls = v[dm.o2 + j];
        else ls = 0.5 * v[dm.o12 + (j - dm.nn)] + 0.5 * v[dm.o13 + (j - dm.nn)];
        vec[ny + c] = mg * ls + z[dm.zs + j];
        vec[ny + d + c] = mg * (0.5 * v[dm.o5 + j - 1] + 0.5 * v[dm.o6 + j - 1]) + z[dm.zt + j - 1];
    }
    spk_block_fence<ZI>();
    // leaf children: x_j <- v14_j + sqrtQN v11_j  (implicit_l.jl:376-396)
    if (s.kids_leaf) {
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int j = s.c0 + c, li = j - dm.nn;
            real t[P::MAXNX];
            sym_mv<P, false>(nx, M + dm.mSqQN + (P::SHARED ? 0 : li) * nx * nx, v + dm.o11 + li * nx, t);
            SPK_UNROLL for (int r = 0; r < nx; ++r) s.xl[c][r] = mg * (v[dm.o14 + li * nx + r] + t[r]) + z[dm.zx + j * nx + r];
        }
    }
    if (raw) {  // operator L': store every owned entry as is
        SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + i * nx + k] = xbar[k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) zo[dm.zu + i * nu + k] = ubar[k];
        SPK_UNROLL for (int k = 0; k < ny; ++k) zo[dm.zy + i * ny + k] = vec[k];
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            const int j = s.c0 + c;
            zo[dm.zs + j] = vec[ny + c];
            zo[dm.zt + j - 1] = vec[ny + d + c];
            if (s.kids_leaf) { SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + j * nx + k] = s.xl[c][k]; }
        }
        if (i == 0) zo[dm.zs] = mg * v[dm.o2] + z[dm.zs];
        return;
    }
    // s_1: (z - gamma L'v)[s_1] - gamma  (prox_f!, implicit_l.jl:741)
    if (i == 0) zo[dm.zs] = (mg * v[dm.o2] + z[dm.zs]) - gamma;
    spk_block_fence<ZI>();
    a_finish(p, dm, M, zo, s, xbar, ubar, vec);
}

// second half of phase A: S2 on [y_i; s_ch; tau_ch] = vec (its part of the next zbar goes to zo), then the node-local terms of
// the S1 recursion c_i -> s.xb, e_i -> s.ub from xbar, ubar (the leaves' xbar stays in s.xl)
template <class P, class ZO>
SPK_DEV void a_finish_s2(const P &p, const Dims &dm, MatPtr M, ZO zo, const TState<P> &s, const real *vec) {
    const int d = p.d(), ny = p.ny();
    const int i = s.node;
    // S2: projector onto ker [E' -I -I ; F' 0 0] applied to [y_i; s_ch; tau_ch]  (implicit_l.jl:708-726)
    const int m = ny + 2 * d;
    const MatPtr LS = M + dm.mLs + (dm.nRm > 1 ? i * m * m : 0);       // ls_matrix[i] (implicit_l.jl:717)
    SPK_UNROLL for (int r = 0; r < m; ++r) {
        real t = 0.;
        SPK_UNROLL for (int k = 0; k < m; ++k) t += LS[r + k * m] * vec[k];
        if (r < ny) zo[dm.zy + i * ny + r] = t;
        else if (r < ny + d) zo[dm.zs + s.c0 + (r - ny)] = t;
        else zo[dm.zt + s.c0 + (r - ny - d) - 1] = t;
    }
}
template <class P>
SPK_DEV void a_finish_ce(const P &p, const Dims &dm, MatPtr M, TState<P> &s, const real *xbar, const real *ubar) {
    const int nx = p.nx(), nu = p.nu();
    // node-local terms of the S1 recursion
    const MatPtr Ut = M + dm.mU + s.stage * nx * nu;
    const MatPtr Ri = M + dm.mRinv + s.stage * nu * nu;
    SPK_UNROLL for (int r = 0; r < nx; ++r) {
        real t = 0.;
        SPK_UNROLL for (int k = 0; k < nu; ++k) t += Ut[r + k * nx] * ubar[k];
        s.xb[r] = t - xbar[r];
    }
    SPK_UNROLL for (int r = 0; r < nu; ++r) {
        real t = 0.;
        SPK_UNROLL for (int k = 0; k < nu; ++k) t += Ri[r + k * nu] * ubar[k];
        s.ub[r] = t;
    }
}
template <class P, class ZO>
SPK_DEV void a_finish(const P &p, const Dims &dm, MatPtr M, ZO zo, TState<P> &s, const real *xbar, const real *ubar, const real *vec) {
    a_finish_s2(p, dm, M, zo, s, vec);
    spk_block_fence<ZO>();
    a_finish_ce(p, dm, M, s, xbar, ubar);
}

// plain-pointer form (instance-major layout): the restrict qualifiers tell the compiler that the iterate it reads and the
// one it writes do not overlap, which lets it issue the loads of a phase ahead of that phase's stores
template <class P>
SPK_DEV void a_local(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z,
                     const real *__restrict__ v, real *__restrict__ zo, TState<P> &s, real gamma, unsigned flags,
                     const real *v2x = nullptr) {
    a_local_g(p, dm, M, z, v, zo, s, gamma, flags, v2x);
}

// ------------------------------------------------------------------------------------------------
// one level of the backward recursion (implicit_l.jl:584-637, restructured: see spock_host.cpp) for the
// thread's node at global stage t:  q_i = c_i + sum_j W_{t,j} q_j,  d_i = e_i - sum_j V_{t,j} q_j.
//   ex    : LDS exchange buffer, [group non-leaf local index][nx]
//   qg    : global q of the group leaves when they are subtree roots (crown), else nullptr
//   q_out : where the group root's q goes (subtree -> global), or nullptr
//   SW    : the sweep block of stage t ([W | V | ABK | K], spock_host.cpp) -- in LDS (const real *) or in the
//           constant-address-space pool (MatPtr)
template <class P, class MPT, class EX, class QG, class QO>
SPK_DEV void bwd_node(const P &p, const Dims &dm, MPT SW, int tau, EX ex, QG qg, QO q_out, TState<P> &s) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    real qi[P::MAXNX], di[P::MAXNU];
    SPK_UNROLL for (int k = 0; k < nx; ++k) qi[k] = s.xb[k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) di[k] = s.ub[k];
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        real qc[P::MAXNX];
        if (s.kids_leaf) { SPK_UNROLL for (int k = 0; k < nx; ++k) qc[k] = -s.xl[c][k]; }   // q_leaf = -xbar_leaf (:578-582)
        else if (s.kids_gleaf) { SPK_UNROLL for (int k = 0; k < nx; ++k) qc[k] = qg[(s.pos * d + c) * nx + k]; }
        else { SPK_UNROLL for (int k = 0; k < nx; ++k) qc[k] = ex[(tau * d + 1 + c) * nx + k]; }
        const MPT W = SW + dm.sw_oW + c * nx * nx;
        const MPT V = SW + dm.sw_oV + c * nu * nx;
        SPK_UNROLL for (int r = 0; r < nx; ++r) {
            real a = 0.;
            SPK_UNROLL for (int k = 0; k < nx; ++k) a += W[r + k * nx] * qc[k];
            qi[r] += a;
        }
        SPK_UNROLL for (int r = 0; r < nu; ++r) {
            real a = 0.;
            SPK_UNROLL for (int k = 0; k < nx; ++k) a += V[r + k * nu] * qc[k];
            di[r] -= a;
        }
    }
    SPK_UNROLL for (int k = 0; k < nu; ++k) s.dd[k] = di[k];
    if (s.lvl > 0) { SPK_UNROLL for (int k = 0; k < nx; ++k) ex[tau * nx + k] = qi[k]; }
    else if (q_out) { SPK_UNROLL for (int k = 0; k < nx; ++k) q_out[k] = qi[k]; }
}

// one level of the forward roll-out (implicit_l.jl:667-681).  xroot: x of the group root (x0 or the
// value the crown stored).  Nothing is stored to global memory here: a store in front of the level's
// barrier would make every level wait for its write acknowledgement; fwd_store() runs after the sweep.
template <class P, class MPT, class EX, class XR>
SPK_DEV void fwd_node(const P &p, const Dims &dm, MPT SW, MPT Bm, int tau, EX ex, XR xroot, TState<P> &s) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    real xi[P::MAXNX];
    if (s.lvl == 0) { SPK_UNROLL for (int k = 0; k < nx; ++k) xi[k] = xroot[k]; }
    else { SPK_UNROLL for (int k = 0; k < nx; ++k) xi[k] = ex[tau * nx + k]; }
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const MPT Bw = Bm + slot_event(c, d) * nx * nu;
        const MPT Ab = SW + dm.sw_oABK + c * nx * nx;
        SPK_UNROLL for (int r = 0; r < nx; ++r) {
            real a = 0., b2 = 0.;
            SPK_UNROLL for (int k = 0; k < nx; ++k) a += Ab[r + k * nx] * xi[k];
            SPK_UNROLL for (int k = 0; k < nu; ++k) b2 += Bw[r + k * nx] * s.dd[k];
            const real xc = a + b2;
            if (s.kids_gleaf) s.xl[c][r] = xc;       // group leaves: kept in registers, stored by fwd_store
            else ex[(tau * d + 1 + c) * nx + r] = xc;
        }
    }
    const MPT Kt = SW + dm.sw_oK;
    SPK_UNROLL for (int r = 0; r < nu; ++r) {
        real a = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += Kt[r + k * nu] * xi[k];
        s.ub[r] = a + s.dd[r];
    }
    SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = xi[k];
}

// after the forward sweep: x_i, u_i (and the x of children that are group leaves) -> zbar
template <class P, class ZO>
SPK_DEV void fwd_store(const P &p, const Dims &dm, ZO zo, const TState<P> &s) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + s.node * nx + k] = s.xb[k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) zo[dm.zu + s.node * nu + k] = s.ub[k];
    if (s.kids_gleaf) {
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            SPK_UNROLL for (int k = 0; k < nx; ++k) zo[dm.zx + (s.c0 + c) * nx + k] = s.xl[c][k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TEAM sweeps: T lanes per node, lane r owns row r of every mat-vec; all sweep vectors live in LDS:
//   XQ [group non-leaf][nx]  c_i -> q_i (backward, in place) -> x_i (forward)
//   ED [group non-leaf][nu]  e_i -> d_i (backward, in place)
//   UL [group non-leaf][nu]  u_i (forward)
//   XL [group leaves][nx]    q of the group leaves (backward input) -> x of the group leaves (forward output)
// A tree level with cnt nodes is cnt*T work items strided over the workgroup, so the per-level critical path
// is ~2 d nx FMAs per lane instead of ~2 d nx (nx + nu) for a single lane, and narrow levels no longer run
// with 63 of 64 lanes masked.  Matrices come from the LDS-staged sweep block or from the pool (MPT).
struct TeamLds { real *XQ, *ED, *UL, *XL; };

template <class P, class MPT>
SPK_DEV void bwd_level_team(const P &p, const Dims &dm, MPT SW, int lvl_first, int cnt, bool kids_gleaf, int T,
                            int tid, int nthreads, const TeamLds &L) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    for (int w = tid; w < cnt * T; w += nthreads) {
        const int pos = w / T, r = w - pos * T, li = lvl_first + pos;   // local index of the node
        if (r >= nx && r >= nu) continue;
        real q = (r < nx) ? L.XQ[li * nx + r] : 0., dd = (r < nu) ? L.ED[li * nu + r] : 0.;
        for (int c = 0; c < d; ++c) {
            const real *qc = kids_gleaf ? L.XL + (pos * d + c) * nx : L.XQ + (li * d + 1 + c) * nx;
            const MPT W = SW + dm.sw_oW + c * nx * nx;
            const MPT V = SW + dm.sw_oV + c * nu * nx;
            if (r < nx) { SPK_UNROLL4 for (int k = 0; k < nx; ++k) q += W[r + k * nx] * qc[k]; }
            if (r < nu) { SPK_UNROLL4 for (int k = 0; k < nx; ++k) dd -= V[r + k * nu] * qc[k]; }
        }
        if (r < nx) L.XQ[li * nx + r] = q;
        if (r < nu) L.ED[li * nu + r] = dd;
    }
}

template <class P, class MPT>
SPK_DEV void fwd_level_team(const P &p, const Dims &dm, MPT SW, MPT Bm, int lvl_first, int cnt, bool kids_gleaf, int T,
                            int tid, int nthreads, const TeamLds &L) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    for (int w = tid; w < cnt * T; w += nthreads) {
        const int pos = w / T, r = w - pos * T, li = lvl_first + pos;
        if (r >= nx && r >= nu) continue;
        const real *xi = L.XQ + li * nx, *di = L.ED + li * nu;
        if (r < nx) {
            for (int c = 0; c < d; ++c) {
                const MPT Bw = Bm + slot_event(c, d) * nx * nu;
                const MPT Ab = SW + dm.sw_oABK + c * nx * nx;
                real a = 0., b2 = 0.;
                SPK_UNROLL4 for (int k = 0; k < nx; ++k) a += Ab[r + k * nx] * xi[k];
                SPK_UNROLL4 for (int k = 0; k < nu; ++k) b2 += Bw[r + k * nx] * di[k];
                real *dst = kids_gleaf ? L.XL + (pos * d + c) * nx : L.XQ + (li * d + 1 + c) * nx;
                dst[r] = a + b2;
            }
        }
        if (r < nu) {
            const MPT Kt = SW + dm.sw_oK;
            real a = 0.;
            SPK_UNROLL4 for (int k = 0; k < nx; ++k) a += Kt[r + k * nu] * xi[k];
            L.UL[li * nu + r] = a + di[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA sweeps (gfx950 only; BASELINE config 5: nx = 64, nu = 16).  With the matrices shared by all nodes of a stage, one tree
// level of the sweep is a matrix product with the level's nodes as columns:
//   backward   [q_i ; d_i] (+)= [W_0 W_1 ; -V_0 -V_1] (80 x 128)  x  [q_child0 ; q_child1] (128 x cnt)
//   forward    x_child_c     =  [ABK_c  B_c]          (64 x 80)   x  [x_i ; d_i]           (80 x cnt),   u_i = K x_i + d_i
// evaluated with v_mfma_f64_16x16x4_f64 on 16 x 16 output tiles (16 matrix rows x 16 nodes), one wave per tile, the vectors
// in the same LDS arrays as the lane-team sweep (TeamLds), the matrix fragments straight from L2 (a stage's matrices are 155 KB).
// Fragment maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C / D: column j = lane & 15, rows (lane >> 4) + 4 * reg.  A tile of a narrow level (cnt < 16) runs with its surplus columns
// zeroed and not written back.  The f64 MFMA rate equals the vector FMA rate on MI355X (78.6 TFLOP/s); what it buys is operand
// traffic: one 8-byte matrix element and one 8-byte vector element per lane feed 2048 flops, against one of each per FMA.
// Rounding: the order of summation over k differs from the scalar sweeps (parity is checked at 1e-12 relative).
#ifndef SPOCK_HOST_EMU
typedef double spk_v4d __attribute__((ext_vector_type(4)));
typedef float spk_v4f __attribute__((ext_vector_type(4)));
// One 16 x 16 x 4 matrix-core step in the precision of `real`: v_mfma_f64_16x16x4_f64 in libspock_amd.so, v_mfma_f32_16x16x4_f32 in
// libspock_amd_f32.so (BASELINE config 5's fp32 leg then runs its sweep levels on the matrix cores too).  A[i = lane & 15][k = lane >> 4]
// and B[k = lane >> 4][j = lane & 15] are laid out alike; the accumulator's rows are not: register `reg` of a lane holds row
// (lane >> 4) + 4 reg (f64) or 4 (lane >> 4) + reg (f32) of column lane & 15 (cdna_hip_programming.md section 3).
typedef real spk_vacc __attribute__((ext_vector_type(4)));
SPK_DEV spk_v4d spk_mfma(double a, double b, spk_v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
SPK_DEV spk_v4f spk_mfma(float a, float b, spk_v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
SPK_DEV constexpr int spk_mfma_row(int kq, int reg) { return kRealIsDouble ? kq + 4 * reg : 4 * kq + reg; }

// instantiated shapes (nx, nu, d): BASELINE config 5 (64, 16, 2), config 4 (20, 20, 2) and config 3 (20, 20, 3) -- for nx = nu = 20
// the row tiles are padded to 16 and the rows past nx / nu masked
SPK_DEV bool mfma_sweep_applies(int nx, int nu, int d) {
    return (nx == 64 && nu == 16 && d == 2) || (nx == 20 && nu == 20 && (d == 2 || d == 3));
}
// (the dedicated sweep kernels of spock_skernels.hip add nx = nu = 50, d = 2 -- examples/server_heat/scaling.jl's shape: 13 k-steps
//  per child, the columns past 50 masked)

// One wave owns a ROW tile m of the level's product and keeps its matrix fragments (KF k-steps: one 8-byte element per lane
// each) in registers while it runs over the level's node tiles: the fragments are fetched from L2 once per level and wave --
// their latency (the sweep is a chain of small dependent levels) is paid once, not per node tile.
// (row_ok: this lane's matrix row exists -- the last row tile of a shape that is not a multiple of 16 is padded with zeros)
// nk: columns of the matrix (k-steps cover 4 KF >= nk of them: the columns past nk of a shape that is not a multiple of 4 are zero)
template <int KF, class MPT>
SPK_DEV void mfma_load_frags(real (&af)[KF], MPT Mx, long ld, int kq, int nk, int k_base, bool row_ok = true) {
#pragma unroll
    for (int q = 0; q < KF; ++q) af[q] = (row_ok && 4 * q + kq < nk) ? Mx[(long)(k_base + 4 * q + kq) * ld] : 0.;
}

// NXC / NUC / DC: compile-time nx / nu / d of the instantiation (64 / 16 / 2 for BASELINE config 5).  Work item = (row tile m,
// share sp of the node tiles): the wave keeps the fragments of BOTH children's matrices for its row tile and sums over the
// children in a fixed order, so the result does not depend on which wave ran when (bitwise repeatable).
// SXP / SUP: stride (in elements) of a node's x-sized / u-sized vector in the LDS arrays XQ / ED / UL; sxl: that of XL.  The dedicated
// sweep kernels (spock_skernels.hip) pad them by 4: with stride nx = 64 the 16 nodes of a tile sit 512 B apart and their operand
// reads hit the same LDS banks (measured: a 64-node level 13.6 -> 8.5 us, a 16-node one 4.8 -> 4.1 us).
template <int NXC, int NUC, int DC, int SXP = NXC, int SUP = NUC, class P, class MPT>
SPK_DEV void bwd_level_mfma(const P &p, const Dims &dm, MPT SW, int lvl_first, int cnt, bool kids_gleaf, int tid, int nthreads,
                            const TeamLds &L, int sxl = NXC) {
    constexpr int nx = NXC, nu = NUC, d = DC, KQ = (NXC + 3) / 4;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr int mq = (nx + 15) / 16, mt = mq + (nu + 15) / 16;
    const int ntile = (cnt + 15) / 16;
    int nsplit = nwaves / mt;
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit < 1) nsplit = 1;
    for (int it = wave; it < mt * nsplit; it += nwaves) {
        const int m = it % mt, sp = it / mt;
        const bool isq = m < mq;
        const int row0 = isq ? m * 16 : (m - mq) * 16, ld = isq ? nx : nu, nrow = isq ? nx : nu;
        const bool row_ok = row0 + j < nrow;
        real af[DC][KQ];
#pragma unroll
        for (int c = 0; c < d; ++c)
            mfma_load_frags<KQ>(af[c], (isq ? SW + dm.sw_oW + c * nx * nx : SW + dm.sw_oV + c * nu * nx) + (row_ok ? row0 + j : 0), ld, kq, nx, 0, row_ok);
        for (int nt = sp; nt < ntile; nt += nsplit) {
            const int pos = nt * 16 + j;
            const bool valid = pos < cnt;
            const int li = lvl_first + (valid ? pos : 0), posv = valid ? pos : 0;
            // one accumulator chain PER CHILD (independent matrix-core pipelines: the dependent chain is KQ long, not d * KQ),
            // added in child order afterwards: bitwise repeatable
            spk_vacc accc[DC];
#pragma unroll
            for (int c = 0; c < d; ++c) accc[c] = spk_vacc{0., 0., 0., 0.};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
#pragma unroll
                for (int c = 0; c < d; ++c) {
                    const real *qc = kids_gleaf ? L.XL + (long)(posv * d + c) * sxl : L.XQ + (long)(li * d + 1 + c) * SXP;
                    const real bv = (valid && (nx % 4 == 0 || 4 * q + kq < nx)) ? qc[4 * q + kq] : 0.;
                    accc[c] = spk_mfma(af[c][q], bv, accc[c]);
                }
            }
            spk_vacc acc = accc[0];
#pragma unroll
            for (int c = 1; c < d; ++c) acc += accc[c];
            if (valid) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = row0 + spk_mfma_row(kq, reg);
                    if (r >= nrow) continue;
                    if (isq) L.XQ[(long)li * SXP + r] += acc[reg];
                    else L.ED[(long)li * SUP + r] -= acc[reg];
                }
            }
        }
    }
}

template <int NXC, int NUC, int DC, int SXP = NXC, int SUP = NUC, class P, class MPT>
SPK_DEV void fwd_level_mfma(const P &p, const Dims &dm, MPT SW, MPT Bm, int lvl_first, int cnt, bool kids_gleaf, int tid, int nthreads,
                            const TeamLds &L, int sxl = NXC) {
    constexpr int nx = NXC, nu = NUC, d = DC, KQ = (NXC + 3) / 4, KU = (NUC + 3) / 4;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr int mq = (nx + 15) / 16, mu = (nu + 15) / 16;
    constexpr int mt = d * mq + mu;
    const int ntile = (cnt + 15) / 16;
    int nsplit = nwaves / mt;
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit < 1) nsplit = 1;
    for (int it = wave; it < mt * nsplit; it += nwaves) {
        const int m = it % mt, sp = it / mt;
        const bool isx = m < d * mq;
        const int c = isx ? m / mq : 0, row0 = isx ? (m - c * mq) * 16 : (m - d * mq) * 16, nrow = isx ? nx : nu;
        const bool row_ok = row0 + j < nrow;
        const int rj = row_ok ? row0 + j : 0;
        real af[KQ], bf[KU];
        if (isx) {
            mfma_load_frags<KQ>(af, SW + dm.sw_oABK + c * nx * nx + rj, nx, kq, nx, 0, row_ok);
            mfma_load_frags<KU>(bf, Bm + slot_event(c, d) * nx * nu + rj, nx, kq, nu, 0, row_ok);
        } else {
            mfma_load_frags<KQ>(af, SW + dm.sw_oK + rj, nu, kq, nx, 0, row_ok);
#pragma unroll
            for (int q = 0; q < KU; ++q) bf[q] = 0.;
        }
        for (int nt = sp; nt < ntile; nt += nsplit) {
            const int pos = nt * 16 + j;
            const bool valid = pos < cnt;
            const int li = lvl_first + (valid ? pos : 0), posv = valid ? pos : 0;
            const real *xi = L.XQ + (long)li * SXP, *di = L.ED + (long)li * SUP;
            // two independent accumulator chains over the even / odd k-steps of [ABK_c | K] x_i, a third for B_c d_i; added in a
            // fixed order afterwards
            spk_vacc acc = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.}, accb = {0., 0., 0., 0.};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const real bv = (valid && (nx % 4 == 0 || 4 * q + kq < nx)) ? xi[4 * q + kq] : 0.;
                if (q & 1) acc1 = spk_mfma(af[q], bv, acc1);
                else acc = spk_mfma(af[q], bv, acc);
            }
            acc += acc1;
            if (isx) {
#pragma unroll
                for (int q = 0; q < KU; ++q) {
                    const real bv = (valid && (nu % 4 == 0 || 4 * q + kq < nu)) ? di[4 * q + kq] : 0.;
                    accb = spk_mfma(bf[q], bv, accb);
                }
                acc += accb;
                if (valid) {
                    real *dst = kids_gleaf ? L.XL + (long)(posv * d + c) * sxl : L.XQ + (long)(li * d + 1 + c) * SXP;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) { const int r = row0 + spk_mfma_row(kq, reg); if (r < nx) dst[r] = acc[reg]; }
                }
            } else if (valid) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { const int r = row0 + spk_mfma_row(kq, reg); if (r < nu) L.UL[(long)li * SUP + r] = acc[reg] + di[r]; }
            }
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// Register-resident lane-team sweep (P::REGROWS).  Thread w <-> (node li = w / T, row r = w % T) for the WHOLE sweep;
// it loads its rows of W, V, ABK, B, K of its node's stage once, up front, so that a tree level costs one LDS read of
// the children's / parent's vector, ~2 d nx FMAs, one LDS write and the barrier -- no memory access on the dependent
// path.  Needs T * (group non-leaf nodes) <= workgroup size; vectors in LDS as for the other team sweep (TeamLds).
// which: 1 = rows of the backward sweep (W, V), 2 = rows of the forward sweep (ABK, B, K), 3 = both.  A kernel that runs the
// forward sweep first and the backward sweep last (SUB_BA) loads them in two steps so that the second set is not live
// across the node-local phases (with both sets live the fused kernel spills registers).
template <class P>
SPK_DEV void team_reg_init(const P &p, const Dims &dm, MatPtr M, int root_stage, int stages, int T, int tid, TState<P> &s, int which = 3) {
    const int nx = p.nx(), nu = p.nu(), d = p.d();
    const int li = tid / T, r = tid - li * T;
    int lvl = 0, first = 0, cnt = 1;
    while (lvl < stages - 1 && li >= first + cnt) { first += cnt; cnt *= d; ++lvl; }
    const bool ok = (lvl < stages - 1) && (r < nx || r < nu);
    s.tn_li = ok ? li : -1; s.tn_r = r; s.tn_lvl = lvl; s.tn_pos = li - first;
    if (!ok) return;
    const MatPtr SW = M + dm.mSweep + (root_stage + lvl) * dm.sw_stride;
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const MatPtr W = SW + dm.sw_oW + c * nx * nx, V = SW + dm.sw_oV + c * nu * nx, A = SW + dm.sw_oABK + c * nx * nx;
        const MatPtr B = M + dm.mB + slot_event(c, d) * nx * nu;
        SPK_UNROLL for (int k = 0; k < nx; ++k) {   // (this load order is the one the register allocator copes with best)
            if (which & 1) s.tW[c][k] = (r < nx) ? W[r + k * nx] : 0.;
            if (which & 2) s.tA[c][k] = (r < nx) ? A[r + k * nx] : 0.;
            if (which & 1) s.tV[c][k] = (r < nu) ? V[r + k * nu] : 0.;
        }
        if (which & 2) { SPK_UNROLL for (int k = 0; k < nu; ++k) s.tB[c][k] = (r < nx) ? B[r + k * nx] : 0.; }
    }
    if (which & 2) {
        const MatPtr K = SW + dm.sw_oK;
        SPK_UNROLL for (int k = 0; k < nx; ++k) s.tK[k] = (r < nu) ? K[r + k * nu] : 0.;
    }
}

template <class P>
SPK_DEV void bwd_team_reg(const P &p, int stages, const TeamLds &L, const TState<P> &s) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), li = s.tn_li, r = s.tn_r;
    const bool kg = (s.tn_lvl == stages - 2);
    real q = (r < nx) ? L.XQ[li * nx + r] : 0., dd = (r < nu) ? L.ED[li * nu + r] : 0.;
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const real *qc = kg ? L.XL + (s.tn_pos * d + c) * nx : L.XQ + (li * d + 1 + c) * nx;
        SPK_UNROLL for (int k = 0; k < nx; ++k) { const real v = qc[k]; q += s.tW[c][k] * v; dd -= s.tV[c][k] * v; }
    }
    if (r < nx) L.XQ[li * nx + r] = q;
    if (r < nu) L.ED[li * nu + r] = dd;
}

template <class P>
SPK_DEV void fwd_team_reg(const P &p, int stages, const TeamLds &L, const TState<P> &s) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), li = s.tn_li, r = s.tn_r;
    const bool kg = (s.tn_lvl == stages - 2);
    const real *xi = L.XQ + li * nx, *di = L.ED + li * nu;
    real xv[P::MAXNX], dv[P::MAXNU];
    SPK_UNROLL for (int k = 0; k < nx; ++k) xv[k] = xi[k];
    SPK_UNROLL for (int k = 0; k < nu; ++k) dv[k] = di[k];
    if (r < nx) {
        SPK_UNROLL for (int c = 0; c < d; ++c) {
            real a = 0., b2 = 0.;
            SPK_UNROLL for (int k = 0; k < nx; ++k) a += s.tA[c][k] * xv[k];
            SPK_UNROLL for (int k = 0; k < nu; ++k) b2 += s.tB[c][k] * dv[k];
            real *dst = kg ? L.XL + (s.tn_pos * d + c) * nx : L.XQ + (li * d + 1 + c) * nx;
            dst[r] = a + b2;
        }
    }
    if (r < nu) {
        real a = 0.;
        SPK_UNROLL for (int k = 0; k < nx; ++k) a += s.tK[k] * xv[k];
        L.UL[li * nu + r] = a + dv[r];
    }
}

// ------------------------------------------------------------------------------------------------
// phase B, node-local.  zb = zbar (x_i, u_i and the leaf x_j come from the thread state unless
// F_NO_SWEEP), z/v = the input iterate, vo = vbar.  mx[0..2] accumulate |xi_1|, |xi_2|, |xi| maxima.
struct ProxScal {
    real sigma, inv_sigma;                     // 1/sigma: arg/sigma of prox_h_conj! as a multiplication
    real gamma;
    real inv_sigma_neg, inv_gamma_neg;         // -1/sigma, -1/gamma as the reference forms them (cp.jl:73-75)
    bool want_xi;                                // LOG-mode CP: xi = xi_1 + L' xi_2 (cp.jl:83-86)
};

// vbar.v2 of node j alone, the same arithmetic as b_local's risk block.  The fused B+A kernel's crown workgroup needs
// it for the subtree roots, whose own threads sit in other workgroups of the same launch.
template <class P, class ZI, class VI, class ZB>
SPK_DEV real risk_v2_only(const P &p, const Dims &dm, MatPtr M, ZI z, VI v, ZB zb, int j, const ProxScal &ps) {
    const int ny = p.ny();
    const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? j * ny : 0);
    real bw = 0.;
    SPK_UNROLL for (int k = 0; k < ny; ++k) {
        const real zbk = zb[dm.zy + j * ny + k], zk = z[dm.zy + j * ny + k];
        const real wy = 2. * zbk + -1. * zk;
        bw += bb[k] * wy;
    }
    const real zbs = zb[dm.zs + j], zs_ = z[dm.zs + j];
    const real ws = 2. * zbs + -1. * zs_;
    real t2 = ps.sigma * (ws - bw) + v[dm.o2 + j];
    t2 = t2 * ps.inv_sigma;
    const real p2 = t2 >= 0. ? t2 : 0.;
    return ps.sigma * (t2 - p2);
}

// Phase A of the NEXT iteration gathered while phase B of this one has its results in registers (element-major kernels, nodes
// whose children are leaves): z <- zbar, v <- vbar at lambda = 1 (cp.jl:34-52), so (-gamma) L'vbar + zbar on the node's own
// entries needs nothing from memory.  ax / au accumulate (L'vbar) on x_i / u_i, vec = [y_i; s_ch; tau_ch], xl the leaves' xbar.
template <class P>
struct ANext {
    real ax[P::MAXNX], au[P::MAXNU], vec[P::MAXM], xl[P::MAXD][P::MAXNX];
};

// SPM (SP driver, spock_sp.hpp): the inner products of run_sp! that involve the residual r = (z, v) - T(z, v) of THIS
// evaluation, accumulated where its pieces are at hand instead of in a separate pass that applies L and L' again.
// With dz = zbar - z, dv = vbar - v and the termination residuals xi_1 = L'dv - dz/gamma, xi_2 = L dz - dv/sigma:
//   <r, P r> = sum_z dz (dz - gamma L'dv) + sum_v dv (dv - sigma L dz) = -gamma sum_z dz xi_1 - sigma sum_v dv xi_2
//   <r, P d> = -sum_z d_z (dz - sigma L'dv) - sum_v d_v (dv - gamma L dz)        (implicit_l.jl:485-516)
//   SPM 1: sp[0] += sum_z dz xi_1, sp[1] += sum_v dv xi_2;   SPM 2: also sp[2] += <r, P d> with d = (dirz, dirv)
// PART: 0 everything; 1 without the leaf-children blocks; 2 only the leaf-children blocks.  The leaf blocks share nothing
// with the rest but the residual maxima and the SP sums (both reduced over the workgroup anyway), so in the single-instance
// regime, where the workgroup has idle waves during phase B, a second thread does them at the same time (spock_group.hpp).
template <class P, int SPM, int PART, class ZI, class VI, class ZB, class VO>
SPK_DEV void b_local_g(const P &p, const Dims &dm, MatPtr M, ZI z, VI v, ZB zb, VO vo, TState<P> &s,
                       const ProxScal &ps, unsigned flags, real *mx, real *sp = nullptr, const real *dirz = nullptr,
                       const real *dirv = nullptr, ANext<P> *an = nullptr) {
    const int nx = p.nx(), nu = p.nu(), d = p.d(), ny = p.ny();
    const int i = s.node, nxu = nx + nu;
    const real mgn = -ps.gamma;   // (an: next phase A)
    const real sg = ps.sigma, isg = ps.inv_sigma, nis = ps.inv_sigma_neg, nig = ps.inv_gamma_neg;
    const bool wz = (flags & F_W_IS_ZBAR) != 0, noprox = (flags & F_NO_PROX) != 0, wx = ps.want_xi;
    real m1 = mx[0], m2 = mx[1], m3 = mx[2];
    real poison = IsIlvView<ZI>::value ? mx[3] : 0.;   // element-major instantiation: mx has a fourth slot (spk_resmax)
    constexpr bool MAIN = (PART != 2), LEAF = (PART != 1);
    real sp_z = 0., sp_v = 0., sp_d = 0.;
    const real gm = ps.gamma;
    // one z entry (index iz, dz, xi_1) / one v entry (index iv, dv, xi_2) of the sums above
    auto acc_z = [&](int iz, real dz, real x1) {
        if (SPM >= 1) sp_z += dz * x1;
        if (SPM == 2) sp_d -= dirz[iz] * (dz - sg * (x1 - nig * dz));
    };
    auto acc_v = [&](int iv, real dv, real x2) {
        if (SPM >= 1) sp_v += dv * x2;
        if (SPM == 2) sp_d -= dirv[iv] * (dv - gm * (x2 - nis * dv));
    };
    if (flags & F_NO_SWEEP) {
        SPK_UNROLL for (int k = 0; k < nx; ++k) s.xb[k] = zb[dm.zx + i * nx + k];
        SPK_UNROLL for (int k = 0; k < nu; ++k) s.ub[k] = zb[dm.zu + i * nu + k];
        if (s.kids_leaf) {
            SPK_UNROLL for (int c = 0; c < d; ++c) {
                SPK_UNROLL for (int k = 0; k < nx; ++k) s.xl[c][k] = zb[dm.zx + (s.c0 + c) * nx + k];
            }
        }
    }
    // w = 2 zbar - z (cp.jl:23-24) and dz = zbar - z on x_i, u_i
    real wxv[P::MAXNX], dzx[P::MAXNX], wuv[P::MAXNU], dzu[P::MAXNU];
    if (MAIN) {
        SPK_UNROLL for (int k = 0; k < nx; ++k) {
            const real zi = z[dm.zx + i * nx + k];
            wxv[k] = wz ? s.xb[k] : 2. * s.xb[k] + -1. * zi;
            dzx[k] = s.xb[k] - zi;
        }
        SPK_UNROLL for (int k = 0; k < nu; ++k) {
            const real zi = z[dm.zu + i * nu + k];
            wuv[k] = wz ? s.ub[k] : 2. * s.ub[k] + -1. * zi;
            dzu[k] = s.ub[k] - zi;
        }
    }
    // accumulators of (L' dv) and (L' xi_2) on x_i, u_i
    real ltx[P::MAXNX], ltu[P::MAXNU], lxx[P::MAXNX], lxu[P::MAXNU];

    // ---- v1, v2 : risk-measure block of node i -------------------------------------------------
    if (MAIN) {
        const MatPtr bb = M + dm.mb + (dm.nRm > 1 ? i * ny : 0);      // rms[i].b (implicit_l.jl:219)
        real wy[P::MAXNY], dzy[P::MAXNY], t1[P::MAXNY], pr[P::MAXNY];
        real bw = 0., bdz = 0.;
        SPK_UNROLL for (int k = 0; k < ny; ++k) {
            const real zbk = zb[dm.zy + i * ny + k], zk = z[dm.zy + i * ny + k];
            wy[k] = wz ? zbk : 2. * zbk + -1. * zk;
            dzy[k] = zbk - zk;
            bw += bb[k] * wy[k];
            bdz += bb[k] * dzy[k];
        }
        const real zbs = zb[dm.zs + i], zs_ = z[dm.zs + i];
        const real ws = wz ? zbs : 2. * zbs + -1. * zs_, dzs = zbs - zs_;
        // v1 = y, v2 = s - b'y  (implicit_l.jl:208-222)
        SPK_UNROLL for (int k = 0; k < ny; ++k) t1[k] = sg * wy[k] + v[i * ny + k];
        real t2 = sg * (ws - bw) + v[dm.o2 + i];
        real nv2;
        if (noprox) {
            SPK_UNROLL for (int k = 0; k < ny; ++k) pr[k] = t1[k];
            nv2 = t2;
        } else {
            SPK_UNROLL for (int k = 0; k < ny; ++k) { t1[k] = t1[k] * isg; pr[k] = t1[k]; }
            // projection onto the dual cone, sub-cone by sub-cone (implicit_l.jl:833-845).  Without second-order
            // sub-cones it is element-wise: the per-element kind mask keeps pr[] in registers (no dynamic indexing).
            if (!dm.risk_has_soc) {
                SPK_UNROLL for (int k = 0; k < ny; ++k) {
                    const int kind = (dm.elem_kind_mask >> (2 * k)) & 3;
                    const real v0 = pr[k];
                    pr[k] = kind == 0 ? 0. : (kind == 1 ? (v0 < 0. ? 0. : v0) : (kind == 2 ? (v0 > 0. ? 0. : v0) : v0));
                }
            } else {
                // general path (second-order sub-cones present): predicated, statically indexed loops so that pr[]
                // stays in registers (a dynamically indexed local array would put the whole kernel on scratch memory)
                int off = 0;
                for (int cn = 0; cn < dm.n_cones; ++cn) {
                    const int kind = dm.cone_kind[cn], dim = dm.cone_dim[cn];
                    real t = 0., ss = 0.;
                    SPK_UNROLL for (int k = 0; k < ny; ++k) {
                        if (k == off) t = pr[k];
                        if (k > off && k < off + dim) ss += pr[k] * pr[k];
                    }
                    const real x_norm = sqrt(ss);
                    // SOC (t; x): keep | zero | scale  (risk_measures.jl:47-71)
                    const bool keep = (x_norm <= t), zero = !keep && (x_norm <= -t);
                    const real t2 = (t + x_norm) / 2., sc = t2 / x_norm;
                    SPK_UNROLL for (int k = 0; k < ny; ++k) {
                        if (k < off || k >= off + dim) continue;
                        const real v0 = pr[k];
                        real r;
                        if (kind == 0) r = 0.;
                        else if (kind == 1) r = v0 < 0. ? 0. : v0;
                        else if (kind == 2) r = v0 > 0. ? 0. : v0;
                        else if (kind == 4) r = keep ? v0 : (zero ? 0. : (k == off ? t2 : sc * v0));
                        else r = v0;
                        pr[k] = r;
                    }
                    off += dim;
                }
            }
            SPK_UNROLL for (int k = 0; k < ny; ++k) pr[k] = sg * (t1[k] - pr[k]);
            t2 = t2 * isg;
            const real p2 = t2 >= 0. ? t2 : 0.;   // implicit_l.jl:849-851
            nv2 = sg * (t2 - p2);
        }
        const real dv2 = nv2 - v[dm.o2 + i];
        const real x2 = (dzs - bdz) + nis * dv2;          // xi_2 on v2
        spk_resmax<ZI>(m2, x2, poison);
        acc_v(dm.o2 + i, dv2, x2);
        vo[dm.o2 + i] = nv2;
        if (an) { SPK_UNROLL for (int k = 0; k < ny; ++k) an->vec[k] = mgn * (pr[k] - bb[k] * nv2) + (wy[k] - dzy[k]); }   // zbar.y = w - dz
        SPK_UNROLL for (int k = 0; k < ny; ++k) {
            const real dv1 = pr[k] - v[i * ny + k];
            vo[i * ny + k] = pr[k];
            const real x2k = dzy[k] + nis * dv1;          // xi_2 on v1
            spk_resmax<ZI>(m2, x2k, poison);
            const real x1k = (dv1 - bb[k] * dv2) + nig * dzy[k];   // xi_1 on y (implicit_l.jl:423-429)
            spk_resmax<ZI>(m1, x1k, poison);
            acc_v(i * ny + k, dv1, x2k); acc_z(dm.zy + i * ny + k, dzy[k], x1k);
            if (wx) spk_resmax<ZI>(m3, (x2k - bb[k] * x2) + x1k, poison);
        }
        const real x1s = dv2 + nig * dzs;                 // xi_1 on s_i (implicit_l.jl:440-442)
        spk_resmax<ZI>(m1, x1s, poison);
        acc_z(dm.zs + i, dzs, x1s);
        if (wx) spk_resmax<ZI>(m3, x2 + x1s, poison);
    }
    spk_block_fence<ZI>();
    // ---- v7 : box on (x_i, u_i)  (implicit_l.jl:275-282, constraints.jl:38-54) ---------------------
    if (MAIN) {
        SPK_UNROLL for (int k = 0; k < nxu; ++k) {
            const bool isx = k < nx;
            const real wv = isx ? wxv[k] : wuv[k - nx];
            const real dzv = isx ? dzx[k] : dzu[k - nx];
            const real vin = v[dm.o7 + i * nxu + k];
            real t = sg * wv + vin, nvv;
            if (noprox) nvv = t;
            else {
                t = t * isg;
                const real pj = isx ? spk_clamp(t, dm.x_min, dm.x_max) : spk_clamp(t, dm.u_min, dm.u_max);
                nvv = sg * (t - pj);
            }
            vo[dm.o7 + i * nxu + k] = nvv;
            if (an) { if (isx) an->ax[k] = nvv; else an->au[k - nx] = nvv; }
            const real dv = nvv - vin;
            const real x2 = dzv + nis * dv;
            spk_resmax<ZI>(m2, x2, poison);
            acc_v(dm.o7 + i * nxu + k, dv, x2);
            if (isx) { ltx[k] = dv; lxx[k] = x2; } else { ltu[k - nx] = dv; lxu[k - nx] = x2; }
        }
    }
    // ---- per child j: (v6_j; v3_j, v4_j, v5_j) in SOC(nx+nu+2)  (implicit_l.jl:226-272, :854-900) ---
    SPK_UNROLL for (int c = 0; c < d; ++c) {
        const int j = s.c0 + c, jn = j - 1;
        spk_block_fence<ZI>();
        if (MAIN) {
        const MatPtr SQ = M + dm.mSqQ + (P::SHARED ? 0 : jn) * nx * nx;
        const MatPtr SR = M + dm.mSqR + (P::SHARED ? 0 : jn) * nu * nu;
        real tt[P::MAXNX + P::MAXNU + 2], pj[P::MAXNX + P::MAXNU + 2], lz[P::MAXNX + P::MAXNU + 2];
        // tt/pj ordering = the reference's workspace: [v6, v3.., v4.., v5]; lz = L dz on the same entries
        const real zbt = zb[dm.zt + jn], zt_ = z[dm.zt + jn];
        const real wt = wz ? zbt : 2. * zbt + -1. * zt_, dzt = zbt - zt_;
        sym_mv<P, false>(nx, SQ, wxv, tt + 1);
        sym_mv<P, false>(nx, SQ, dzx, lz + 1);
        sym_mv<P, false>(nu, SR, wuv, tt + 1 + nx);
        sym_mv<P, false>(nu, SR, dzu, lz + 1 + nx);
        tt[0] = sg * (0.5 * wt) + v[dm.o6 + jn];
        lz[0] = 0.5 * dzt;
        SPK_UNROLL for (int r = 0; r < nx; ++r) tt[1 + r] = sg * tt[1 + r] + v[dm.o3 + jn * nx + r];
        SPK_UNROLL for (int r = 0; r < nu; ++r) tt[1 + nx + r] = sg * tt[1 + nx + r] + v[dm.o4 + jn * nu + r];
        tt[1 + nxu] = sg * (0.5 * wt) + v[dm.o5 + jn];
        lz[1 + nxu] = 0.5 * dzt;
        const int len = nxu + 2;
        if (noprox) {
            SPK_UNROLL for (int k = 0; k < len; ++k) pj[k] = tt[k];
        } else {
            SPK_UNROLL for (int k = 0; k < len; ++k) tt[k] = tt[k] * isg;
            tt[1 + nxu] -= 0.5;   // v5 -= 1/2, v6 += 1/2  (implicit_l.jl:928-933)
            tt[0] += 0.5;
            SPK_UNROLL for (int k = 0; k < len; ++k) pj[k] = tt[k];
            project_soc<P::MAXNX + P::MAXNU + 2>(pj, len);
            SPK_UNROLL for (int k = 0; k < len; ++k) pj[k] = sg * (tt[k] - pj[k]);
        }
        // store, residuals; tt becomes dv, lz becomes xi_2
        vo[dm.o6 + jn] = pj[0];
        tt[0] = pj[0] - v[dm.o6 + jn];
        SPK_UNROLL for (int r = 0; r < nx; ++r) { vo[dm.o3 + jn * nx + r] = pj[1 + r]; tt[1 + r] = pj[1 + r] - v[dm.o3 + jn * nx + r]; }
        SPK_UNROLL for (int r = 0; r < nu; ++r) { vo[dm.o4 + jn * nu + r] = pj[1 + nx + r]; tt[1 + nx + r] = pj[1 + nx + r] - v[dm.o4 + jn * nu + r]; }
        vo[dm.o5 + jn] = pj[1 + nxu];
        if (an) {   // L'vbar on x_i, u_i and on tau_j (implicit_l.jl:365-373, :412-420, :432-437)
            sym_mv<P, true>(nx, SQ, pj + 1, an->ax);
            sym_mv<P, true>(nu, SR, pj + 1 + nx, an->au);
            an->vec[ny + d + c] = mgn * (0.5 * pj[1 + nxu] + 0.5 * pj[0]) + zbt;
        }
        tt[1 + nxu] = pj[1 + nxu] - v[dm.o5 + jn];
        SPK_UNROLL for (int k = 0; k < len; ++k) { lz[k] = lz[k] + nis * tt[k]; spk_resmax<ZI>(m2, lz[k], poison); }
        if (SPM >= 1) {
            acc_v(dm.o6 + jn, tt[0], lz[0]);
            SPK_UNROLL for (int r = 0; r < nx; ++r) acc_v(dm.o3 + jn * nx + r, tt[1 + r], lz[1 + r]);
            SPK_UNROLL for (int r = 0; r < nu; ++r) acc_v(dm.o4 + jn * nu + r, tt[1 + nx + r], lz[1 + nx + r]);
            acc_v(dm.o5 + jn, tt[1 + nxu], lz[1 + nxu]);
        }
        // L' contributions: x_i += sqrtQ dv3, u_i += sqrtR dv4 (implicit_l.jl:365-373, :412-420)
        sym_mv<P, true>(nx, SQ, tt + 1, ltx);
        sym_mv<P, true>(nu, SR, tt + 1 + nx, ltu);
        if (wx) { sym_mv<P, true>(nx, SQ, lz + 1, lxx); sym_mv<P, true>(nu, SR, lz + 1 + nx, lxu); }
        // xi_1 on tau_j: 0.5 dv5 + 0.5 dv6 - dtau/gamma  (implicit_l.jl:432-437)
        const real x1t = (0.5 * tt[1 + nxu] + 0.5 * tt[0]) + nig * dzt;
        spk_resmax<ZI>(m1, x1t, poison);
        acc_z(dm.zt + jn, dzt, x1t);
        if (wx) spk_resmax<ZI>(m3, (0.5 * lz[1 + nxu] + 0.5 * lz[0]) + x1t, poison);
        }

        spk_block_fence<ZI>();
        // ---- leaf child: (v13; v11, v12) in SOC(nx+2), v14 in the x box (implicit_l.jl:286-317, :752-810)
        if (LEAF && s.kids_leaf) {
            const int li = j - dm.nn;
            const MatPtr SN = M + dm.mSqQN + (P::SHARED ? 0 : li) * nx * nx;
            real wl[P::MAXNX], dl[P::MAXNX];
            SPK_UNROLL for (int k = 0; k < nx; ++k) {
                const real zk = z[dm.zx + j * nx + k];
                wl[k] = wz ? s.xl[c][k] : 2. * s.xl[c][k] + -1. * zk;
                dl[k] = s.xl[c][k] - zk;
            }
            const real zbs = zb[dm.zs + j], zs_ = z[dm.zs + j];
            const real ws = wz ? zbs : 2. * zbs + -1. * zs_, dzs = zbs - zs_;
            real t3[P::MAXNX + 2], p3[P::MAXNX + 2], l3[P::MAXNX + 2];   // [v13, v11.., v12]
            sym_mv<P, false>(nx, SN, wl, t3 + 1);
            sym_mv<P, false>(nx, SN, dl, l3 + 1);
            t3[0] = sg * (0.5 * ws) + v[dm.o13 + li];
            l3[0] = 0.5 * dzs;
            SPK_UNROLL for (int r = 0; r < nx; ++r) t3[1 + r] = sg * t3[1 + r] + v[dm.o11 + li * nx + r];
            t3[1 + nx] = sg * (0.5 * ws) + v[dm.o12 + li];
            l3[1 + nx] = 0.5 * dzs;
            const int l2 = nx + 2;
            if (noprox) {
                SPK_UNROLL for (int k = 0; k < l2; ++k) p3[k] = t3[k];
            } else {
                SPK_UNROLL for (int k = 0; k < l2; ++k) t3[k] = t3[k] * isg;
                t3[1 + nx] -= 0.5;   // v12 -= 1/2, v13 += 1/2 (implicit_l.jl:934-939)
                t3[0] += 0.5;
                SPK_UNROLL for (int k = 0; k < l2; ++k) p3[k] = t3[k];
                project_soc<P::MAXNX + 2>(p3, l2);
                SPK_UNROLL for (int k = 0; k < l2; ++k) p3[k] = sg * (t3[k] - p3[k]);
            }
            vo[dm.o13 + li] = p3[0];
            t3[0] = p3[0] - v[dm.o13 + li];
            SPK_UNROLL for (int r = 0; r < nx; ++r) { vo[dm.o11 + li * nx + r] = p3[1 + r]; t3[1 + r] = p3[1 + r] - v[dm.o11 + li * nx + r]; }
            vo[dm.o12 + li] = p3[1 + nx];
            real q11[P::MAXNX];
            if (an) {   // s_leaf and x_leaf entries of (-gamma) L'vbar + zbar (implicit_l.jl:376-396, :445-448)
                an->vec[ny + c] = mgn * (0.5 * p3[1 + nx] + 0.5 * p3[0]) + zbs;
                sym_mv<P, false>(nx, SN, p3 + 1, q11);
            }
            t3[1 + nx] = p3[1 + nx] - v[dm.o12 + li];
            SPK_UNROLL for (int k = 0; k < l2; ++k) { l3[k] = l3[k] + nis * t3[k]; spk_resmax<ZI>(m2, l3[k], poison); }
            if (SPM >= 1) {
                acc_v(dm.o13 + li, t3[0], l3[0]);
                SPK_UNROLL for (int r = 0; r < nx; ++r) acc_v(dm.o11 + li * nx + r, t3[1 + r], l3[1 + r]);
                acc_v(dm.o12 + li, t3[1 + nx], l3[1 + nx]);
            }
            // v14 and the x_j, s_j residuals: x_leaf = v14 + sqrtQN v11 (implicit_l.jl:376-396)
            real a11[P::MAXNX], b11[P::MAXNX];
            sym_mv<P, false>(nx, SN, t3 + 1, a11);
            if (wx) sym_mv<P, false>(nx, SN, l3 + 1, b11);
            SPK_UNROLL for (int r = 0; r < nx; ++r) {
                const real vin = v[dm.o14 + li * nx + r];
                real t = sg * wl[r] + vin, nvv;
                if (noprox) nvv = t;
                else { t = t * isg; nvv = sg * (t - spk_clamp(t, dm.x_min, dm.x_max)); }
                vo[dm.o14 + li * nx + r] = nvv;
                if (an) an->xl[c][r] = mgn * (nvv + q11[r]) + s.xl[c][r];
                const real dv = nvv - vin;
                const real x2 = dl[r] + nis * dv;
                spk_resmax<ZI>(m2, x2, poison);
                const real x1 = (dv + a11[r]) + nig * dl[r];
                spk_resmax<ZI>(m1, x1, poison);
                acc_v(dm.o14 + li * nx + r, dv, x2); acc_z(dm.zx + j * nx + r, dl[r], x1);
                if (wx) spk_resmax<ZI>(m3, (x2 + b11[r]) + x1, poison);
            }
            const real x1s = (0.5 * t3[1 + nx] + 0.5 * t3[0]) + nig * dzs;   // s_leaf (implicit_l.jl:445-448)
            spk_resmax<ZI>(m1, x1s, poison);
            acc_z(dm.zs + j, dzs, x1s);
            if (wx) spk_resmax<ZI>(m3, (0.5 * l3[1 + nx] + 0.5 * l3[0]) + x1s, poison);
        }
    }
    spk_block_fence<ZI>();
    // xi_1 (and xi) on x_i, u_i
    SPK_UNROLL for (int k = 0; k < (MAIN ? nx : 0); ++k) {
        const real x1 = ltx[k] + nig * dzx[k];
        spk_resmax<ZI>(m1, x1, poison);
        acc_z(dm.zx + i * nx + k, dzx[k], x1);
        if (wx) spk_resmax<ZI>(m3, lxx[k] + x1, poison);
    }
    SPK_UNROLL for (int k = 0; k < (MAIN ? nu : 0); ++k) {
        const real x1 = ltu[k] + nig * dzu[k];
        spk_resmax<ZI>(m1, x1, poison);
        acc_z(dm.zu + i * nu + k, dzu[k], x1);
        if (wx) spk_resmax<ZI>(m3, lxu[k] + x1, poison);
    }
    if (IsIlvView<ZI>::value) mx[3] = poison;
    mx[0] = m1; mx[1] = m2; mx[2] = m3;
    if (SPM >= 1) { sp[0] += sp_z; sp[1] += sp_v; }
    if (SPM == 2) sp[2] += sp_d;
}

template <class P, int SPM = 0, int PART = 0>
SPK_DEV void b_local(const P &p, const Dims &dm, MatPtr M, const real *__restrict__ z,
                     const real *__restrict__ v, const real *zb, real *__restrict__ vo, TState<P> &s,
                     const ProxScal &ps, unsigned flags, real *mx, real *sp = nullptr, const real *dirz = nullptr,
                     const real *dirv = nullptr) {
    b_local_g<P, SPM, PART>(p, dm, M, z, v, zb, vo, s, ps, flags, mx, sp, dirz, dirv);
}

// ------------------------------------------------------------------------------------------------
// closed-loop step (examples/server_heat/mpc_simulation.jl:164-177): u = z[u_inds[1:nu]], x0 <- A_w x0 + B_w u.
// One thread per instance-row; `row` in [0, nx).
SPK_DEV real mpc_advance_row(const Dims &dm, MatPtr M, const real *z, long stride_e, const real *x0, int w, int row) {
    const MatPtr A = M + dm.mA + w * dm.nx * dm.nx, B = M + dm.mB + w * dm.nx * dm.nu;
    real a = 0., b = 0.;
    for (int k = 0; k < dm.nx; ++k) a += A[row + k * dm.nx] * x0[k];
    for (int k = 0; k < dm.nu; ++k) b += B[row + k * dm.nx] * z[(long)(dm.zu + k) * stride_e];
    return a + b;
}

// ------------------------------------------------------------------------------------------------
// Termination bookkeeping of run_cp! / should_terminate! (cp.jl:54-123, :205-228), run by ONE thread per
// instance at the start of an iteration (for the previous one) and by the finalize kernel.
// Returns true when the instance is finished.
// Snapshot of the control block taken with ONE batch of loads (together with the residual partials), so that the
// termination logic costs one memory round trip instead of a chain of dependent ones.
struct CtlSnap { int done, pending, log_mode, alg, max_iter, evals, iter; double tol, r0, r1; };
SPK_DEV CtlSnap ctl_snapshot(const Ctl &c) {
    CtlSnap s;
    s.done = c.done; s.pending = c.pending; s.log_mode = c.log_mode; s.alg = c.alg; s.max_iter = c.max_iter;
    s.evals = c.evals; s.iter = c.iter; s.tol = c.tol; s.r0 = c.res0[0]; s.r1 = c.res0[1];
    return s;
}

SPK_DEV bool ctl_prologue(Ctl &c, const CtlSnap &s, double x1, double x2, double x3, double *hist, int hist_cap, int parity,
                          bool start_next) {
    if (s.done) return true;
    if (s.pending) {
        const double tol = s.tol;
        int evals = s.evals, iter = s.iter;
        const bool log = (s.log_mode == 2);
        double xi = NAN;
        if (log) xi = (s.alg == 1) ? x3 : (x1 > x2 ? x1 : x2);   // cp.jl:83-86 / sp.jl:308
        c.xi1 = x1; c.xi2 = x2; c.xi = xi;
        if (hist && evals < hist_cap) {
            double *h = hist + (long)evals * kHistStride;
            h[0] = xi; h[1] = x1; h[2] = x2; h[3] = 0.;
        }
        evals += 1;
        c.evals = evals;
        bool term;
        if (log) term = (xi <= tol);
        else {
            const double t1 = tol * s.r0 > tol ? tol * s.r0 : tol;   // max(tol*res_0, tol)
            const double t2 = tol * s.r1 > tol ? tol * s.r1 : tol;
            term = (x1 <= t1 && x2 <= t2);
        }
        if (s.r0 == -INFINITY) c.res0[0] = x1;
        if (s.r1 == -INFINITY) c.res0[1] = x2;
        if (x1 != x1 || x2 != x2) { c.pending = 0; c.done = 1; c.status = 2; c.result_parity = parity; return true; }
        if (term) { c.pending = 0; c.done = 1; c.status = 0; c.result_parity = parity; return true; }
        iter += 1;
        c.iter = iter;
        if (iter >= s.max_iter) { c.pending = 0; c.done = 1; c.status = 1; c.result_parity = parity; return true; }
    }
    c.pending = start_next ? 1 : 0;
    return false;
}

}  // namespace spock
