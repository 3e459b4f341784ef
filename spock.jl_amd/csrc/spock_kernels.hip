// spock_kernels.hip -- gfx950 kernels of the fused Chambolle-Pock iteration and their launchers.
// The per-thread bodies live in spock_device.hpp, the per-workgroup orchestration in spock_group.hpp;
// this file supplies the GPU execution policy (state in registers, LDS exchange buffer, wave/LDS max
// reduction), instantiates the kernels for the compile-time shapes of the benchmark configurations and
// for run-time dimensions, and launches them on the handle's stream.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_group.hpp"
#include "spock_gpu_exec.hpp"
#include "spock_shapes.hpp"
#include "spock_sp.hpp"
#include "spock_team_local.hpp"

namespace spock {

#ifdef SPOCK_WAVES_PER_EU   // experiment builds only
#define SPK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(SPOCK_WAVES_PER_EU)))
#else
#define SPK_KERNEL_ATTR
#endif

template <class P, int MODE>
__global__ void __launch_bounds__(P::TEAMLOCAL ? 1024 : 512) SPK_KERNEL_ATTR spock_group_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // carved in units of 8 bytes (launch_geometry) whatever `real` is
    GpuExec<P> x;
    x.ex = reinterpret_cast<real *>(smem);
    x.red = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles);
    x.fl = reinterpret_cast<int *>(x.red + 3);
    x.stamps = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles + 4);
    x.sums = reinterpret_cast<double *>(smem + a.ex_doubles + 4 + 128);
    x.lm = reinterpret_cast<real *>(smem + a.ex_doubles + 4 + 128 + 64);
    const bool wave_stamps = (a.flags & F_DBG_STAMPS) && a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && MODE != MODE_FINALIZE;
    unsigned long long *wd = a.dbg ? a.dbg + (long)(MODE == MODE_SUB_A ? 0 : (MODE == MODE_CROWN ? 1 : 2)) * 128 : nullptr;
    if (wave_stamps) wd[2 * (40 + (threadIdx.x >> 6)) + 1] = __builtin_amdgcn_s_memrealtime();
    group_body<P, MODE>(x, a, (int)blockIdx.y, (int)blockIdx.x);
    if (wave_stamps) wd[2 * (50 + (threadIdx.x >> 6)) + 1] = __builtin_amdgcn_s_memrealtime();
}

static inline int round64(int t) { return t < 64 ? 64 : ((t + 63) / 64) * 64; }

template <class P>
static int launch_variant(int mode, const KArgs &a_in, hipStream_t st) {
    const LaunchGeo g = launch_geometry(mode, a_in.dm, a_in.batch);
    if (g.threads > (P::TEAMLOCAL ? 1024 : 512)) return (int)hipErrorInvalidConfiguration;
    KArgs a = a_in;
    // LDS-staged sweep matrices pay off when many workgroups hide the staging copy (batched regime: +18 % at 512
    // instances); with a handful of workgroups the copy is pure latency in front of the sweep (+3 us per kernel)
    bool stage = g.lds_mats && ((long)a.batch * g.grid_x >= 1024 || g.team);
    if (const char *e = std::getenv("SPOCK_AMD_LDS_MATS")) stage = g.lds_mats && e[0] == '1';
    a.lds_mats = stage ? 1 : 0; a.team = g.team; a.team_reg = g.team_reg; a.ex_doubles = g.ex_doubles;
    a.xl_global = (g.team && g.xl_global && a.xlbuf) ? 1 : 0; a.xl_stride = g.xl_stride;
    if (g.team && g.xl_global && !a.xlbuf) return (int)hipErrorInvalidValue;
    if (P::TEAMLOCAL && !g.team_local && mode != MODE_FINALIZE) return (int)hipErrorInvalidConfiguration;
    const size_t lds = g.lds_bytes;   // sized for the staged matrices either way
    const dim3 grid(g.grid_x, a.batch), block(g.threads);
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: raise it once per kernel (160 KiB per CU on MI355X)
        static bool raised_dev[16][N_GROUP_MODES] = {};   // the attribute is per device (and per kernel = per template instance)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
        bool always = false;
        bool &raised_flag = dev >= 0 ? raised_dev[dev][mode] : always;
        if (!raised_flag) {
            const void *fn = mode == MODE_FULL ? (const void *)spock_group_kernel<P, MODE_FULL>
                           : mode == MODE_SUB_A ? (const void *)spock_group_kernel<P, MODE_SUB_A>
                           : mode == MODE_CROWN ? (const void *)spock_group_kernel<P, MODE_CROWN>
                           : mode == MODE_SUB_B ? (const void *)spock_group_kernel<P, MODE_SUB_B>
                           : mode == MODE_SUB_BA ? (const void *)spock_group_kernel<P, MODE_SUB_BA>
                           : mode == MODE_SUB_B_SP ? (const void *)spock_group_kernel<P, MODE_SUB_B_SP>
                           : mode == MODE_SUB_B_LAT ? (const void *)spock_group_kernel<P, P::REGROWS ? MODE_SUB_B_LAT : MODE_SUB_B>
                           : mode == MODE_FULL_SP ? (const void *)spock_group_kernel<P, MODE_FULL_SP>
                                                : (const void *)spock_group_kernel<P, MODE_FINALIZE>;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised_flag = true;
        }
    }
    switch (mode) {
        case MODE_FULL: hipLaunchKernelGGL((spock_group_kernel<P, MODE_FULL>), grid, block, lds, st, a); break;
        case MODE_SUB_A: hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_A>), grid, block, lds, st, a); break;
        case MODE_CROWN: hipLaunchKernelGGL((spock_group_kernel<P, MODE_CROWN>), grid, block, lds, st, a); break;
        case MODE_SUB_B: hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_B>), grid, block, lds, st, a); break;
        case MODE_FINALIZE: hipLaunchKernelGGL((spock_group_kernel<P, MODE_FINALIZE>), grid, block, lds, st, a); break;
        case MODE_SUB_BA:
            hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_BA>), grid, block, lds, st, a);
            break;
        case MODE_SUB_B_SP: hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_B_SP>), grid, block, lds, st, a); break;
        case MODE_SUB_B_LAT:   // exists for the shapes with the register-resident sweep; the others run plain SUB_B
            hipLaunchKernelGGL((spock_group_kernel<P, P::REGROWS ? MODE_SUB_B_LAT : MODE_SUB_B>), grid, block, lds, st, a);
            break;
        case MODE_FULL_SP: hipLaunchKernelGGL((spock_group_kernel<P, MODE_FULL_SP>), grid, block, lds, st, a); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}



int launch_flat_local(int mode, const KArgs &a, void *stream);   // flat node-local kernels (spock_fkernels.hip)
int launch_sweep(int mode, const KArgs &a, void *stream);

bool have_static_variant(int nx, int nu, int d, int ny) {
#define X(a, b, c, e) if (nx == a && nu == b && d == c && ny == e) return true;
    SPOCK_STATIC_SHAPES(X)
#undef X
    return false;
}

int launch_group(int mode, const KArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
    static const bool force_dyn = std::getenv("SPOCK_AMD_FORCE_DYN") != nullptr;   // experiments: run-time-dimension kernels for every shape
    if (a.ilv) return launch_igroup(mode, a, stream);   // element-major batched regime (spock_ikernels.hip)
    if (mode == MODE_SUB_BA_R) return launch_rgroup(a, stream);   // role-lane subtree kernel (spock_rkernels.hip)
    if (mode >= MODE_FLAT_A && mode <= MODE_FLAT_BA) return launch_flat_local(mode, a, stream);   // spock_fkernels.hip
    if (a.flat_local && cm == 2 && (mode == MODE_SUB_A || mode == MODE_CROWN || mode == MODE_SUB_B) && sweep_kernels_apply(dm))
        return launch_sweep(mode, a, stream);                                                      // spock_skernels.hip
#define X(a_, b_, c_, e_)                                                                      \
    if (!force_dyn && dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                  \
        return cm == 2 ? launch_variant<StaticDims<a_, b_, c_, e_, 2>>(mode, a, st) : cm == 1 ? launch_variant<StaticDims<a_, b_, c_, e_, 1>>(mode, a, st)    \
                             : launch_variant<StaticDims<a_, b_, c_, e_, 0>>(mode, a, st);
    SPOCK_STATIC_SHAPES(X)
#undef X
    if (cm == 2 && launch_geometry(mode == MODE_FINALIZE ? MODE_CROWN : mode, dm, a.batch).team_local) {
        static const bool no_mfma = std::getenv("SPOCK_AMD_NO_MFMA") != nullptr;
        if (((dm.nx == 64 && dm.nu == 16 && dm.d == 2) || (dm.nx == 20 && dm.nu == 20 && (dm.d == 2 || dm.d == 3))) && !no_mfma)
            return launch_variant<DynDims<2, true, true>>(mode, a, st);   // MFMA sweep levels (BASELINE configs 5, 4 and 3)
        return launch_variant<DynDims<2, true>>(mode, a, st);
    }
    return cm == 2 ? launch_variant<DynDims<2>>(mode, a, st) : cm == 1 ? launch_variant<DynDims<1>>(mode, a, st) : launch_variant<DynDims<0>>(mode, a, st);
}


// ===================================================================================================
// SP (SuperMann + Anderson) kernels: flat vector steps, node-local L / L' application, control steps
// ===================================================================================================
constexpr int kFlatThreads = 256, kFlatPerThread = 4, kCtlThreads = 256;

int sp_flat_blocks(long nr) {
    long b = (nr + (long)kFlatThreads * kFlatPerThread - 1) / ((long)kFlatThreads * kFlatPerThread);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
int sp_lop_blocks(const Dims &dm) { return dm.t0 > 0 ? dm.nsub + 1 : 1; }

// deterministic workgroup reduction of K per-thread values (sum or NaN-propagating max): fixed shuffle tree
// inside a wave, then the waves in order
template <int K, bool MAXOP>
__device__ __forceinline__ void block_reduce(double *v, double *lds /* [K][16] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(x, off, 64);
            x = MAXOP ? spk_max_nan(x, o) : x + o;
        }
        if (lane == 0) lds[k * 16 + wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x = lds[k * 16];
            for (int w = 1; w < nw; ++w) x = MAXOP ? spk_max_nan(x, lds[k * 16 + w]) : x + lds[k * 16 + w];
            v[k] = x;
        }
    }
}

__global__ void __launch_bounds__(kFlatThreads) spock_sp_flat_kernel(const SpArgs a) {
    const int inst = blockIdx.y;
    const int done = a.ctl[inst].done;
    if (done == 1 || (done && a.op != SP_FLAT5)) return;   // done == 2: the state update of the final iteration still runs
    __shared__ double red[9 * 16];
    const SpCtl c = a.sp[inst];
    // (the two column norms by their own loads from the control block: a dynamic index into the register copy `c` would send it to scratch)
    const int hd = a.sp[inst].head;
    const double rho2 = a.sp[inst].crho[(hd + 1) % 3], rho3 = a.sp[inst].crho[(hd + 2) % 3];
    double sums[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
    const long stride = (long)gridDim.x * kFlatThreads;
    for (long e = (long)blockIdx.x * kFlatThreads + threadIdx.x; e < a.nr; e += stride) sp_flat_elem(a, c, inst, e, sums, rho2, rho3);
    if (a.op == SP_FLAT1 || a.op == SP_FLAT2) {
        block_reduce<9, false>(sums, red);
        if (threadIdx.x == 0) {
            double *dst = a.partF + ((long)inst * a.n_partF + blockIdx.x) * kSpSums;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] = sums[k];
        }
    }
}

template <class P>
__global__ void __launch_bounds__(512) spock_sp_lop_kernel(const SpArgs a) {
    const int inst = blockIdx.y, g = blockIdx.x;
    if (a.ctl[inst].done) return;
    __shared__ double red[4 * 16];
    const Dims &dm = a.dm;
    const P p(dm);
    const bool split = dm.t0 > 0, crown = split && g == dm.nsub;
    TState<P> s;
    init_geometry(p, dm, (int)threadIdx.x, (split && !crown) ? dm.t0 : 0, (split && !crown) ? g : 0,
                  crown ? dm.t0 + 1 : (split ? dm.sub_stages : dm.N), crown, s);
    const long o = (long)inst * a.nr;
    const real *A = nullptr, *B = nullptr, *B2 = nullptr;
    if (a.op == SP_LOP_R) { A = a.R + o; B = a.R + o; }
    else if (a.op == SP_LOP_RT) { A = a.RW + o; B = a.RW + o; B2 = a.D + o; }
    else { B = a.DP + o; }
    double acc[4] = {0., 0., 0., 0.};
    if (P::TEAMLOCAL) {   // T lanes per node, no exchange between lanes: rounds over the group's nodes
        const int T = a.team, nteams = blockDim.x / T, tm = threadIdx.x / T, r = threadIdx.x - tm * T;
        const int stages = crown ? dm.t0 + 1 : (split ? dm.sub_stages : dm.N), nl_grp = group_threads_dev(p.d(), stages);
        for (int li = tm; li < nl_grp && tm < nteams; li += nteams) {
            const TeamNode tn = team_node(dm, p.d(), li, nl_grp, (split && !crown) ? dm.t0 : 0, (split && !crown) ? g : 0, stages, crown);
            tlop(p, dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), A, B, B2, tn, r, a.ctl[inst].gamma, a.ctl[inst].sigma, acc);
        }
    } else if (s.active) lop_local(p, dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), A, B, B2, s, a.ctl[inst].gamma, a.ctl[inst].sigma, acc);
    double mx[2] = {acc[2], acc[3]};
    block_reduce<2, false>(acc, red);
    __syncthreads();
    block_reduce<2, true>(mx, red);
    if (threadIdx.x == 0) {
        double *dst = a.partL + ((long)inst * a.n_partL + g) * kSpSums;
        dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = mx[0]; dst[3] = mx[1];
    }
}

__global__ void __launch_bounds__(kCtlThreads) spock_sp_ctl_kernel(const SpArgs a) {
    const int inst = blockIdx.y;
    Ctl &ctl = a.ctl[inst];
    if (a.op == SP_CTL_M) {   // closed loop on the device: the instance's solve has ended and FLAT5 has applied its last step
        if (ctl.done != 2) return;
        const Dims &dm = a.dm;
        const int t = a.mpc_t[inst], r = threadIdx.x;
        const real *z = a.Z + (long)inst * a.nr;
        real *x0 = a.mpc_x0 + (long)inst * dm.nx;
        const int w = a.mpc_w[(long)inst * a.mpc_steps + t];
        real xn = 0.;
        if (r < dm.nx) {
            if (t == 0) a.mpc_xt[((long)inst * (a.mpc_steps + 1)) * dm.nx + r] = x0[r];
            xn = mpc_advance_row(dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), z, 1, x0, w, r);       // u = z[u_inds[1:nu]], x0 <- A_w x0 + B_w u
        }
        if (r < dm.nu) a.mpc_ut[((long)inst * a.mpc_steps + t) * dm.nu + r] = z[dm.zu + r];
        __syncthreads();   // every row has read the old x0 (and every thread the old `done`)
        if (r < dm.nx) { x0[r] = xn; a.mpc_xt[((long)inst * (a.mpc_steps + 1) + t + 1) * dm.nx + r] = xn; }
        if (r == 0) sp_mpc_next(ctl, a.sp[inst], a.mpc_t + inst, a.mpc_steps, a.mpc_res + (long)inst * a.mpc_steps * kMpcRec);
        return;
    }
    if (ctl.done) {
        if (a.op == SP_CTL_A && ctl.done == 2 && threadIdx.x == 0) ctl.done = 1;   // the final state update has run
        if (a.op == SP_CTL_E && ctl.done == 3 && threadIdx.x == 0) ctl.done = 0;   // corrected variant: parked by an accepted candidate
        return;
    }
    if (a.op == SP_CTL_E) return;
    __shared__ double red[15 * 16];
    SpCtl &c = a.sp[inst];
    const double *pf = a.partF + (long)inst * a.n_partF * kSpSums, *pl = a.partL + (long)inst * a.n_partL * kSpSums;
    double f[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, l[2] = {0., 0.}, m[2] = {0., 0.};
    if (a.op == SP_CTL_A || a.op == SP_CTL_B)
        for (int r = threadIdx.x; r < a.n_partF; r += kCtlThreads)
            for (int k = 0; k < 9; ++k) f[k] += pf[(long)r * kSpSums + k];
    double g2[2] = {0., 0.};
    if ((a.op == SP_CTL_A || a.op == SP_CTL_C || a.op == SP_CTL_CD) && a.fused_lop) {
        const double *pg = a.partG + (long)inst * a.n_partG * kPartialStride;
        const double gam = ctl.gamma, sig = ctl.sigma;
        for (int r = threadIdx.x; r < a.n_partG; r += kCtlThreads) {   // <r, P r> = -gamma sum_z dz xi_1 - sigma sum_v dv xi_2
            l[0] += -gam * pg[(long)r * kPartialStride + 4] - sig * pg[(long)r * kPartialStride + 5];
            l[1] += pg[(long)r * kPartialStride + 6];
            m[0] = spk_max_nan(m[0], pg[(long)r * kPartialStride + 0]); m[1] = spk_max_nan(m[1], pg[(long)r * kPartialStride + 1]);
            g2[0] = spk_max_nan(g2[0], pg[(long)r * kPartialStride + 3]); g2[1] = spk_max_nan(g2[1], pg[(long)r * kPartialStride + 7]);
        }
    } else if (a.op == SP_CTL_A || a.op == SP_CTL_C)
        for (int r = threadIdx.x; r < a.n_partL; r += kCtlThreads) { l[0] += pl[(long)r * kSpSums]; l[1] += pl[(long)r * kSpSums + 1]; }
    if (a.op == SP_CTL_D)
        for (int r = threadIdx.x; r < a.n_partL; r += kCtlThreads) {
            m[0] = spk_max_nan(m[0], pl[(long)r * kSpSums + 2]); m[1] = spk_max_nan(m[1], pl[(long)r * kSpSums + 3]);
        }
    {   // one pass for all fifteen values (eleven sums, four maxima): fixed shuffle tree per wave, then the waves in order
        double val[15] = {f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], l[0], l[1], m[0], m[1], g2[0], g2[1]};
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (kCtlThreads + 63) >> 6;
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            double x = val[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_down(x, off, 64);
                x = (k >= 11) ? spk_max_nan(x, o) : x + o;
            }
            if (lane == 0) red[k * 16 + wave] = x;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                double x = red[k * 16];
                for (int w = 1; w < nw; ++w) x = (k >= 11) ? spk_max_nan(x, red[k * 16 + w]) : x + red[k * 16 + w];
                val[k] = x;
            }
            for (int k = 0; k < 9; ++k) f[k] = val[k];
            l[0] = val[9]; l[1] = val[10]; m[0] = val[11]; m[1] = val[12]; g2[0] = val[13]; g2[1] = val[14];
        }
    }
    if (threadIdx.x == 0) {
        double *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
        if (a.op == SP_CTL_CD && a.variant == 1) sp_ctl_trial(c, ctl, l[0], l[1], m[0], m[1], g2[0], g2[1], a.tau, a.trial, a.last_trial, hist, a.hist_cap);
        else if (a.op == SP_CTL_CD) sp_ctl_cd(c, ctl, l[0], l[1], m[0], m[1], g2[0], g2[1], hist, a.hist_cap);
        else if (a.op == SP_CTL_A && a.variant == 1) sp_ctl_a_ls(c, f, l[0], m[0], m[1]);
        else if (a.op == SP_CTL_A) sp_ctl_a(c, f, l[0], m[0], m[1]);
        else if (a.op == SP_CTL_B) sp_ctl_b(c, f);
        else if (a.op == SP_CTL_C) sp_ctl_c(c, ctl, l[0], l[1]);
        else if (a.op == SP_CTL_D) sp_ctl_d(c, ctl, m[0], m[1], a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr, a.hist_cap);
    }
}

__global__ void __launch_bounds__(64) spock_mpc_advance_kernel(const MpcArgs a) {
    const int inst = blockIdx.x, r = threadIdx.x;
    const Dims &dm = a.dm;
    const real *z = a.z + (long)inst * a.stride_z;
    real *x0 = a.x0 + (long)inst * dm.nx;
    const int w = a.w_seq[(long)inst * a.steps + a.t];
    real xn = 0.;
    if (r < dm.nx) {
        if (a.t == 0) a.x_traj[((long)inst * (a.steps + 1)) * dm.nx + r] = x0[r];
        xn = mpc_advance_row(dm, (MatPtr)(a.mats + (long)inst * a.mats_stride), z, a.stride_e, x0, w, r);
    }
    if (r < dm.nu) a.u_traj[((long)inst * a.steps + a.t) * dm.nu + r] = z[(long)(dm.zu + r) * a.stride_e];
    __syncthreads();   // every row has read the old x0
    if (r < dm.nx) { x0[r] = xn; a.x_traj[((long)inst * (a.steps + 1) + a.t + 1) * dm.nx + r] = xn; }
}

int launch_mpc_advance(const MpcArgs &a, void *stream) {
    hipLaunchKernelGGL(spock_mpc_advance_kernel, dim3(a.batch), dim3(64), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

template <class P>
static int launch_lop(const SpArgs &a_in, hipStream_t st) {
    const Dims &dm = a_in.dm;
    const int sub_thr = group_threads(dm.d, dm.sub_stages), crown_thr = group_threads(dm.d, dm.t0 + 1);
    int nthr = round64(dm.t0 > 0 ? (sub_thr > crown_thr ? sub_thr : crown_thr) : group_threads(dm.d, dm.N));
    SpArgs a = a_in;
    a.team = 0;
    if (P::TEAMLOCAL) { a.team = 8; while (a.team < dm.nx || a.team < dm.nu) a.team *= 2; nthr = 512; }
    hipLaunchKernelGGL((spock_sp_lop_kernel<P>), dim3(sp_lop_blocks(dm), a.batch), dim3(nthr), 0, st, a);
    return (int)hipGetLastError();
}

int launch_sp(const SpArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    switch (a.op) {
        case SP_FLAT1: case SP_FLAT2: case SP_FLAT3: case SP_FLAT4: case SP_FLAT5: case SP_FLATW:
            hipLaunchKernelGGL(spock_sp_flat_kernel, dim3(a.n_partF, a.batch), dim3(kFlatThreads), 0, st, a);
            return (int)hipGetLastError();
        case SP_CTL_A: case SP_CTL_B: case SP_CTL_C: case SP_CTL_D: case SP_CTL_CD: case SP_CTL_E: case SP_CTL_M:
            hipLaunchKernelGGL(spock_sp_ctl_kernel, dim3(1, a.batch), dim3(kCtlThreads), 0, st, a);
            return (int)hipGetLastError();
        default: break;
    }
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
#define X(a_, b_, c_, e_)                                                          \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                    \
        return cm == 2 ? launch_lop<StaticDims<a_, b_, c_, e_, 2>>(a, st)           \
             : cm == 1 ? launch_lop<StaticDims<a_, b_, c_, e_, 1>>(a, st) : launch_lop<StaticDims<a_, b_, c_, e_, 0>>(a, st);
    SPOCK_STATIC_SHAPES(X)
#undef X
    if (cm == 2 && team_local_applies(dm)) return launch_lop<DynDims<2, true>>(a, st);
    return cm == 2 ? launch_lop<DynDims<2>>(a, st) : cm == 1 ? launch_lop<DynDims<1>>(a, st) : launch_lop<DynDims<0>>(a, st);
}

}  // namespace spock
