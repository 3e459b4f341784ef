// spock_skernels.hip -- gfx950 SWEEP kernels of the lane-team shapes whose S1 levels run on the matrix cores (nx = 64 / nu = 16 / d = 2:
// BASELINE config 5; nx = nu = 20, d = 2 or 3: configs 4 and 3), used when the node-local phases run as flat kernels
// (spock_flat_local.hpp, KArgs::flat_local).  They do what MODE_SUB_A / MODE_CROWN / MODE_SUB_B of spock_group.hpp do in that mode --
// the backward sweep of a subtree (projection_S1!, implicit_l.jl:584-637), termination test + both sweeps through the crown, the
// forward roll-out of a subtree (implicit_l.jl:667-681) -- and nothing else: the general group kernel carries the node-local
// phases of every regime in the same function (700+ bytes of scratch per lane, 128 registers), and its matrix-core levels paid for
// it (measured: 5 us per level whatever the level's width).  Here a kernel is: coalesced copies of the sweep inputs zbar.x / zbar.u
// -> LDS, the level loop of bwd_level_mfma / fwd_level_mfma (spock_device.hpp) with the level's matrix fragments in registers, and
// coalesced copies out.  The group leaves' vector of the UP kernel is always staged in LDS (the general kernel reads it from global
// memory inside the matrix-core loop when all four sweep vectors do not fit: 23 us for that one level at nx = 64).
// A translation unit of its own (compiled side by side with the others).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_device.hpp"
#include "spock_gpu_exec.hpp"

namespace spock {

// the policy the level functions are instantiated with (they only read the compile-time shape from their template arguments)
typedef DynDims<2, true, true> SweepP;

constexpr int kSweepThreads = 1024;
enum { SW_UP = 0, SW_CROWN = 1, SW_DOWN = 2 };

template <class D, class S>
__device__ __forceinline__ void sw_copy(D dst, S src, int n, int tid, int nthr) {
    int k = tid;
    for (; k + 3 * nthr < n; k += 4 * nthr) {
        const real a0 = src[k], a1 = src[k + nthr], a2 = src[k + 2 * nthr], a3 = src[k + 3 * nthr];
        dst[k] = a0; dst[k + nthr] = a1; dst[k + 2 * nthr] = a2; dst[k + 3 * nthr] = a3;
    }
    for (; k < n; k += nthr) dst[k] = src[k];
}

// rows of W elements between an array with row stride SP (LDS, padded) and one with stride W (global memory, contiguous)
template <int W, int SP, bool TO_LDS, class D, class S>
__device__ __forceinline__ void sw_copy_rows(D dst, S src, int rows, int tid, int nthr) {
    const int n = rows * W;
    int k = tid;
    auto di = [&](int e) { return TO_LDS ? (e / W) * SP + (e % W) : e; };
    auto si = [&](int e) { return TO_LDS ? e : (e / W) * SP + (e % W); };
    for (; k + 3 * nthr < n; k += 4 * nthr) {
        const real a0 = src[si(k)], a1 = src[si(k + nthr)], a2 = src[si(k + 2 * nthr)], a3 = src[si(k + 3 * nthr)];
        dst[di(k)] = a0; dst[di(k + nthr)] = a1; dst[di(k + 2 * nthr)] = a2; dst[di(k + 3 * nthr)] = a3;
    }
    for (; k < n; k += nthr) dst[di(k)] = src[si(k)];
}

__device__ __forceinline__ int sw_nonleaf(int d, int stages) {
    int cnt = 0, lvl = 1;
    for (int t = 0; t < stages - 1; ++t) { cnt += lvl; lvl *= d; }
    return cnt;
}

// LDS carve (doubles), node vectors with padded strides SX = nx + 4, SU = nu + 4 (LDS banks, see bwd_level_mfma):
// [XQ nl*SX][ED nl*SU][UL nl*SU (crown, down)][XL ngl*SX when it fits] then the reduction scratch of GpuExec
// (nx = nu = 50: pad 3 -- an odd stride, and 127-node groups with three vectors per node then just fit the 160 KB of LDS)
constexpr int sweep_pad(int n) { return n == 50 ? 3 : 4; }
template <int NX, int NU, int D, int MODE>
__global__ void __launch_bounds__(kSweepThreads) spock_sweep_kernel(const KArgs a, const int xl_up_lds, const int xl_parts) {
    // xl_up_lds: on the way up the group leaves' q is staged in LDS (behind ED: the u vector does not exist yet);
    // xl_parts : on the way down the group leaves' x goes through an LDS staging area behind UL in this many parts (1: all at once;
    //            0: no room at all, the matrix cores' result lanes write it to global memory themselves)
    constexpr int SX = NX + sweep_pad(NX), SU = NU + sweep_pad(NU);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = a.dm;
    const int inst = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    Ctl &ctl = a.ctl[inst];
    const SweepP p(dm);
    const bool crown = (MODE == SW_CROWN);
    const int root_stage = crown ? 0 : dm.t0, root_pos = crown ? 0 : g;
    const int stages = crown ? dm.t0 + 1 : dm.sub_stages;
    const int nl = sw_nonleaf(D, stages);
    int ngl = 1;
    for (int k = 0; k < stages - 1; ++k) ngl *= D;
    real *lds = reinterpret_cast<real *>(smem);
    TeamLds tl;
    tl.XQ = lds; tl.ED = tl.XQ + nl * SX; tl.UL = tl.ED + nl * SU;
    real *xl_up = tl.UL, *xl_down = tl.UL + nl * SU;
    real *zo = a.z_out + (long)inst * a.stride_z;
    const long gl_first = dm.first[root_stage + stages - 1] + (long)root_pos * ngl;
    real *qroot = a.qroot + (long)inst * dm.nsub * NX;
    const MatPtr M = (MatPtr)(a.mats + (long)inst * a.mats_stride);   // (per-instance pools: spock_update_dynamics_batch)

    // ---- control: finished instance -> exit; the crown also runs the termination test of the previous iteration (cp.jl:54-123)
    if (MODE == SW_CROWN && !(a.flags & F_SP) && !(a.flags & F_NO_CTL)) {
        GpuExec<SweepP> x;
        x.red = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles);
        x.fl = reinterpret_cast<int *>(x.red + 3);
        const int n_groups = a.flat_chunks;
        const double *partials = a.partials + (long)inst * n_groups * kPartialStride;
        const CtlSnap snap = ctl_snapshot(ctl);
        double mx[3] = {0., 0., 0.};
        for (int gg = tid; gg < n_groups; gg += nthr)
            for (int k = 0; k < 3; ++k) mx[k] = spk_max_nan(mx[k], partials[gg * kPartialStride + k]);
        x.stash3(tid, mx);
        if (snap.pending && !snap.done) x.reduce3();   // (every thread has taken its snapshot before thread 0 updates the block)
        else x.barrier();
        if (tid == 0) {
            double *hist = a.hist ? a.hist + (long)inst * a.hist_cap * kHistStride : nullptr;
            x.flag() = ctl_prologue(ctl, snap, x.red3(0), x.red3(1), x.red3(2), hist, a.hist_cap, a.parity, true) ? 1 : 0;
        }
        x.barrier();
        if (x.flag()) return;
    } else if (!(a.flags & F_NO_CTL)) {
        if (ctl.done) return;
    }

    auto level_geo = [&](int l, int &lfirst, int &cnt, long &gfirst) {
        cnt = 1; lfirst = 0;
        for (int k = 0; k < l; ++k) { lfirst += cnt; cnt *= D; }
        gfirst = dm.first[root_stage + l] + (long)root_pos * cnt;
    };

    // ---- backward sweep (UP, CROWN): c_i, e_i from zbar.x / zbar.u (parked there by the flat phase-A kernel), q of the group leaves
    //      from zbar.x of the tree leaves (subtrees) / qroot (crown)
    if (MODE != SW_DOWN) {
        for (int l = 0; l <= stages - 2; ++l) {
            int lfirst, cnt; long gfirst;
            level_geo(l, lfirst, cnt, gfirst);
            sw_copy_rows<NX, SX, true>(tl.XQ + lfirst * SX, zo + dm.zx + gfirst * NX, cnt, tid, nthr);
            sw_copy_rows<NU, SU, true>(tl.ED + lfirst * SU, zo + dm.zu + gfirst * NU, cnt, tid, nthr);
        }
        const real *xl_src = crown ? (const real *)qroot : (const real *)(zo + dm.zx + gl_first * NX);
        if (xl_up_lds) { sw_copy_rows<NX, SX, true>(xl_up, xl_src, ngl, tid, nthr); tl.XL = xl_up; }
        else tl.XL = const_cast<real *>(xl_src);
        const int sxl = xl_up_lds ? SX : NX;                        // the leaves' vector in global memory is contiguous
        __syncthreads();
        for (int l = stages - 2; l >= 0; --l) {
            int lfirst, cnt; long gfirst;
            level_geo(l, lfirst, cnt, gfirst);
            const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
            bwd_level_mfma<NX, NU, D, SX, SU>(p, dm, SW, lfirst, cnt, l == stages - 2, tid, nthr, tl, sxl);
            __syncthreads();
        }
        if (MODE == SW_UP) {   // q at the subtree root and every d_i -> HBM (level chunks are contiguous)
            real *q_out = qroot + (long)g * NX;
            real *db = a.dbuf + (long)inst * dm.nn * NU;
            if (tid < NX) q_out[tid] = tl.XQ[tid];
            for (int l = 0; l <= stages - 2; ++l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                sw_copy_rows<NU, SU, false>(db + gfirst * NU, tl.ED + lfirst * SU, cnt, tid, nthr);
            }
            return;
        }
    }

    // ---- forward roll-out (CROWN, DOWN) from the group root; x_i, u_i and the group leaves' x go to zbar
    {
        const real *xroot = crown ? a.x0 + (long)inst * NX : zo + dm.zx + (long)(dm.first[root_stage] + root_pos) * NX;
        if (MODE == SW_DOWN) {
            const real *db = a.dbuf + (long)inst * dm.nn * NU;
            for (int l = 0; l <= stages - 2; ++l) {
                int lfirst, cnt; long gfirst;
                level_geo(l, lfirst, cnt, gfirst);
                sw_copy_rows<NU, SU, true>(tl.ED + lfirst * SU, db + gfirst * NU, cnt, tid, nthr);
            }
        }
        if (tid < NX) tl.XQ[tid] = xroot[tid];
        real *xl_dst = zo + dm.zx + gl_first * NX;
        __syncthreads();
        const MatPtr gB = M + dm.mB;
        for (int l = 0; l <= stages - 2; ++l) {
            int lfirst, cnt; long gfirst;
            level_geo(l, lfirst, cnt, gfirst);
            const MatPtr SW = M + dm.mSweep + (root_stage + l) * dm.sw_stride;
            if (l < stages - 2) {
                fwd_level_mfma<NX, NU, D, SX, SU>(p, dm, SW, gB, lfirst, cnt, false, tid, nthr, tl, SX);
                __syncthreads();
            } else if (xl_parts == 0) {   // the group leaves' x straight to global memory
                tl.XL = xl_dst;
                fwd_level_mfma<NX, NU, D, SX, SU>(p, dm, SW, gB, lfirst, cnt, true, tid, nthr, tl, NX);
                __syncthreads();
            } else {                      // ... through LDS, in parts: the matrix cores' result lanes scatter 32-byte pieces, the copy is coalesced
                const int per = cnt / xl_parts;
                tl.XL = xl_down;
                for (int h = 0; h < xl_parts; ++h) {
                    fwd_level_mfma<NX, NU, D, SX, SU>(p, dm, SW, gB, lfirst + h * per, per, true, tid, nthr, tl, SX);
                    __syncthreads();
                    sw_copy_rows<NX, SX, false>(xl_dst + (long)h * per * D * NX, xl_down, per * D, tid, nthr);
                    if (h + 1 < xl_parts) __syncthreads();
                }
            }
        }
        for (int l = 0; l <= stages - 2; ++l) {
            int lfirst, cnt; long gfirst;
            level_geo(l, lfirst, cnt, gfirst);
            sw_copy_rows<NX, SX, false>(zo + dm.zx + gfirst * NX, tl.XQ + lfirst * SX, cnt, tid, nthr);
            sw_copy_rows<NU, SU, false>(zo + dm.zu + gfirst * NU, tl.UL + lfirst * SU, cnt, tid, nthr);
        }
    }
}

template <int NX, int NU, int D>
static int launch_sweep_shape(int mode, const KArgs &a_in, hipStream_t st) {
    const Dims &dm = a_in.dm;
    KArgs a = a_in;
    const int sw = mode == MODE_SUB_A ? SW_UP : (mode == MODE_CROWN ? SW_CROWN : SW_DOWN);
    const int stages = sw == SW_CROWN ? dm.t0 + 1 : dm.sub_stages;
    int nl = 0, ngl = 1;
    { int lvl = 1; for (int t = 0; t < stages - 1; ++t) { nl += lvl; lvl *= D; } ngl = lvl; }
    constexpr int SX = NX + sweep_pad(NX), SU = NU + sweep_pad(NU);
    const long tail = 4 + 64;                                                         // GpuExec's reduction scratch (crown)
    const long budget = 160 * 1024 / 8 - 64;
    const long up_base = (long)nl * (SX + SU), down_base = (long)nl * (SX + 2 * SU), xl_full = (long)ngl * SX;
    if ((sw == SW_UP ? up_base : down_base) + tail > budget) return (int)hipErrorInvalidConfiguration;
    // way up (UP, CROWN): XQ, ED + the staged leaves' vector; way down (CROWN, DOWN): XQ, ED, UL + a staging area for the leaves' x
    const int xl_up_lds = (sw != SW_DOWN && up_base + xl_full + tail <= budget) ? 1 : 0;
    int xl_parts = 0;
    long need = sw == SW_UP ? up_base + (xl_up_lds ? xl_full : 0) : down_base;
    if (sw != SW_UP) {
        const int bottom_cnt = ngl / D;                                               // nodes of the group's bottom non-leaf level
        for (int parts = 1; parts <= bottom_cnt && parts <= 8; parts *= 2)
            if (bottom_cnt % parts == 0 && down_base + xl_full / parts + tail <= budget) { xl_parts = parts; break; }
        if (xl_parts) need = down_base + xl_full / xl_parts;
        if (sw == SW_CROWN && xl_up_lds && up_base + xl_full > need) need = up_base + xl_full;
    }
    a.ex_doubles = (int)need;
    const size_t lds = (size_t)(a.ex_doubles + tail) * 8;
    // workgroup size: 16 waves when few workgroups are in flight (every row tile and node-tile share of a level gets a wave of its own);
    // 4 or 8 when a batch fills the chip with workgroups anyway -- the narrow levels of a sweep have 4 - 8 row-tile tasks, and smaller
    // workgroups let a CU interleave more of these latency-bound level chains
    int threads = kSweepThreads;
    const long wgs = (long)a.batch * (sw == SW_CROWN ? 1 : dm.nsub);
    if (NX <= 32 && wgs >= 1024) threads = 256;
    else if (NX <= 32 && wgs >= 256) threads = 512;
    if (const char *e = std::getenv("SPOCK_AMD_SWEEP_THREADS")) { threads = std::atoi(e); if (threads < 64) threads = 64; if (threads > 1024) threads = 1024; threads = (threads / 64) * 64; }
    const dim3 grid(sw == SW_CROWN ? 1 : dm.nsub, a.batch), block(threads);
    static bool raised[16][3] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (lds > 64 * 1024 && !raised[dev][sw]) {
        const void *fn = sw == SW_UP ? (const void *)spock_sweep_kernel<NX, NU, D, SW_UP>
                       : sw == SW_CROWN ? (const void *)spock_sweep_kernel<NX, NU, D, SW_CROWN> : (const void *)spock_sweep_kernel<NX, NU, D, SW_DOWN>;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        raised[dev][sw] = true;
    }
    switch (sw) {
        case SW_UP: hipLaunchKernelGGL((spock_sweep_kernel<NX, NU, D, SW_UP>), grid, block, lds, st, a, xl_up_lds, xl_parts); break;
        case SW_CROWN: hipLaunchKernelGGL((spock_sweep_kernel<NX, NU, D, SW_CROWN>), grid, block, lds, st, a, xl_up_lds, xl_parts); break;
        default: hipLaunchKernelGGL((spock_sweep_kernel<NX, NU, D, SW_DOWN>), grid, block, lds, st, a, xl_up_lds, xl_parts); break;
    }
    return (int)hipGetLastError();
}

// true when the dedicated sweep kernels exist for this problem (flat node-local kernels + a matrix-core shape; either precision)
bool sweep_kernels_apply(const Dims &dm) {
    static const bool off = std::getenv("SPOCK_AMD_NO_SWEEPK") != nullptr || std::getenv("SPOCK_AMD_NO_MFMA") != nullptr;
    if (off || dm.t0 <= 0) return false;
    return (dm.nx == 64 && dm.nu == 16 && dm.d == 2) || (dm.nx == 20 && dm.nu == 20 && (dm.d == 2 || dm.d == 3)) ||
           (dm.nx == 50 && dm.nu == 50 && dm.d == 2);
}

int launch_sweep(int mode, const KArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    if (dm.nx == 64) return launch_sweep_shape<64, 16, 2>(mode, a, st);
    if (dm.nx == 50) return launch_sweep_shape<50, 50, 2>(mode, a, st);
    if (dm.d == 2) return launch_sweep_shape<20, 20, 2>(mode, a, st);
    return launch_sweep_shape<20, 20, 3>(mode, a, st);
}

}  // namespace spock
