// spock_lkernels.hip -- gfx950 kernels of the level-streaming part of the element-major batched regime (spock_ilevel.hpp): role
// waves, one launch per tree stage.  A translation unit of its own (compiled side by side with spock_ikernels.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_ilevel.hpp"
#include "spock_shapes.hpp"

namespace spock {
template <class P>
struct GpuIExec {
    IState<P> st;
    real *ex;
    double *rd;      // residual maxima across the waves: double whatever `real` is
    int *lf;
    template <class F>
    __device__ __forceinline__ void each(F &&f) { f((int)threadIdx.x, st); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    // role waves (spock_ilevel.hpp): the body holds the barrier itself, once on every wave-uniform path
    template <class F>
    __device__ __forceinline__ void phased(F &&f) { f((int)threadIdx.x, st, true, true); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ real *lds() { return ex; }
    __device__ __forceinline__ double *red() { return rd; }
    __device__ __forceinline__ int *live_flags() { return lf; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
};

// Level-streaming kernels (spock_ilevel.hpp): one wave per role, 3 .. 7 waves per workgroup.  SPOCK_ILV_LEVEL_WAVES: waves per
// SIMD they are compiled for (register budget 512 / that).
#ifndef SPOCK_ILV_LEVEL_WAVES
#define SPOCK_ILV_LEVEL_WAVES 3
#endif
template <class P, int MODE, bool WX>
__global__ void __launch_bounds__(64 * (2 + 2 * P::MAXD)) __attribute__((amdgpu_waves_per_eu(MODE == MODE_LA0 ? 1 : SPOCK_ILV_LEVEL_WAVES)))
spock_ilevel_kernel(const KArgs a, const real *__restrict__ zin, const real *__restrict__ vin, real *__restrict__ zout,
                    real *__restrict__ vout, real *__restrict__ znext) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpuIExec<P> x;
    x.ex = reinterpret_cast<real *>(smem);
    x.rd = smem;   // the residual maxima reuse the exchange rows (behind a barrier)
    x.lf = nullptr;
    ilevel_body<P, MODE, WX>(x, a, zin, vin, zout, vout, znext, (int)blockIdx.y + a.tile0, (int)blockIdx.x);
}

template <class P>
static int launch_lvariant(int mode, const KArgs &a_in, hipStream_t st) {
    const IlvGeo g = ilv_geometry(mode, a_in.dm, a_in.grp_level, a_in.want_xi);
    KArgs a = a_in;
    a.ex_doubles = g.ex_doubles; a.cchunk = g.cchunk; a.n_cchunks = g.n_cchunks;
    a.ls_cnt = g.ls_cnt; a.ls_first = g.ls_first; a.ls_first_child = g.ls_first_child; a.ls_slot = g.ls_slot; a.ls_chunk_k = g.ls_chunk_k;
    const dim3 grid(g.grid_x, (unsigned)(a.n_tiles > 0 ? a.n_tiles : a.bp / 64)), block(g.threads);
    const size_t lds = g.lds_bytes;
    if (lds > 160 * 1024) return (int)hipErrorInvalidConfiguration;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit (the exchange rows of the wider shapes): raise it once per kernel and device
        static bool raised_dev[16][5][2] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
        bool always = false;
        bool &raised_flag = dev >= 0 ? raised_dev[dev][mode == MODE_LBA2 ? 4 : mode - MODE_LB][a.want_xi ? 1 : 0] : always;
        if (!raised_flag) {
            const void *fn = mode == MODE_LB ? (a.want_xi ? (const void *)spock_ilevel_kernel<P, MODE_LB, true> : (const void *)spock_ilevel_kernel<P, MODE_LB, false>)
                           : mode == MODE_LBA2 ? (a.want_xi ? (const void *)spock_ilevel_kernel<P, MODE_LBA2, true> : (const void *)spock_ilevel_kernel<P, MODE_LBA2, false>)
                           : mode == MODE_LBA ? (a.want_xi ? (const void *)spock_ilevel_kernel<P, MODE_LBA, true> : (const void *)spock_ilevel_kernel<P, MODE_LBA, false>)
                           : mode == MODE_LA ? (const void *)spock_ilevel_kernel<P, MODE_LA, false> : (const void *)spock_ilevel_kernel<P, MODE_LA0, false>;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised_flag = true;
        }
    }
    const real *cn = nullptr;
    real *dn = nullptr;
    switch (mode) {
        case MODE_LB:
            if (a.want_xi) hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LB, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, dn);
            else hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LB, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, dn);
            break;
        case MODE_LBA:
            if (a.want_xi) hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LBA, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            else hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LBA, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            break;
        case MODE_LBA2:
            if (a.want_xi) hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LBA2, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            else hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LBA2, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            break;
        case MODE_LA: hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LA, false>), grid, block, lds, st, a, cn, cn, a.z_out, a.v_out, a.z_next); break;
        case MODE_LA0: hipLaunchKernelGGL((spock_ilevel_kernel<P, MODE_LA0, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, dn, dn); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int launch_ilevel(int mode, const KArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
#define X(a_, b_, c_, e_)                                                                      \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                                \
        return cm == 2 ? launch_lvariant<StaticDims<a_, b_, c_, e_, 2>>(mode, a, st) : cm == 1 ? launch_lvariant<StaticDims<a_, b_, c_, e_, 1>>(mode, a, st)  \
                             : (int)hipErrorInvalidConfiguration;
    SPOCK_STATIC_SHAPES(X)
#undef X
    return (int)hipErrorInvalidConfiguration;
}

}  // namespace spock
