// spock_fkernels.hip -- gfx950 kernels of the flat node-local phases of the lane-team shapes (spock_flat_local.hpp).  A translation
// unit of its own (compiled side by side with spock_kernels.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_flat_local.hpp"
#include "spock_gpu_exec.hpp"

namespace spock {

// flat node-local kernels of the lane-team shapes (spock_flat_local.hpp): blockIdx.x = chunk of 16 consecutive non-leaf nodes
#ifndef FLAT_WAVES          // experiment builds: waves per SIMD the phase-B kernels are compiled for (register budget 512 / that)
#define FLAT_WAVES 0
#endif
#if FLAT_WAVES
#define SPK_FLAT_ATTR __attribute__((amdgpu_waves_per_eu(FLAT_WAVES, FLAT_WAVES)))
#else
#define SPK_FLAT_ATTR
#endif
template <class P, int MODE>
__global__ void __launch_bounds__(256) SPK_FLAT_ATTR spock_flat_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpuExec<P> x;
    x.ex = reinterpret_cast<real *>(smem);
    x.red = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles);
    x.fl = reinterpret_cast<int *>(x.red + 3);
    x.stamps = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles + 4);
    x.sums = reinterpret_cast<double *>(smem + a.ex_doubles + 4 + 128);
    x.lm = nullptr;
    flat_body<P, MODE>(x, a, (int)blockIdx.y, (int)blockIdx.x);
}

template <class P>
static int launch_flat(int mode, const KArgs &a_in, hipStream_t st) {
    if (!P::TEAMLOCAL) return (int)hipErrorInvalidConfiguration;
    const FlatGeo g = flat_geometry(a_in.dm);
    KArgs a = a_in;
    a.team = g.team; a.ex_doubles = g.ex_doubles; a.flat_chunks = g.chunks;
    if (g.lds_bytes > 160 * 1024) return (int)hipErrorInvalidConfiguration;
    const dim3 grid(g.chunks, a.batch), block(g.threads);
    const int mi = mode - MODE_FLAT_A;
    if (g.lds_bytes > 64 * 1024) {   // wide shapes (nx, nu ~ 64: the staged U_t, Rt^-1_t alone are 50 KB): raise the dynamic-LDS limit once per kernel and device
        static bool raised[16][4] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (!raised[dev][mi]) {
            const void *fn = mode == MODE_FLAT_A ? (const void *)spock_flat_kernel<P, MODE_FLAT_A> : mode == MODE_FLAT_B ? (const void *)spock_flat_kernel<P, MODE_FLAT_B>
                           : mode == MODE_FLAT_B_SP ? (const void *)spock_flat_kernel<P, MODE_FLAT_B_SP> : (const void *)spock_flat_kernel<P, MODE_FLAT_BA>;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised[dev][mi] = true;
        }
    }
    switch (mode) {
        case MODE_FLAT_A: hipLaunchKernelGGL((spock_flat_kernel<P, MODE_FLAT_A>), grid, block, g.lds_bytes, st, a); break;
        case MODE_FLAT_B: hipLaunchKernelGGL((spock_flat_kernel<P, MODE_FLAT_B>), grid, block, g.lds_bytes, st, a); break;
        case MODE_FLAT_B_SP: hipLaunchKernelGGL((spock_flat_kernel<P, MODE_FLAT_B_SP>), grid, block, g.lds_bytes, st, a); break;
        case MODE_FLAT_BA: hipLaunchKernelGGL((spock_flat_kernel<P, MODE_FLAT_BA>), grid, block, g.lds_bytes, st, a); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}


int launch_flat_local(int mode, const KArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
    if (cm != 2 || !team_local_applies(dm)) return (int)hipErrorInvalidConfiguration;
    // lane-team policies with the branching factor compiled in (FlatDims, spock_device.hpp)
    return dm.d == 2 ? launch_flat<FlatDims<2>>(mode, a, st) : dm.d == 3 ? launch_flat<FlatDims<3>>(mode, a, st) : launch_flat<FlatDims<4>>(mode, a, st);
}

}  // namespace spock
