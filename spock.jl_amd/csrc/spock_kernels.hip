// spock_kernels.hip -- gfx950 kernels of the fused Chambolle-Pock iteration and their launchers.
// The per-thread bodies live in spock_device.hpp, the per-workgroup orchestration in spock_group.hpp;
// this file supplies the GPU execution policy (state in registers, LDS exchange buffer, wave/LDS max
// reduction), instantiates the kernels for the compile-time shapes of the benchmark configurations and
// for run-time dimensions, and launches them on the handle's stream.
#include <hip/hip_runtime.h>

#include "spock_group.hpp"

namespace spock {

template <class P>
struct GpuExec {
    TState<P> st;
    double m[3];
    double *ex;
    unsigned long long *red;
    int *fl;
    template <class F>
    __device__ __forceinline__ void each(F &&f) { f((int)threadIdx.x, st); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    __device__ __forceinline__ double *lds() { return ex; }
    __device__ __forceinline__ int &flag() { return *fl; }
    __device__ __forceinline__ void stash3(int, const double *mx) { m[0] = mx[0]; m[1] = mx[1]; m[2] = mx[2]; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ void clocks(unsigned long long *d) { d[0] = __builtin_readcyclecounter(); d[1] = __builtin_amdgcn_s_memrealtime(); }
    // values are |.| (sign bit clear) or NaN: the unsigned order of the bit patterns is the
    // NaN-propagating max order, so an integer max does the reduction
    __device__ __forceinline__ void reduce3() {
        __syncthreads();
        if (threadIdx.x < 3) red[threadIdx.x] = 0ull;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned long long b = (unsigned long long)__double_as_longlong(m[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(b, off, 64);
                b = o > b ? o : b;
            }
            if ((threadIdx.x & 63) == 0) atomicMax(&red[k], b);
        }
        __syncthreads();
    }
    __device__ __forceinline__ double red3(int k) const { return __longlong_as_double((long long)red[k]); }
};

template <class P, int MODE>
__global__ void __launch_bounds__(512) spock_group_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpuExec<P> x;
    x.ex = smem;
    x.red = reinterpret_cast<unsigned long long *>(smem + (size_t)blockDim.x * a.dm.nx);
    x.fl = reinterpret_cast<int *>(x.red + 3);
    group_body<P, MODE>(x, a, (int)blockIdx.y, (int)blockIdx.x);
}

static inline int round64(int t) { return t < 64 ? 64 : ((t + 63) / 64) * 64; }

template <class P>
static int launch_variant(int mode, const KArgs &a, hipStream_t st) {
    const Dims &dm = a.dm;
    int nthr, gx;
    const int sub_thr = group_threads(dm.d, dm.sub_stages), crown_thr = group_threads(dm.d, dm.t0 + 1);
    switch (mode) {
        case MODE_FULL: nthr = group_threads(dm.d, dm.N); gx = 1; break;
        case MODE_SUB_A: case MODE_SUB_B: nthr = sub_thr > crown_thr ? sub_thr : crown_thr; gx = dm.nsub + 1; break;
        case MODE_CROWN: nthr = crown_thr; gx = 1; break;
        default: nthr = 64; gx = 1; break;
    }
    const int thr = round64(nthr);
    if (thr > 512) return (int)hipErrorInvalidConfiguration;
    const size_t lds = ((size_t)thr * dm.nx + 4) * sizeof(double);
    const dim3 grid(gx, a.batch), block(thr);
    switch (mode) {
        case MODE_FULL: hipLaunchKernelGGL((spock_group_kernel<P, MODE_FULL>), grid, block, lds, st, a); break;
        case MODE_SUB_A: hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_A>), grid, block, lds, st, a); break;
        case MODE_CROWN: hipLaunchKernelGGL((spock_group_kernel<P, MODE_CROWN>), grid, block, lds, st, a); break;
        case MODE_SUB_B: hipLaunchKernelGGL((spock_group_kernel<P, MODE_SUB_B>), grid, block, lds, st, a); break;
        case MODE_FINALIZE: hipLaunchKernelGGL((spock_group_kernel<P, MODE_FINALIZE>), grid, block, lds, st, a); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

// compile-time shapes: (nx, nu, d, ny)
#define SPOCK_STATIC_SHAPES(X) X(5, 2, 2, 5) X(5, 5, 2, 5) X(2, 1, 2, 5) X(3, 2, 3, 7)

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
#define X(a_, b_, c_, e_)                                                                      \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                                \
        return cm == 2 ? launch_variant<StaticDims<a_, b_, c_, e_, 2>>(mode, a, st) : cm == 1 ? launch_variant<StaticDims<a_, b_, c_, e_, 1>>(mode, a, st)    \
                             : launch_variant<StaticDims<a_, b_, c_, e_, 0>>(mode, a, st);
    SPOCK_STATIC_SHAPES(X)
#undef X
    return cm == 2 ? launch_variant<DynDims<2>>(mode, a, st) : cm == 1 ? launch_variant<DynDims<1>>(mode, a, st) : launch_variant<DynDims<0>>(mode, a, st);
}

}  // namespace spock
