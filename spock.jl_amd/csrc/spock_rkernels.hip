// spock_rkernels.hip -- gfx950 kernel of the role-lane subtree kernel of the single-instance regime (spock_rgroup.hpp): wave = role,
// lane = node.  A translation unit of its own.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_rgroup.hpp"
#include "spock_shapes.hpp"

namespace spock {

template <class P>
struct GpuRExec {
    RState<P> st;
    double m[3];          // residual maxima: reduced in double whatever `real` is
    real *ex;
    unsigned long long *red;
    template <class F>
    __device__ __forceinline__ void each(F &&f) { f((int)threadIdx.x, st); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    template <class F>
    __device__ __forceinline__ void phased(F &&f) { f((int)threadIdx.x, st, true, true); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ real *lds() { return ex; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ void clocks(unsigned long long *d) { d[0] = __builtin_readcyclecounter(); d[1] = __builtin_amdgcn_s_memrealtime(); }
    template <class T>
    __device__ __forceinline__ void stash3(int, const T *mx) { m[0] = (double)mx[0]; m[1] = (double)mx[1]; m[2] = (double)mx[2]; }
    // values are |.| (sign bit clear) or NaN: the unsigned order of the bit patterns is the NaN-propagating max order
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
    __device__ __forceinline__ real red3(int k) const { return __longlong_as_double((long long)red[k]); }
    // the same maximum without barriers of its own: max3_init() ... barrier ... max3_add() ... barrier ... max3_get()
    __device__ __forceinline__ void max3_init() { if (threadIdx.x < 3) red[threadIdx.x] = 0ull; }
    template <class T>
    __device__ __forceinline__ void max3_add(int, const T *mx) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned long long b = (unsigned long long)__double_as_longlong((double)mx[k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(b, off, 64);
                b = o > b ? o : b;
            }
            if ((threadIdx.x & 63) == 0 && b) atomicMax(&red[k], b);
        }
    }
    __device__ __forceinline__ real max3_get(int k) const { return __longlong_as_double((long long)red[k]); }
};

template <class P, bool WX>
__global__ void __launch_bounds__(kRgroupThreads) spock_rgroup_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpuRExec<P> x;
    x.ex = reinterpret_cast<real *>(smem);
    x.red = reinterpret_cast<unsigned long long *>(smem + a.ex_doubles);
    rgroup_body<P, WX>(x, a, (int)blockIdx.y, (int)blockIdx.x);
}

template <class P>
static int launch_rvariant(const KArgs &a_in, hipStream_t st) {
    KArgs a = a_in;
    const Dims &dm = a.dm;
    a.ex_doubles = rgroup_lds_doubles(dm.nx, dm.nu, dm.d, a.want_xi != 0);
    const size_t lds = (size_t)a.ex_doubles * 8 + 64;
    if (lds > 128 * 1024) return (int)hipErrorInvalidConfiguration;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit (nx = nu = 5 with the L'xi_2 rows): raise it once per kernel and device
        static bool raised_dev[16][2] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
        bool always = false;
        bool &raised_flag = dev >= 0 ? raised_dev[dev][a.want_xi ? 1 : 0] : always;
        if (!raised_flag) {
            const void *fn = a.want_xi ? (const void *)spock_rgroup_kernel<P, true> : (const void *)spock_rgroup_kernel<P, false>;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            raised_flag = true;
        }
    }
    const dim3 grid(dm.nsub + 1, a.batch), block(kRgroupThreads);
    if (a.want_xi) hipLaunchKernelGGL((spock_rgroup_kernel<P, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((spock_rgroup_kernel<P, false>), grid, block, lds, st, a);
    return (int)hipGetLastError();
}

int launch_rgroup(const KArgs &a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
#define X(a_, b_, c_, e_)                                                                      \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                                \
        return cm == 2 ? launch_rvariant<StaticDims<a_, b_, c_, e_, 2>>(a, st) : cm == 1 ? launch_rvariant<StaticDims<a_, b_, c_, e_, 1>>(a, st)  \
                             : (int)hipErrorInvalidConfiguration;
    SPOCK_STATIC_SHAPES(X)
#undef X
    return (int)hipErrorInvalidConfiguration;
}

}  // namespace spock
