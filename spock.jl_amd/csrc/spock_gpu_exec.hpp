// spock_gpu_exec.hpp -- the GPU execution policy of the one-thread-per-node / lane-team kernels (spock_group.hpp,
// spock_flat_local.hpp): state in registers, LDS exchange buffer, wave / LDS reductions.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "spock_device.hpp"

namespace spock {

template <class P>
struct GpuExec {
    TState<P> st;
    double m[3];          // residual maxima: reduced in double whatever `real` is
    real *ex, *lm;
    unsigned long long *red, *stamps;
    int *fl;
    template <class F>
    __device__ __forceinline__ void each(F &&f) { f((int)threadIdx.x, st); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
    // exchange among the lanes of ONE wave (lane teams of T <= 64 lanes): LDS and global accesses of the wave complete and become
    // visible to its other lanes; no s_barrier -- the waves of the workgroup keep running at their own pace
    __device__ __forceinline__ void team_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __device__ __forceinline__ real *lds() { return ex; }
    __device__ __forceinline__ real *lds_mats() { return lm; }
    __device__ __forceinline__ int &flag() { return *fl; }
    template <class T>
    __device__ __forceinline__ void stash3(int, const T *mx) { m[0] = (double)mx[0]; m[1] = (double)mx[1]; m[2] = (double)mx[2]; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ unsigned long long *stamp_lds() { return stamps; }
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
    __device__ __forceinline__ real red3(int k) const { return __longlong_as_double((long long)red[k]); }
    // deterministic sums over the workgroup (SP inner products): fixed shuffle tree per wave, then the waves in order
    double sm[3];
    double *sums;   // LDS [3][16 waves] + [3] results
    template <class T>
    __device__ __forceinline__ void stash_sum3(int, const T *v) { sm[0] = (double)v[0]; sm[1] = (double)v[1]; sm[2] = (double)v[2]; }
    __device__ __forceinline__ void reduce_sum3() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = sm[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) sums[k * 16 + wave] = v;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double v = sums[threadIdx.x * 16];
            for (int w = 1; w < nw; ++w) v += sums[threadIdx.x * 16 + w];
            sums[48 + threadIdx.x] = v;
        }
        __syncthreads();
    }
    __device__ __forceinline__ double sum3(int k) const { return sums[48 + k]; }
};

}  // namespace spock
