// Micro-benchmark: the LBA workgroup structure without arithmetic -- five role waves (X, R, C, L0, L1) per workgroup, every role
// loads / stores its rows of a node (512 B each), one barrier per node, `npw` nodes per workgroup.  Variants: with / without the
// barrier, workgroups per CU (through the LDS size), rows in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/role_pattern.hip -o tools/micro/role_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0)); }
__device__ __forceinline__ void bstore(double x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, v, s, 0); }

// unique rows per role (X: x_i d_i z.x z.u v7 | R | C: 2 x 11 | L: 19) -- the redundant re-loads of x_i, d_i, z.x, z.u by C / L come on top (RED)
template <int NL, int NS, int NS2, int RED>
__device__ __forceinline__ void role(__amdgpu_buffer_rsrc_t ri, __amdgpu_buffer_rsrc_t ro, unsigned voff, unsigned rs8, int rbase, int wbase, int xbase,
                                     int node, bool bar, double *lds, int lane) {
    double t[NL + RED];
#pragma unroll
    for (int k = 0; k < RED; ++k) t[NL + k] = bload(ri, voff, (unsigned)(xbase + node * 21 + k) * rs8);
#pragma unroll
    for (int k = 0; k < NL; ++k) t[k] = bload(ri, voff, (unsigned)(rbase + node * NL + k) * rs8);
    double acc = 0.;
#pragma unroll
    for (int k = 0; k < NL + RED; ++k) acc += t[k];
#pragma unroll
    for (int k = 0; k < NS; ++k) bstore(acc + k, ro, voff, (unsigned)(wbase + node * (NS + NS2) + k) * rs8);
    if (NS2 == 0) { for (int k = 0; k < 8; ++k) lds[k * 64 + lane] = acc; }
    if (bar) __syncthreads();
    if (NS2 > 0) {
        for (int k = 0; k < 16; ++k) acc += lds[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < NS2; ++k) bstore(acc + k, ro, voff, (unsigned)(wbase + node * (NS + NS2) + NS + k) * rs8);
    }
}

__global__ void __launch_bounds__(320) role_kernel(const double *in, double *out, unsigned rs8, int nodes, int npw, int bar) {
    extern __shared__ double lds[];
    const int tile = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (long)tile * 64), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (long)tile * 64), 0, 0xffffffff, 0x00020000);
    const unsigned voff = lane * 8;
    // row bases of the roles' regions (nodes x rows each)
    const int rX = 0, rR = rX + nodes * 21, rC = rR + nodes * 18, rL0 = rC + nodes * 22, rL1 = rL0 + nodes * 19;
    const int wX = 0, wR = wX + nodes * 16, wC = wR + nodes * 15, wL0 = wC + nodes * 18, wL1 = wL0 + nodes * 17;
    for (int n = 0; n < npw; ++n) {
        const int node = blockIdx.x * npw + n;
        if (node >= nodes) break;
        double *l = lds + ((n & 1) * 40 + (wave >= 2 ? (wave - 2) * 8 : 0)) * 64;
        if (wave == 0) role<21, 9, 7, 0>(ri, ro, voff, rs8, rX, wX, rX, node, bar, lds + (n & 1) * 40 * 64, lane);
        else if (wave == 1) role<18, 6, 9, 0>(ri, ro, voff, rs8, rR, wR, rX, node, bar, lds + (n & 1) * 40 * 64, lane);
        else if (wave == 2) role<22, 18, 0, 14>(ri, ro, voff, rs8, rC, wC, rX, node, bar, l, lane);
        else if (wave == 3) role<19, 17, 0, 7>(ri, ro, voff, rs8, rL0, wL0, rX, node, bar, l, lane);
        else role<19, 17, 0, 7>(ri, ro, voff, rs8, rL1, wL1, rX, node, bar, l, lane);
    }
}

int main() {
    const int bp = 2048, nodes = 2048, tiles = bp / 64;
    const long rrows = (long)nodes * 99, wrows = (long)nodes * 83;
    double *in, *out;
    CK(hipMalloc(&in, rrows * bp * 8)); CK(hipMalloc(&out, wrows * bp * 8));
    CK(hipMemset(in, 0, rrows * bp * 8)); CK(hipMemset(out, 0, wrows * bp * 8));
    CK(hipFuncSetAttribute((const void *)role_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)nodes * tiles * (99 + 83) * 512.;
    printf("unique bytes per launch %.2f GB\n", bytes * 1e-9);
    for (int bar = 1; bar >= 0; --bar)
        for (int wgs : {2, 3, 4, 5, 6})
            for (int npw : {8, 32}) {
                const int lds = 160 * 1024 / wgs - 512;
                dim3 grid((nodes + npw - 1) / npw, tiles);
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(role_kernel, grid, dim3(320), lds, 0, in, out, (unsigned)bp * 8u, nodes, npw, bar);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                printf("barrier=%d wgs/CU=%d (%2d waves) npw=%2d : %7.3f ms  %7.1f GB/s\n", bar, wgs, wgs * 5, npw, ms, bytes / ms * 1e-6);
            }
    return 0;
}
