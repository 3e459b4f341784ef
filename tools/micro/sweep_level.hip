// Micro-benchmark: ONE workgroup running a chain of S1 sweep levels on the matrix cores (bwd_level_mfma / fwd_level_mfma of
// spock_device.hpp, nx = 64, nu = 16, d = 2), as the subtree / crown kernels do: what does a level cost, and which part of it?
//   variants: 0 as shipped (fragments fetched from global memory every level) | 1 fragments fetched once (registers reused: the
//   matrix-core chain + LDS operands + barrier alone) | levels of 1 / 16 / 64 nodes
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spock.jl_amd/csrc tools/micro/sweep_level.hip -o tools/micro/sweep_level.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "spock_device.hpp"
using namespace spock;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef DynDims<2, true, true> PP;

template <int NX, int NU, int DIR>
__global__ void __launch_bounds__(1024) level_kernel(Dims dm, const real *mats, int cnt, int reps, int nlev, double *out) {
    extern __shared__ double smem[];
    const PP p(dm);
    TeamLds tl;
    const int nl = 2 * cnt + 1;
    tl.XQ = smem; tl.ED = tl.XQ + nl * NX; tl.UL = tl.ED + nl * NU; tl.XL = tl.UL + nl * NU;
    for (int k = threadIdx.x; k < nl * (NX + 2 * NU); k += blockDim.x) smem[k] = 1e-3 * (k % 17);
    __syncthreads();
    const MatPtr M = (MatPtr)mats;
    for (int r = 0; r < reps; ++r) {
        for (int l = 0; l < nlev; ++l) {
            const MatPtr SW = M + dm.mSweep + l * dm.sw_stride;
            if (DIR == 0) bwd_level_mfma<NX, NU, 2>(p, dm, SW, 0, cnt, false, (int)threadIdx.x, (int)blockDim.x, tl);
            else fwd_level_mfma<NX, NU, 2>(p, dm, SW, M + dm.mB, 0, cnt, false, (int)threadIdx.x, (int)blockDim.x, tl);
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) out[0] = tl.XQ[0];
}

template <int NX, int NU>
void run(const char *name) {
    Dims dm; std::memset(&dm, 0, sizeof(dm));
    dm.nx = NX; dm.nu = NU; dm.d = 2;
    dm.sw_oW = 0; dm.sw_oV = 2 * NX * NX; dm.sw_oABK = dm.sw_oV + 2 * NU * NX; dm.sw_oK = dm.sw_oABK + 2 * NX * NX; dm.sw_stride = dm.sw_oK + NU * NX;
    const int nstage = 14;
    dm.mB = 0; dm.mSweep = 2 * NX * NU;
    std::vector<double> h((size_t)dm.mSweep + (size_t)nstage * dm.sw_stride);
    for (size_t k = 0; k < h.size(); ++k) h[k] = 1e-3 * ((k * 7) % 13) - 5e-3;
    double *d, *o; CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&o, 64));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int dir = 0; dir < 2; ++dir)
        for (int cnt : {1, 8, 16, 32, 64}) {
            const size_t lds = (size_t)(2 * cnt + 1) * (NX + 2 * NU) * 8 + 1024;
            auto fn = dir == 0 ? level_kernel<NX, NU, 0> : level_kernel<NX, NU, 1>;
            CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const int reps = 200, nlev = 7;
            hipLaunchKernelGGL(fn, dim3(1), dim3(1024), lds, 0, dm, d, cnt, 2, nlev, o);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(1), dim3(1024), lds, 0, dm, d, cnt, reps, nlev, o);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s %s cnt %2d: %.2f us per level\n", name, dir == 0 ? "bwd" : "fwd", cnt, ms * 1e3 / (reps * nlev));
        }
}
int main() { run<64, 16>("nx64/nu16"); run<20, 20>("nx20/nu20"); return 0; }
