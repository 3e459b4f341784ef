// spock_bkernels.hip -- gfx950 kernel of the on-device build step (spock_build.hpp): ricatti_offline and the composite sweep
// matrices for new dynamics, one workgroup.
#include <hip/hip_runtime.h>

#include "spock_build.hpp"

namespace spock {

struct GpuBuildExec {
    template <class F>
    __device__ __forceinline__ void par(int n, F &&f) {
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) f(i);
        __syncthreads();
    }
    template <class F>
    __device__ __forceinline__ void one(F &&f) {
        if (threadIdx.x == 0) f();
        __syncthreads();
    }
};

__global__ void __launch_bounds__(256) spock_build_riccati_kernel(const BuildArgs a_in) {
    // one workgroup per instance: a batch of draws of the dynamics builds side by side, each into its own pool
    BuildArgs a = a_in;
    const long b = blockIdx.x;
    const Dims &dm = a_in.dm;
    a.mats = a_in.mats + b * a_in.mats_stride;
    a.A = a_in.A + b * dm.d * dm.nx * dm.nx; a.B = a_in.B + b * dm.d * dm.nx * dm.nu;
    a.ws = a_in.ws + b * build_ws_doubles_dev(dm.nx, dm.nu, dm.d); a.status = a_in.status + b;
    GpuBuildExec x;
    build_riccati_body(x, a);
}

int launch_build_riccati(const BuildArgs &a, void *stream) {
    hipLaunchKernelGGL(spock_build_riccati_kernel, dim3(a.n_inst > 0 ? a.n_inst : 1), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace spock
