// spock_ikernels.hip -- gfx950 kernels of the element-major batched regime (spock_igroup.hpp): one lane per problem
// instance, one wave per tree node; plus the layout conversions between the API's instance-major arrays and that layout.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spock_igroup.hpp"
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

// The iterate buffers come in as separate restrict-qualified kernel arguments (not inside KArgs): the compiler may then move
// the loads of a node's inputs ahead of the stores of the previous node's outputs.
#ifndef SPOCK_ILV_MIN_WAVES
#define SPOCK_ILV_MIN_WAVES 2   // waves per SIMD the node-local kernels are compiled for (<= 256 VGPRs)
#endif
template <class P, int MODE, bool WX>
__global__ void __launch_bounds__(256, MODE == MODE_FINALIZE ? 1 : SPOCK_ILV_MIN_WAVES)
spock_igroup_kernel(const KArgs a, const real *__restrict__ zin, const real *__restrict__ vin, real *__restrict__ zout,
                    real *__restrict__ vout, real *__restrict__ znext) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // carved in units of 8 bytes (ilv_geometry) whatever `real` is
    GpuIExec<P> x;
    x.ex = reinterpret_cast<real *>(smem);
    x.rd = smem + a.ex_doubles;
    x.lf = reinterpret_cast<int *>(x.rd + (blockDim.x >> 6) * 3 * 64);
    igroup_body<P, MODE, WX>(x, a, zin, vin, zout, vout, znext, (int)blockIdx.y + a.tile0, (int)blockIdx.x);
}

template <class P>
static int launch_ivariant(int mode, const KArgs &a_in, hipStream_t st) {
    const IlvGeo g = ilv_geometry(mode, a_in.dm, a_in.grp_level, a_in.want_xi);
    KArgs a = a_in;
    a.ex_doubles = g.ex_doubles; a.cchunk = g.cchunk; a.n_cchunks = g.n_cchunks;
    a.ls_cnt = g.ls_cnt; a.ls_first = g.ls_first; a.ls_first_child = g.ls_first_child; a.ls_slot = g.ls_slot; a.ls_chunk_k = g.ls_chunk_k;

    const dim3 grid(g.grid_x, (unsigned)(a.n_tiles > 0 ? a.n_tiles : a.bp / 64)), block(g.threads);
    const size_t lds = g.lds_bytes;
    if (lds > 64 * 1024) return (int)hipErrorInvalidConfiguration;   // ilv_geometry / choose_split_stage_ilv keep it below
    // znext aliases nothing the kernel reads (third z buffer); for SUB_A / CROWN it is unused
    const real *cn = nullptr;
    real *dn = nullptr;
    switch (mode) {
        case MODE_SUB_A: hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_SUB_A, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, dn); break;
        case MODE_ICROWN_A: hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_ICROWN_A, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, dn); break;
        case MODE_CROWN:
            if (a.want_xi) hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_CROWN, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            else hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_CROWN, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            break;
        case MODE_SUB_BA:
            if (a.want_xi) hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_SUB_BA, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            else hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_SUB_BA, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            break;
        case MODE_I_DOWN:
            if (a.want_xi) hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_I_DOWN, true>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            else hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_I_DOWN, false>), grid, block, lds, st, a, a.z_in, a.v_in, a.z_out, a.v_out, a.z_next);
            break;
        case MODE_I_UP: hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_I_UP, false>), grid, block, lds, st, a, cn, cn, dn, dn, dn); break;
        case MODE_FINALIZE: hipLaunchKernelGGL((spock_igroup_kernel<P, MODE_FINALIZE, false>), grid, block, lds, st, a, cn, cn, dn, dn, dn); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) spock_transpose_kernel(real *__restrict__ dst, const real *__restrict__ src, long rows, long cols,
                                                              long ld_src, long ld_dst) {
    __shared__ real tile[32][33];
    // one-dimensional grid (block = column tile + row tile * column tiles): nz / 32 row tiles pass grid.y's 65535 from nz ~ 2.1 M on
    const long nbx = (cols + 31) / 32;
    const long r0 = ((long)blockIdx.x / nbx) * 32, c0 = ((long)blockIdx.x % nbx) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = src[r * ld_src + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) dst[c * ld_dst + r] = tile[tx][k];
    }
}

int launch_transpose(real *dst, const real *src, long rows, long cols, long ld_src, long ld_dst, void *stream) {
    if (rows <= 0 || cols <= 0) return 0;
    const long nblk = ((cols + 31) / 32) * ((rows + 31) / 32);
    if (nblk > 0x7fffffffL) return (int)hipErrorInvalidConfiguration;
    const dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL(spock_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, dst, src, rows, cols, ld_src, ld_dst);
    return (int)hipGetLastError();
}

// after a solve: the final iterate of instance b sits in the buffers its control block's result_parity names; bring it to slot 0
__global__ void __launch_bounds__(256) spock_ilv_collect_kernel(real *z0, const real *z1, const real *z2, real *v0, const real *v1,
                                                                const Ctl *ctl, long nz, long nv, long bp, int batch) {
    const long b = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (b >= batch) return;
    const int rp = ctl[b].result_parity;
    const real *zs = (rp % 3 == 1) ? z1 : ((rp % 3 == 2) ? z2 : nullptr);
    const real *vs = (rp % 2) ? v1 : nullptr;
    const int sub = threadIdx.x >> 6;   // 4 element rows per pass
    if (zs) for (long e = (long)blockIdx.y * 4 + sub; e < nz; e += (long)gridDim.y * 4) z0[e * bp + b] = zs[e * bp + b];
    if (vs) for (long e = (long)blockIdx.y * 4 + sub; e < nv; e += (long)gridDim.y * 4) v0[e * bp + b] = vs[e * bp + b];
}

int launch_ilv_collect(real *const *z3, real *const *v2, const Ctl *ctl, long nz, long nv, long bp, int batch, void *stream) {
    const dim3 grid((unsigned)(bp / 64), 256);
    hipLaunchKernelGGL(spock_ilv_collect_kernel, grid, dim3(256), 0, (hipStream_t)stream, z3[0], z3[1], z3[2], v2[0], v2[1], ctl, nz, nv, bp, batch);
    return (int)hipGetLastError();
}


int launch_ilevel(int mode, const KArgs &a, void *stream);   // spock_lkernels.hip

int launch_igroup(int mode, const KArgs &a, void *stream) {
    if (mode == MODE_LB || mode == MODE_LBA || mode == MODE_LA || mode == MODE_LA0 || mode == MODE_LBA2) return launch_ilevel(mode, a, stream);
    hipStream_t st = (hipStream_t)stream;
    const Dims &dm = a.dm;
    const int cm = (dm.nQ == 1 && dm.nR == 1 && dm.nQN == 1) ? (dm.diag_costs ? 2 : 1) : 0;
#define X(a_, b_, c_, e_)                                                                      \
    if (dm.nx == a_ && dm.nu == b_ && dm.d == c_ && dm.ny == e_)                                \
        return cm == 2 ? launch_ivariant<StaticDims<a_, b_, c_, e_, 2>>(mode, a, st) : cm == 1 ? launch_ivariant<StaticDims<a_, b_, c_, e_, 1>>(mode, a, st)  \
                             : (int)hipErrorInvalidConfiguration;   /* per-node cost matrices: instance-major kernels (spock_create) */
    SPOCK_STATIC_SHAPES(X)
#undef X
    return (int)hipErrorInvalidConfiguration;
}

}  // namespace spock
