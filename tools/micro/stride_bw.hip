// Micro-benchmark (exploration): HBM throughput of the per-thread access shape used by the phase kernels
// (each lane owns W consecutive doubles, lanes consecutive) against a flat coalesced copy.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W> __global__ void k_chunk(const double *__restrict__ in, double *__restrict__ out, long n_chunks) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    double v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = in[t * W + k];
#pragma unroll
    for (int k = 0; k < W; ++k) out[t * W + k] = v[k] * 1.0000001;
}
__global__ void k_flat(const double2 *__restrict__ in, double2 *__restrict__ out, long n2) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n2) return;
    double2 v = in[t]; v.x *= 1.0000001; v.y *= 1.0000001; out[t] = v;
}
template <int W> __global__ void k_lds(const double *__restrict__ in, double *__restrict__ out, long n_chunks) {
    // coalesced global <-> LDS, per-lane chunks from LDS (what an LDS-staged phase kernel would do)
    __shared__ double sm[256 * W];
    const long base = (long)blockIdx.x * blockDim.x * W;
    for (int k = threadIdx.x; k < blockDim.x * W; k += blockDim.x) sm[k] = in[base + k];
    __syncthreads();
    double v[W];
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = sm[threadIdx.x * W + k] * 1.0000001;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) sm[threadIdx.x * W + k] = v[k];
    __syncthreads();
    for (int k = threadIdx.x; k < blockDim.x * W; k += blockDim.x) out[base + k] = sm[k];
}
int main() {
    const long n = 1L << 27;   // 128M doubles = 1 GiB in, 1 GiB out
    double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMemset(a, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.1f GB/s (read+write)\n", name, 2.0 * n * 8 * 5 / (ms * 1e-3) / 1e9);
    };
    for (int bs : {64, 256}) {
        printf("block %d\n", bs);
        run("  chunk W=5 (stride 40 B)", [&] { long c = n / 5; hipLaunchKernelGGL(k_chunk<5>, dim3((c + bs - 1) / bs), dim3(bs), 0, 0, a, b, c); });
        run("  chunk W=7 (stride 56 B)", [&] { long c = n / 7; hipLaunchKernelGGL(k_chunk<7>, dim3((c + bs - 1) / bs), dim3(bs), 0, 0, a, b, c); });
        run("  chunk W=2 (stride 16 B)", [&] { long c = n / 2; hipLaunchKernelGGL(k_chunk<2>, dim3((c + bs - 1) / bs), dim3(bs), 0, 0, a, b, c); });
        run("  chunk W=1", [&] { long c = n; hipLaunchKernelGGL(k_chunk<1>, dim3((c + bs - 1) / bs), dim3(bs), 0, 0, a, b, c); });
        run("  flat double2", [&] { long c = n / 2; hipLaunchKernelGGL(k_flat, dim3((c + bs - 1) / bs), dim3(bs), 0, 0, (const double2 *)a, (double2 *)b, c); });
        run("  LDS-staged W=5", [&] { long c = n / 5 / bs * bs; hipLaunchKernelGGL(k_lds<5>, dim3(c / bs), dim3(bs), 0, 0, a, b, c); });
    }
    return 0;
}
