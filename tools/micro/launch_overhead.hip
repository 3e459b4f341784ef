// Micro-benchmark (exploration): cost of a dependent kernel boundary on one stream, eager and in a hipGraph,
// for a trivial kernel, a kernel with a large kernarg struct, and 512-thread blocks with dynamic LDS.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double pad[70]; double *p; };
__global__ void k_small(double *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += b.pad[3]; }
__global__ void k_lds(Big b) { extern __shared__ double sm[]; sm[threadIdx.x] = b.pad[1]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += sm[5]; }
template <class F> double timeit(hipStream_t s, int n, F f) {
    f(); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *p; hipMalloc(&p, 64); hipMemset(p, 0, 64);
    Big b{}; b.p = p;
    const int N = 3000;
    printf("eager small 1 WG        : %.2f us/launch\n", timeit(s, N, [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p); }));
    printf("eager small 65 WG       : %.2f us/launch\n", timeit(s, N, [&] { hipLaunchKernelGGL(k_small, dim3(65), dim3(64), 0, s, p); }));
    printf("eager bigarg 65 WG      : %.2f us/launch\n", timeit(s, N, [&] { hipLaunchKernelGGL(k_big, dim3(65), dim3(64), 0, s, b); }));
    printf("eager lds 65 WG x512 16K: %.2f us/launch\n", timeit(s, N, [&] { hipLaunchKernelGGL(k_lds, dim3(65), dim3(512), 16384, s, b); }));
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 96; ++i) {
            if (variant == 0) hipLaunchKernelGGL(k_small, dim3(65), dim3(64), 0, s, p);
            else if (variant == 1) hipLaunchKernelGGL(k_big, dim3(65), dim3(64), 0, s, b);
            else hipLaunchKernelGGL(k_lds, dim3(65), dim3(512), 16384, s, b);
        }
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double us = timeit(s, 100, [&] { hipGraphLaunch(ge, s); });
        printf("graph of 96, variant %d  : %.2f us/kernel\n", variant, us / 96);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    hipStream_t s0; hipStreamCreate(&s0);
    printf("eager small, blocking stream: %.2f us/launch\n", timeit(s0, N, [&] { hipLaunchKernelGGL(k_small, dim3(65), dim3(64), 0, s0, p); }));
    return 0;
}
